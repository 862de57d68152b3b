"""pytest configuration: markers, import paths, in-tree builds."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'layered-scene-inference_amd')
for p in (PKG, os.path.join(ROOT, 'oracle'), ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (ROCm device)')


@pytest.fixture(scope='session')
def built_lib():
  """liblsi_hip.so, (re)built when hipcc is available, else used as shipped."""
  so = os.path.join(PKG, 'liblsi_hip.so')
  if shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc'):
    sys.path.insert(0, PKG)
    import build as lsi_build  # layered-scene-inference_amd/build.py
    lsi_build.build()
  if not os.path.exists(so):
    pytest.fail('liblsi_hip.so missing and hipcc not available')
  return so


@pytest.fixture(scope='session')
def ref_cpu():
  """The plain-C oracle (oracle/lsi_ref_cpu.c), built with gcc."""
  import ref_cpu as rc
  rc.build()
  return rc


def golden(name):
  import numpy as np
  return np.load(os.path.join(GOLDEN, name))
