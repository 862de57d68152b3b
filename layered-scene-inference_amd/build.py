"""Builds liblsi_hip.so (gfx950) in-tree with hipcc.

    python layered-scene-inference_amd/build.py [--force]

The shared library is plain C ABI (include/lsi_hip.h); it links only the HIP
runtime.  Cross-compiles without a GPU.  The .so is git-ignored but travels
with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
SO = os.path.join(PKG, 'liblsi_hip.so')
SOURCES = ['lsi_splat.hip', 'lsi_splat_stream.hip', 'lsi_splat_stream2.hip',
           'lsi_splat_tile.hip', 'lsi_splat_bwd_stream.hip',
           'lsi_splat_sweep.hip',
           'lsi_sampling.hip', 'lsi_loss.hip', 'lsi_bn.hip', 'lsi_host.hip', 'lsi_conv.hip',
           'lsi_conv_wgrad.hip', 'lsi_conv_igemm.hip', 'lsi_conv_wgrad_igemm.hip',
           'lsi_conv_first.hip']
HEADERS = [os.path.join(CSRC, 'lsi_common.h'),
           os.path.join(CSRC, 'lsi_splat_internal.h'),
           os.path.join(ROOT, 'include', 'lsi_hip.h')]

HIPCC_FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
    '-ffp-contract=off',        # index arithmetic must not be FMA-contracted
    '-munsafe-fp-atomics',      # fp32 atomicAdd -> global_atomic_add_f32/ds_add_f32
    '-fno-fast-math', '-Wall', '-Wno-unused-function',
]


def hipcc():
  exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  if not os.path.exists(exe):
    raise RuntimeError('hipcc not found; ROCm toolchain is required')
  return exe


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


SO_HOOKS = os.path.join(PKG, 'liblsi_hip_hooks.so')
LAST_BUILD = {}   # object / library -> 'compiled' | 'linked' | 'reused' (last build() call)


def build(force=False, verbose=False, hooks=False):
  """hooks=True builds liblsi_hip_hooks.so: the same library with the stream
  kernel's timing-experiment hooks compiled in (tools/phase_probe.py,
  bench.py --debug-flags; select it with LSI_HIP_LIB=hooks)."""
  srcs = [os.path.join(CSRC, s) for s in SOURCES]
  so = SO_HOOKS if hooks else SO
  ext = '.hooks.o' if hooks else '.o'
  flags = HIPCC_FLAGS + (['-DLSI_STREAM_HOOKS=1'] if hooks else [])
  objs, jobs = [], []
  report = LAST_BUILD
  report.clear()
  if not force and not _stale(so, srcs + HEADERS):
    # (the library is newer than every source and header: nothing to do -- the
    # objects need not exist)
    report[os.path.basename(so)] = 'reused'
    return so
  for src in srcs:
    obj = src[:-4] + ext
    stale = force or _stale(obj, [src] + HEADERS)
    report[os.path.basename(obj)] = 'compiled' if stale else 'reused'
    if stale:
      cmd = [hipcc()] + flags + ['-c', src, '-o', obj]
      if verbose:
        print(' '.join(cmd))
      jobs.append((cmd, subprocess.Popen(cmd)))  # the sources compile side by side
    objs.append(obj)
  for cmd, proc in jobs:
    if proc.wait() != 0:
      raise subprocess.CalledProcessError(proc.returncode, cmd)
  relink = force or _stale(so, objs)
  report[os.path.basename(so)] = 'linked' if relink else 'reused'
  if relink:
    cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + objs
    if verbose:
      print(' '.join(cmd))
    subprocess.check_call(cmd)
  return so


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose=True,
              hooks='--hooks' in sys.argv))
  print(LAST_BUILD)
