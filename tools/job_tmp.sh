cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sampling_gpu.py -x -q -k "scene_generator" 2>&1 | grep -A12 "^E " | head -40
python - <<'PY'
import sys
sys.path.insert(0,'layered-scene-inference_amd')
import torch
from lsi.data import synthetic_planes
gen = synthetic_planes.SceneGenerator(128, 128, n_obj=2, device='cuda', seed=3)
src, trg, k_s, k_t, rot, t, d_src, d_trg = gen.forward(2)
print(src.shape, float(src.min()), float(src.max()), float(d_src.min()), float(d_src.max()), float(d_trg.min()), float(d_trg.max()))
print(rot[0], t[0])
PY
