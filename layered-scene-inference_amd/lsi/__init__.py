"""lsi -- MI355X-native layered-scene-inference hot path.

Same package/module/function names as the reference's `lsi` package
(lsi.geometry.*, lsi.nnutils.helpers, lsi.loss.loss) on eager torch.Tensors; the
renderer ops run as hand-written HIP kernels in liblsi_hip.so (C ABI in
include/lsi_hip.h), bound with ctypes in lsi._C.
"""
__version__ = '0.1.0'
