// Internal (non-ABI) declarations shared by the splat translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lsi_hip.h"

struct SplatArgs {
  LsiSplatDesc d;
  const float* tex;
  const float* disp;
  const float* mask;
  const float* M;
  float* out_img;
  float* out_wts;
  float* out_disp;
  float* canvas;  // caller's workspace: ATOMIC canvases / STREAM exchange area
  size_t ws_bytes;
  int nch;        // canvas channels (4, or 5 with disparity)
  int ncanv;      // canvases per batch element (1 or L)
  int shared;     // 1: compose without disparity, all layers share a canvas
  int band_rows;  // ROWBAND: target rows per workgroup
  // lsi_splat_fwd_both: the composed outputs next to the per-layer ones (NULL
  // otherwise)
  float* out_img_c;
  float* out_wts_c;
};

// lsi_stream_ok's return value (kept in LsiSplatDesc.tune_window): window cells,
// plus this bit when every batch element has M rows 2, 3 = (0,0,1,0), (0,0,0,1).
constexpr int LSI_STREAM_SIMPLE_BIT = 1 << 20;
// (also set by lsi_stream_ok) every batch element's vertical magnification
// M[1][1] is >= 1: a band of R target rows reads at most ceil((R + 1) / s)
// source rows, what the launchers plan their tables for
constexpr int LSI_STREAM_ROWS_BIT = 1 << 21;
constexpr int LSI_STREAM_FLAG_BITS = LSI_STREAM_SIMPLE_BIT | LSI_STREAM_ROWS_BIT;

// LSI_PATH_STREAM launcher and workspace need (lsi_splat_stream.hip).
size_t lsi_stream_workspace_bytes(const LsiSplatDesc* d);
int lsi_stream_launch(const SplatArgs& a, hipStream_t stream);
// The compact STREAM instance (lsi_splat_stream2.hip): compose mode, no mask,
// unit normaliser, channels-last textures or RGBD pixels, W % 4 == 0.
bool lsi_stream2_applies(const SplatArgs& a, bool simple, int layout);
// disp_pass: the per-layer-tile instance as the disparity pass (only output:
// a.out_disp = max over layers of each layer's normalised splatted disparity)
int lsi_stream2_launch(const SplatArgs& a, int wmax, hipStream_t stream,
                       bool disp_pass = false);

// The streamed backward for rectified pairs (lsi_splat_bwd_stream.hip).  It
// derives the gradient canvas from the forward's outputs and their incoming
// gradients itself (no pre-pass): `ci` the per-layer (or the only) canvases,
// `cc` lsi_splat_bwd_both's composed one; either may be NULL / have g_img NULL.
struct LsiBwdCanvas {
  const float* img;
  const float* wts;
  const float* g_img;
  const float* g_wts;  // may be NULL
};
bool lsi_bwd_stream_applies(const LsiSplatDesc* d, const float* tex,
                            const float* disp, const float* mask,
                            const float* g_tex, const float* g_disp,
                            const float* g_mask);
int lsi_bwd_stream_launch(const LsiSplatDesc* d, const float* tex,
                          const float* disp, const float* mask, const float* M,
                          const LsiBwdCanvas* ci, const LsiBwdCanvas* cc,
                          float* g_tex, float* g_disp, float* g_mask,
                          hipStream_t stream);

// LSI_PATH_TILE launcher and workspace need (lsi_splat_tile.hip).
size_t lsi_tile_workspace_bytes(const LsiSplatDesc* d);
int lsi_tile_launch(const SplatArgs& a, hipStream_t stream);

// The any-pose sweep kernel (lsi_splat_sweep.hip), launched by lsi_tile_launch
// after the disparity ranges (range[(l * B + b) * LSI_RANGE_SLICES + k] =
// {min, max} of a slice of rows) are on the stream.
#define LSI_RANGE_SLICES 8
#define LSI_SWEEP_MAXL 16
int lsi_sweep_launch(const SplatArgs& a, const float2* range, hipStream_t stream);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when this (device,
// kernel) has not been granted `bytes` yet: an eager caller launches the same
// kernel with the same plan thousands of times (lsi_splat.hip).
int lsi_ensure_dynamic_lds(const void* fn, size_t bytes);

// Batch-norm workspace layout per group, in floats (lsi_bn.hip; the convolution
// kernels that accumulate the statistics in their epilogue write the same
// places): [0] arrival counter (int), [ACC, ACC + 4096) the accumulators -- both
// zero between launches --, from CONST the 2 C constants of the second pass and
// (backward) the group's C sums of dz.
#define LSI_BN_WS_ACC 16
#define LSI_BN_WS_CONST (16 + 4096)
#define LSI_BN_WS_STRIDE (16 + 4096 + 3 * 2048)
// Statistics left by a convolution's epilogue (lsi_conv2d_*_bnstats): plain sums
// of y and y * y in `lsi_bn_stat_slots(C)` copies of the accumulators (slot s of
// a group: ACC + s * 2 C; thousands of workgroups adding to the same two cache
// lines would take ~8 ns each, one after the other), folded, turned into the
// constants and cleared by lsi_bn_relu_norm.
// The hand-over is checked on the device: the producer's first workgroup of a
// group leaves LSI_BN_TAG(C, groups) in the group's word [1]; lsi_bn_relu_norm
// expects exactly that tag, the kernels that accumulate their own statistics
// (lsi_bn_relu_fwd / _bwd) expect 0.  A kernel that finds something else writes
// NaN constants (its output is NaN: loud in any loss), and clears accumulators
// and tag, so that the calls after it are right again.
#define LSI_BN_WS_TAG 1
#define LSI_BN_TAG(C, groups) (0x5A000000 | (((groups) & 0xfff) << 12) | ((C) & 0xfff))
static inline int lsi_bn_stat_slots(int C) {
  int ns = 1;
  while (ns < 32 && 2 * ns * 2 * C <= 4096) ns *= 2;
  return ns;
}
