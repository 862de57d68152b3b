// 3x3 stride-1 SAME convolutions over 32 input channels on the matrix cores
// (gfx950, v_mfma_f32_16x16x32_bf16): the full-resolution layers of the LDI
// heads, reference nets.py:104-111 (`upcnv1b`: 32 -> 32, before batch norm) and
// nets.py:150-158 (`pred_l`: 32 -> 4 or 5, bias, sigmoid).  At 256 x 768 these
// two layers per LDI layer are where the library convolutions are furthest from
// the hardware: 32 input channels make K = 288, and MIOpen's implicit-GEMM
// kernels run them at 6 % (`upcnv1b`) and 0.6 % (`pred_l`) of the bf16 MFMA
// peak (profiles/r04/conv_util_bf16.json) -- they are bound by the activation
// traffic, which this kernel moves once.
//
// Formulation.  Activations are channels-last bf16 (N x H x W x 32: what the
// fused batch-norm kernels write), so one pixel's 32 channels are 64
// contiguous bytes -- exactly the K = 32 of one MFMA.  A 3x3 convolution is
// nine such MFMAs per 16-pixel tile, one per tap:
//     D[co][px] += A_tap[co][ci] * B_tap[ci][px],   B_tap[ci][px] = x[px + tap][ci]
// The WEIGHTS are the A operand (16 output channels x 32 input channels per
// tap, put into fragment order once per workgroup, resident in registers), the
// PIXELS the B operand (lane l holds channels 8 (l >> 4) .. + 7 of pixel
// l & 15: one 16-byte load), so that the accumulator of lane l is four
// consecutive output channels 4 (l >> 4) .. + 3 of ONE pixel: channels-last
// stores without a transpose, and in the `pred_l` form the 16 lanes of the
// first quarter hold a pixel's RGBD value -- bias, sigmoid and one 16-byte
// fp32 store (the RGBD pixels the renderer reads packed).
//
// A wave owns a strip of 16 pixel columns and walks down a chunk of rows; the
// fragments of three input rows x three column shifts live in registers (the
// shifted fragments are L1 hits of the same lines), the next row's loads are in
// flight while a row is multiplied: every activation is fetched from memory
// once (plus 2 halo rows per 32-row chunk), every output written once.
// Zero padding = zero fragments.  fp32 accumulation; bf16 inputs as under
// torch.autocast (what the MIOpen path of the same layer computes).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lsi_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CV_ROWS = 32;  // output rows per wave (row chunk)

struct ConvArgs {
  const __bf16* x;   // N x H x W x 32
  const float* w;    // cout x 32 x 3 x 3 (the layer's parameter)
  int tr;            // 1: data gradient (weights transposed and flipped)
  const float* bias; // [cout] or NULL
  void* out;         // plain: bf16 N x H x W x cout (cout = 16 COT); pred: fp32 N x H x W x 4
  int N, H, W, cout;
  float scale3;      // pred: factor of channel 3 (the disparity) after the sigmoid
};

// COT: tiles of 16 output channels; PRED: bias + sigmoid, fp32 RGBD pixels
template <int COT, bool PRED>
__global__ __launch_bounds__(256) void conv3x3_c32_kernel(ConvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int x0 = (blockIdx.x * 4 + wave) * 16;
  const int n = blockIdx.z;
  const int y0 = blockIdx.y * CV_ROWS;
  const int y1 = min(a.H, y0 + CV_ROWS);
  const int H = a.H, W = a.W;

  // The weights as A fragments, straight from the layer's fp32 parameter
  // (cout x 32 x 3 x 3): converted to bf16 and laid out in fragment order in
  // LDS by the workgroup, one 16-byte read per fragment and lane.  Lane
  // (g = lane / 16, r = lane % 16) of tile c, tap (ky, kx) holds
  // A[co = 16 c + r][ci = 8 g .. 8 g + 7]; `tr` (the data gradient): the
  // transposed, flipped kernel A[co][ci] = W[ci][co][2 - ky][2 - kx].
  __shared__ bf16x8 wl[9 * COT * 64];
  for (int i = threadIdx.x; i < 9 * COT * 64 * 8; i += 256) {
    const int j = i & 7, ln = (i >> 3) & 63, tc = i >> 9;
    const int c = tc % COT, t = tc / COT;
    const int ky = t / 3, kx = t - 3 * ky;
    const int co = 16 * c + (ln & 15), ci = 8 * (ln >> 4) + j;
    float v = 0.0f;
    if (co < a.cout)
      v = a.tr ? a.w[((ci * 32 + co) * 3 + (2 - ky)) * 3 + (2 - kx)]
               : a.w[((co * 32 + ci) * 3 + ky) * 3 + kx];
    reinterpret_cast<__bf16*>(wl)[i] = (__bf16)v;
  }
  __syncthreads();
  if (x0 >= a.W) return;
  bf16x8 wf[9][COT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < COT; ++c) wf[t][c] = wl[(t * COT + c) * 64 + lane];

  const __bf16* const xn = a.x + (size_t)n * H * W * 32 + 8 * g;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  // the three column shifts of input row y for this lane's pixel (zero outside)
  auto load_row = [&](int y, bf16x8 (&f)[3]) {
    const bool yin = y >= 0 && y < H;
    const int yc = min(max(y, 0), H - 1);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int x = x0 + px + d - 1;
      const int xc = min(max(x, 0), W - 1);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(xn + ((size_t)yc * W + xc) * 32);
      f[d] = (yin && x >= 0 && x < W) ? v : zero;
    }
  };

  bf16x8 f0[3], f1[3], f2[3], f3[3];
  load_row(y0 - 1, f0);
  load_row(y0, f1);
  load_row(y0 + 1, f2);
  for (int y = y0; y < y1; ++y) {
    load_row(y + 2, f3);  // (in flight while this row is multiplied)
    f32x4 acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int c = 0; c < COT; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0 + d][c], f0[d], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[3 + d][c], f1[d], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[6 + d][c], f2[d], acc[c], 0, 0, 0);
      }
    const size_t pix = ((size_t)n * H + y) * W + x0 + px;
    if (PRED) {
      // rows 0 .. 3 of the accumulator tile = this pixel's 4 output channels
      if (g == 0) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float z = acc[0][k] + ((a.bias && k < a.cout) ? a.bias[k] : 0.0f);
          v[k] = 1.0f / (1.0f + __expf(-z));
        }
        v[3] *= a.scale3;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + 4 * pix) =
            f32x4{v[0], v[1], v[2], v[3]};
      }
    } else {
      __bf16* const o = reinterpret_cast<__bf16*>(a.out) + pix * a.cout + 4 * g;
#pragma unroll
      for (int c = 0; c < COT; ++c) {
        const bf16x4 r = {(__bf16)acc[c][0], (__bf16)acc[c][1], (__bf16)acc[c][2],
                          (__bf16)acc[c][3]};
        *reinterpret_cast<bf16x4*>(o + 16 * c) = r;
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { f0[d] = f1[d]; f1[d] = f2[d]; f2[d] = f3[d]; }
  }
}

// ---- backward of the prediction head (`pred_l`: 32 -> 4, bias, sigmoid) --------
// y = sigmoid(z), z = conv(x, W) + b.  gz = g * y * (1 - y) is formed in
// registers from the incoming gradient g and the saved output y (fp32 RGBD
// pixels, N x H x W x 4) -- never written.
//
// Data gradient: dL/dx[q][ci] = sum over taps (ky, kx) and co of
// gz[q + (1 - ky, 1 - kx)][co] * W[co][ci][ky][kx]: K = 9 taps x 4 channels = 36,
// padded to two MFMA steps of 32.  The weights are the A operand again
// (A[ci][k = 4 tap + co]), the gradients the B operand: lane (px, g) of step 0
// holds the gz of taps 2g and 2g + 1 at its pixel (two fp32 RGBD loads of g and
// of y each, L1 hits), of step 1 tap 8 (g = 0) or zeros.  Output bf16
// N x H x W x 32 as in the forward kernel.
struct PredBwdArgs {
  const float* g;    // N x H x W x 4 (only the first `cout` channels count)
  const float* y;    // N x H x W x 4 (the forward's output)
  const float* w;    // cout x 32 x 3 x 3
  const __bf16* x;   // (unused by the data gradient)
  void* out;         // bf16 N x H x W x 32
  int N, H, W, cout;
};

__device__ __forceinline__ f32x4 pred_gz(const PredBwdArgs& a, int n, int y, int x) {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (y < 0 || y >= a.H || x < 0 || x >= a.W) return r;
  const size_t pix = ((size_t)n * a.H + y) * a.W + x;
  const f32x4 gg = *reinterpret_cast<const f32x4*>(a.g + 4 * pix);
  const f32x4 yy = *reinterpret_cast<const f32x4*>(a.y + 4 * pix);
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = k < a.cout ? gg[k] * yy[k] * (1.0f - yy[k]) : 0.0f;
  return r;
}

__global__ __launch_bounds__(256) void pred_bwd_data_kernel(PredBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int x0 = (blockIdx.x * 4 + wave) * 16;
  const int n = blockIdx.z;
  const int y0 = blockIdx.y * CV_ROWS;
  const int y1 = min(a.H, y0 + CV_ROWS);
  // A fragments: [step 2][ci tile 2][64 lanes] x 8 bf16; lane (g, r) of tile c,
  // step s: A[ci = 16 c + r][k = 32 s + 8 g + j], k = 4 tap + co
  __shared__ bf16x8 wl[2 * 2 * 64];
  for (int i = threadIdx.x; i < 2 * 2 * 64 * 8; i += 256) {
    const int j = i & 7, ln = (i >> 3) & 63, c = (i >> 9) & 1, st = i >> 10;
    const int ci = 16 * c + (ln & 15), k = 32 * st + 8 * (ln >> 4) + j;
    const int tap = k >> 2, co = k & 3;
    float v = 0.0f;
    if (tap < 9 && co < a.cout) v = a.w[(co * 32 + ci) * 9 + tap];
    reinterpret_cast<__bf16*>(wl)[i] = (__bf16)v;
  }
  __syncthreads();
  if (x0 >= a.W) return;
  bf16x8 wf[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int c = 0; c < 2; ++c) wf[st][c] = wl[(st * 2 + c) * 64 + lane];
  for (int y = y0; y < y1; ++y) {
    const int x = x0 + px;
    // step 0: taps 2g, 2g + 1 (tap = 3 ky + kx reads pixel + (1 - ky, 1 - kx))
    const int t0 = 2 * g, t1 = 2 * g + 1;
    const f32x4 z0 = pred_gz(a, n, y + 1 - t0 / 3, x + 1 - t0 % 3);
    const f32x4 z1 = pred_gz(a, n, y + 1 - t1 / 3, x + 1 - t1 % 3);
    f32x4 z8 = {0.f, 0.f, 0.f, 0.f};
    if (g == 0) z8 = pred_gz(a, n, y - 1, x - 1);  // tap 8 = (2, 2)
    const bf16x8 b0 = {(__bf16)z0[0], (__bf16)z0[1], (__bf16)z0[2], (__bf16)z0[3],
                       (__bf16)z1[0], (__bf16)z1[1], (__bf16)z1[2], (__bf16)z1[3]};
    const bf16x8 b1 = {(__bf16)z8[0], (__bf16)z8[1], (__bf16)z8[2], (__bf16)z8[3],
                       (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    f32x4 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][c], b0, acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][c], b1, acc[c], 0, 0, 0);
    }
    const size_t pix = ((size_t)n * a.H + y) * a.W + x;
    __bf16* const o = reinterpret_cast<__bf16*>(a.out) + pix * 32 + 4 * g;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x4 r = {(__bf16)acc[c][0], (__bf16)acc[c][1], (__bf16)acc[c][2],
                        (__bf16)acc[c][3]};
      *reinterpret_cast<bf16x4*>(o + 16 * c) = r;
    }
  }
}

// Weight and bias gradient of the head: gW[co][ci][tap] = sum over pixels of
// gz[p][co] * x[p + tap][ci], gb[co] = sum gz[p][co] -- a GEMM with K = all
// pixels: the weight-gradient kernel of lsi_conv_wgrad.hip (operands through the
// LDS transpose read) with gz formed while a row is staged.  Four outputs fill
// 4 of the matrix tile's 16 rows; rows 4 .. 7 carry what rounding gz to bf16
// left (hi + lo: ~16 mantissa bits; MIOpen's path rounds gz to bf16 once), the
// bias gradient is one more product with a fragment of ones.  History at
// 8 x 256 x 768 (tools/time_pred_bwd.py): on the vector unit with fp32 atomics
// into the result 496 us (1.5 M atomics onto 1156 addresses); one tap per wave,
// per-workgroup partial sums + reduce kernel, DPP gz exchange, four-row prefetch
// through buffer loads 132 us; on the matrix cores: see DESIGN.md 4.6.

int launch_rc() { return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH; }

}  // namespace

// (lsi_conv_wgrad.hip)
size_t lsi_pred_wgrad_workspace_bytes(int N, int H, int W, int cout);
int lsi_pred_wgrad_launch(int N, int H, int W, int cout, const float* g, const float* y,
                          const void* x, float* g_wb, void* workspace, size_t workspace_bytes,
                          hipStream_t stream);

extern "C" int lsi_conv3x3_c32_fwd(int32_t N, int32_t H, int32_t W, int32_t cout,
                                   int32_t mode, const void* x, const float* weight,
                                   const float* bias, float scale3, void* out,
                                   lsi_stream_t stream) {
  const int pred = mode == 1, tr = mode == 2;
  if (mode < 0 || mode > 2) return LSI_EINVAL;
  if (N <= 0 || H <= 0 || W <= 0 || W % 16 != 0 || N > 65535) return LSI_EINVAL;
  if (pred ? (cout < 1 || cout > 4) : (tr ? cout != 32 : (cout != 16 && cout != 32)))
    return LSI_EINVAL;
  if (!x || !weight || !out) return LSI_ENULL;
  if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return LSI_EINVAL;
  // lsi_conv3x3_pred_bwd forms sigmoid' = y (1 - y) from the stored output: a
  // scaled channel 3 would give wrong gradients (the caller scales afterwards)
  if (pred && scale3 != 1.0f) return LSI_EINVAL;
  ConvArgs a;
  a.x = reinterpret_cast<const __bf16*>(x);
  a.w = weight;
  a.tr = tr;
  a.bias = bias;
  a.out = out;
  a.N = N; a.H = H; a.W = W; a.cout = cout;
  a.scale3 = scale3;
  const dim3 grid((W / 16 + 3) / 4, (H + CV_ROWS - 1) / CV_ROWS, N), block(256);
  if (pred)
    hipLaunchKernelGGL((conv3x3_c32_kernel<1, true>), grid, block, 0, (hipStream_t)stream, a);
  else if (cout == 16)
    hipLaunchKernelGGL((conv3x3_c32_kernel<1, false>), grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv3x3_c32_kernel<2, false>), grid, block, 0, (hipStream_t)stream, a);
  return launch_rc();
}

extern "C" size_t lsi_conv3x3_pred_bwd_workspace_bytes(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return lsi_pred_wgrad_workspace_bytes(N, H, W, 4);
}

extern "C" int lsi_conv3x3_pred_bwd(int32_t N, int32_t H, int32_t W, int32_t cout,
                                    const float* g, const float* y, const void* x,
                                    const float* weight, void* g_x, float* g_wb,
                                    void* workspace, size_t workspace_bytes,
                                    lsi_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || W % 16 != 0 || N > 65535 || cout < 1 || cout > 4)
    return LSI_EINVAL;
  if (!g || !y || !weight || (!g_x && !g_wb) || (g_wb && (!x || !workspace))) return LSI_ENULL;
  if (((uintptr_t)g & 15) || ((uintptr_t)y & 15) || ((uintptr_t)x & 15) ||
      ((uintptr_t)g_x & 15) || ((uintptr_t)workspace & 15))
    return LSI_EINVAL;
  if (g_wb && workspace_bytes < lsi_pred_wgrad_workspace_bytes(N, H, W, cout))
    return LSI_EWORKSPACE;
  if (g_x) {
    PredBwdArgs a;
    a.g = g; a.y = y; a.w = weight; a.x = reinterpret_cast<const __bf16*>(x);
    a.N = N; a.H = H; a.W = W; a.cout = cout;
    a.out = g_x;
    const dim3 grid((W / 16 + 3) / 4, (H + CV_ROWS - 1) / CV_ROWS, N), block(256);
    hipLaunchKernelGGL(pred_bwd_data_kernel, grid, block, 0, (hipStream_t)stream, a);
    if (launch_rc() != LSI_OK) return LSI_ELAUNCH;
  }
  if (g_wb)
    return lsi_pred_wgrad_launch(N, H, W, cout, g, y, x, g_wb, workspace, workspace_bytes,
                                 (hipStream_t)stream);
  return LSI_OK;
}
