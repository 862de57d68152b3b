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
  const __bf16* x;   // N x H x W x 32 (weight gradient)
  void* out;         // data: bf16 N x H x W x 32; weight: fp32 [cout*288 + cout] (+=)
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
// gz[p][co] * x[p + tap][ci], gb[co] = sum gz[p][co] -- a reduction over all
// N H W pixels into 4 x 288 + 4 numbers.  N = 4 output channels leaves nothing
// for a matrix tile (and both MFMA operands would want 8 consecutive PIXELS of
// one channel, the transpose of the channels-last layout); on the vector unit:
// the nine taps are the workgroup's nine waves, lane (pixel slot = lane / 4,
// channel block = lane % 4) holds 8 input channels of its shifted pixel (one
// 16-byte load) and ONE channel of the pixel's gz (the four lanes of a pixel
// exchange them with DPP quad permutes): 32 multiply-adds as 16 v_pk_fma_f32
// into 32 accumulators.  Workgroups loop over strips of 16 columns x 32 rows,
// four rows multiplied while the next four are in flight; at
// the end the 16 pixel slots are summed with shuffles and every workgroup
// writes its 4 x 288 + 4 partial sums to the workspace.  A first version added
// them to the result with fp32 atomics: 1.5 M atomics onto 1156 addresses cost
// 350 - 500 us (tools/time_pred_bwd.py); pred_bwd_reduce_kernel sums the
// partials instead.  gz stays fp32 (MIOpen's path rounds it to bf16 first).
constexpr int PW_STRIDE = 1160;  // floats per workgroup in the workspace (>= 4 * 288 + 4)
constexpr int PW_MAXWG = 424;    // strip sets: x 9 waves ~ 15 per CU (110 VGPRs: 4 per SIMD), a multiple of 8

// One item of the weight kernel: four rows of a strip, as loaded by one lane --
// its channel of g and y (the four lanes of a pixel exchange the products) and 8
// input channels of the shifted pixel.
struct PredRows {
  float g[4], y[4];
  bf16x8 v[4];
  bool m[4];
};

__device__ __forceinline__ float quad_bcast(float v, int k) {
  const int i = __builtin_bit_cast(int, v);
  int r;
  switch (k) {  // quad_perm(k, k, k, k)
    case 0: r = __builtin_amdgcn_mov_dpp(i, 0x00, 0xf, 0xf, false); break;
    case 1: r = __builtin_amdgcn_mov_dpp(i, 0x55, 0xf, 0xf, false); break;
    case 2: r = __builtin_amdgcn_mov_dpp(i, 0xaa, 0xf, 0xf, false); break;
    default: r = __builtin_amdgcn_mov_dpp(i, 0xff, 0xf, 0xf, false); break;
  }
  return __builtin_bit_cast(float, r);
}

__global__ __launch_bounds__(192) void pred_bwd_weight_kernel(PredBwdArgs a, int nstrip,
                                                              float* part) {
  const int lane = threadIdx.x & 63;
  // workgroup = the three taps of one kernel row over one set of strips; the
  // three rows of a set are workgroups b, b + 8, b + 16: the same XCD (b % 8),
  // dispatched together -- their re-reads of x are L2 hits
  const int wset = (blockIdx.x / 24) * 8 + (blockIdx.x & 7), nset = gridDim.x / 3;
  const int tap = __builtin_amdgcn_readfirstlane(3 * ((blockIdx.x >> 3) % 3) + (threadIdx.x >> 6));
  const int ps = lane >> 2, cb = lane & 3;
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  f32x2 acc[8][2];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c][0] = acc[c][1] = f32x2{0.f, 0.f};
  f32x2 gb[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
  const int sx = a.W / 16, sy = (a.H + CV_ROWS - 1) / CV_ROWS;
  // items = (strip of this workgroup, group of four rows); the loads of item
  // i + 1 are in flight while item i is multiplied
  constexpr int GPS = CV_ROWS / 4;
  const int nitem = wset < nstrip ? GPS * ((nstrip - wset + nset - 1) / nset) : 0;
  // Branch-free buffer loads from clamped addresses (row offsets in scalar
  // registers, 32-bit lane offsets: no 64-bit address per load in flight); what lies outside the image -- the row of
  // gz, the shifted row or column of x, channels >= cout -- is removed by
  // zeroing gz (`m`), which is exact: the bias sum is taken by the centre tap,
  // whose shifted pixel is the pixel itself.
  const unsigned npix = (unsigned)a.N * a.H * a.W;
  const __amdgpu_buffer_rsrc_t rg_ =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, npix * 16u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry_ =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.y), 0, npix * 16u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx_ =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, npix * 64u, 0x00020000);
  auto load_item = [&](int it_, PredRows& p) {
    const int it = min(it_, nitem - 1);  // (the last prefetch is never used)
    const int sidx = wset + (it / GPS) * nset;
    // (integer division runs on the vector unit: pin the results to scalars)
    const int n = __builtin_amdgcn_readfirstlane(sidx / (sx * sy)), r = sidx - n * sx * sy;
    const int ry = __builtin_amdgcn_readfirstlane(r / sx);
    const int yb = ry * CV_ROWS + 4 * (it % GPS);
    const int x = (r - ry * sx) * 16 + ps, xx = x + dx;
    const bool lin = xx >= 0 && xx < a.W && cb < a.cout;
    const unsigned poff = 4u * (4 * x + min(cb, a.cout - 1));  // bytes
    const unsigned xoff = 2u * (32 * min(max(xx, 0), a.W - 1) + 8 * cb);
    const int row0 = n * a.H;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int yu = yb + u, yy = yu + dy;
      // byte offsets of the rows (< 4 GiB: checked by the launcher), scalar
      const unsigned rg = __builtin_amdgcn_readfirstlane(
          (unsigned)(row0 + min(yu, a.H - 1)) * (unsigned)a.W * 16u);
      const unsigned rx = __builtin_amdgcn_readfirstlane(
          (unsigned)(row0 + min(max(yy, 0), a.H - 1)) * (unsigned)a.W * 64u);
      p.g[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg_, poff, rg, 0));
      p.y[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, poff, rg, 0));
      p.v[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx_, xoff, rx, 0));
      p.m[u] = lin && yu < a.H && yy >= 0 && yy < a.H;
    }
  };
  auto mul_item = [&](const PredRows& p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float gz = p.m[u] ? p.g[u] * p.y[u] * (1.0f - p.y[u]) : 0.0f;
      const f32x2 zl = {quad_bcast(gz, 0), quad_bcast(gz, 1)};
      const f32x2 zh = {quad_bcast(gz, 2), quad_bcast(gz, 3)};
      gb[0] += zl;
      gb[1] += zh;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float xv = (float)p.v[u][c];
        const f32x2 x2 = {xv, xv};
        acc[c][0] = __builtin_elementwise_fma(zl, x2, acc[c][0]);
        acc[c][1] = __builtin_elementwise_fma(zh, x2, acc[c][1]);
      }
    }
  };
  if (nitem > 0) {  // (a set past the last strip writes zeros)
    PredRows pa, pb;
    load_item(0, pa);
    for (int it = 0; it < nitem; it += 2) {
      load_item(it + 1, pb);
      mul_item(pa);
      load_item(it + 2, pa);
      mul_item(pb);
    }
  }
  // sum over the 16 pixel slots (lanes with equal lane % 4); lane cb of the wave
  // writes its 8 channels x 4 outputs: part[(co * 32 + ci) * 9 + tap]
  float* const out = part + (size_t)wset * PW_STRIDE;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = acc[c][k >> 1][k & 1];
      for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if (ps == 0 && k < a.cout) out[(k * 32 + 8 * cb + c) * 9 + tap] = v;
    }
  if (tap == 4) {  // the bias gradient: every pixel once (each of its 4 lanes holds it)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = cb == 0 ? gb[k >> 1][k & 1] : 0.0f;
      for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if (lane == 0 && k < a.cout) out[a.cout * 288 + k] = v;
    }
  }
}

// g_wb[o] = sum over the workgroups' partials: 64 outputs x 16 row groups per
// workgroup, the groups folded through LDS.  Writes (the caller need not clear).
__global__ __launch_bounds__(1024) void pred_bwd_reduce_kernel(const float* part, int nwg,
                                                               int nout, float* out) {
  __shared__ float red[16][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  float s = 0.0f;
  if (o < nout)
    for (int w = grp; w < nwg; w += 16) s += part[(size_t)w * PW_STRIDE + o];
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && o < nout) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
    out[o] = t;
  }
}

int launch_rc() { return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH; }

}  // namespace

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

extern "C" size_t lsi_conv3x3_pred_bwd_workspace_bytes(void) {
  return (size_t)PW_MAXWG * PW_STRIDE * sizeof(float);
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
      ((uintptr_t)g_x & 15) || ((uintptr_t)workspace & 3))
    return LSI_EINVAL;
  if (g_wb && workspace_bytes < lsi_conv3x3_pred_bwd_workspace_bytes()) return LSI_EWORKSPACE;
  if (g_wb && (uint64_t)N * H * W * 64 >= (1ull << 32)) return LSI_EUNSUPPORTED;  // 32-bit offsets
  PredBwdArgs a;
  a.g = g; a.y = y; a.w = weight; a.x = reinterpret_cast<const __bf16*>(x);
  a.N = N; a.H = H; a.W = W; a.cout = cout;
  if (g_x) {
    a.out = g_x;
    const dim3 grid((W / 16 + 3) / 4, (H + CV_ROWS - 1) / CV_ROWS, N), block(256);
    hipLaunchKernelGGL(pred_bwd_data_kernel, grid, block, 0, (hipStream_t)stream, a);
    if (launch_rc() != LSI_OK) return LSI_ELAUNCH;
  }
  if (g_wb) {
    a.out = nullptr;
    const int nstrip = N * (W / 16) * ((H + CV_ROWS - 1) / CV_ROWS);
    // equal shares: sets of ceil(nstrip / PW_MAXWG) strips, rounded up to whole
    // groups of 8 sets (sets past the last strip write zero partials)
    const int per = (nstrip + PW_MAXWG - 1) / PW_MAXWG;
    const int nwg = (((nstrip + per - 1) / per) + 7) / 8 * 8;
    const int nout = cout * 288 + cout;
    float* const part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(pred_bwd_weight_kernel, dim3(3 * nwg), dim3(192), 0,
                       (hipStream_t)stream, a, nstrip, part);
    if (launch_rc() != LSI_OK) return LSI_ELAUNCH;
    hipLaunchKernelGGL(pred_bwd_reduce_kernel, dim3((nout + 63) / 64), dim3(1024), 0,
                       (hipStream_t)stream, part, nwg, nout, g_wb);
    if (launch_rc() != LSI_OK) return LSI_ELAUNCH;
  }
  return LSI_OK;
}
