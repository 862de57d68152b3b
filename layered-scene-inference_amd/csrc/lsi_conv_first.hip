// The network's FIRST convolution on the matrix cores (gfx950,
// v_mfma_f32_16x16x32_bf16, fp32 accumulation): reference nets.py:273 `cnv1`
// (and :53 in encoder_simple) -- slim.conv2d(inp_img, 32, [7, 7], stride=2), TF
// `SAME` padding, batch norm + ReLU behind it -- forward and weight gradient.
// (Its input is the image: there is no data gradient.)
//
// Why its own kernel.  With 3 input channels the implicit-GEMM kernel's K (32
// consecutive channels of one pixel, lsi_conv_igemm.hip) does not exist.  Here a
// pixel is padded to 4 bf16 channels in LDS (8 bytes) and the K of one MFMA is a
// whole kernel ROW: k = 4 kx + c, 7 taps x 4 channels = 28 of 32 (the weights
// of k >= 28 and of c = 3 are zero).  With stride 2 the B fragment of output
// pixel j and lane group kg -- taps kx = 2 kg, 2 kg + 1 -- is the two
// neighbouring staged pixels 2 (j + kg), 2 (j + kg) + 1: ONE aligned 16-byte LDS
// read, no im2col.  Seven MFMAs (one per ky) per 16 pixels x 16 channels.
//
// The layer is 0.46 GFLOP per 256 x 768 image against 2.4 MB read + 3.1 MB
// written: memory-bound by a wide margin (the roofline here is HBM: 5.5 MB per
// image at ~5 TB/s = 1.1 us); what this kernel has to do is read the image
// once, write the activation once and leave the batch-norm sums behind
// (lsi_conv2d_*_bnstats protocol), instead of a cast kernel + the library's
// generic path + a statistics pass.
//
// Weight gradient: gW[co][c][ky][kx] = sum over output pixels of
// gy[n][i][j][co] * x[n][2 i + ky - pt][2 j + kx - pl][c] -- a GEMM with K =
// output pixels (393 k per 8 images), M = 32, N = 7 x 28.  Both operands need 8
// CONSECUTIVE PIXELS per lane: the image is staged de-interleaved by column
// parity and channel (planes [parity][c][row][col / 2]: consecutive output
// pixels then read consecutive elements for any kx), gy pixel-major with a
// 68-byte pitch; fragments are gathered with 2-byte LDS reads (112 + 16 per 28
// MFMAs: at 0.46 GFLOP the LDS pipe has time).  A wave keeps the whole 32 x 224
// gradient in 28 accumulator tiles; workgroups walk the tiles persistently, add
// their four waves through LDS and write one partial per workgroup; a second
// kernel folds the partials in a fixed order (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lsi_hip.h"
#include "lsi_splat_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int KH = 7, KW = 7, ST = 2;       // the one geometry the networks have
constexpr int F_TH = 8, F_TW = 64;          // forward tile: output rows x columns
constexpr int F_PH = (F_TH - 1) * ST + KH;  // 21 staged rows
constexpr int F_PW = (F_TW - 1) * ST + KW;  // 133 staged pixels per row
constexpr int F_PWP = 136;                  // row pitch in pixels (8 bytes each)

struct FirstArgs {
  const void* x;      // N x H x W x Cin, fp32 or bf16 (channels innermost)
  const float* w;     // the fp32 parameter, element strides below
  __bf16* out;        // N x OH x OW x 32
  int N, H, W, Cin, OH, OW;
  int pad_t, pad_l;
  long wsco, wsc, wsky, wskx;
  float* st_ws;       // batch-norm accumulators (NULL: none), as lsi_conv_igemm.hip
  int st_groups, st_ns;
};

__device__ __forceinline__ unsigned short bf16_bits(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}

__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}

// One input pixel's channels as 4 bf16 (zero outside the image / past Cin).
template <bool XBF16>
__device__ __forceinline__ u32x2 load_pixel4(const void* x, long pix, int cin, bool ok) {
  unsigned short c[4] = {0, 0, 0, 0};
  if (ok) {
    if (XBF16) {
      const unsigned short* p = static_cast<const unsigned short*>(x) + pix * cin;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < cin) c[k] = p[k];
    } else {
      const float* p = static_cast<const float*>(x) + pix * cin;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < cin) c[k] = bf16_bits(p[k]);   // (round to nearest even: what autocast's cast does)
    }
  }
  u32x2 r;
  r[0] = (unsigned)c[0] | ((unsigned)c[1] << 16);
  r[1] = (unsigned)c[2] | ((unsigned)c[3] << 16);
  return r;
}

template <bool XBF16>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(FirstArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[F_PH * F_PWP * 8];
  __shared__ float red[4 * 2 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pxl = lane & 15, kg = lane >> 4;
  const int n = blockIdx.z, i0 = blockIdx.y * F_TH, j0 = blockIdx.x * F_TW;
  // ---- the weights as A fragments, straight from the fp32 parameter -----------
  // af[ky][ct]: output channel 16 ct + pxl, k = 8 kg + e = 4 kx + c
  bf16x8 af[KH][2];
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float* wr = a.w + (long)(16 * ct + pxl) * a.wsco + (long)ky * a.wsky;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kx = 2 * kg + (e >> 2), c = e & 3;
        float v = 0.0f;
        if (kx < KW && c < a.Cin) v = wr[(long)c * a.wsc + (long)kx * a.wskx];
        af[ky][ct][e] = (__bf16)v;
      }
    }
  // ---- the image patch: 21 rows x 133 pixels, 4 bf16 per pixel ---------------
  {
    const int iy0 = i0 * ST - a.pad_t, ix0 = j0 * ST - a.pad_l;
    for (int idx = tid; idx < F_PH * F_PWP; idx += 256) {
      const int py = idx / F_PWP, px = idx - py * F_PWP;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = px < F_PW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const u32x2 v = load_pixel4<XBF16>(a.x, ((long)n * a.H + iy) * a.W + ix, a.Cin, ok);
      *reinterpret_cast<u32x2*>(patch + (size_t)idx * 8) = v;
    }
  }
  __syncthreads();
  f32x4 acc[F_TH][2];
#pragma unroll
  for (int r = 0; r < F_TH; ++r) { acc[r][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[r][1] = acc[r][0]; }
  // lane's fragment of staged row p: pixels 2 (16 wave + pxl + kg), + 1
  const unsigned char* const bp = patch + (size_t)(2 * (16 * wave + pxl + kg)) * 8;
#pragma unroll
  for (int p = 0; p < F_PH; ++p) {
    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bp + (size_t)p * F_PWP * 8);
#pragma unroll
    for (int r = 0; r < F_TH; ++r) {
      const int ky = p - ST * r;       // (compile time: both loops are unrolled)
      if (ky >= 0 && ky < KH) {
        acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ky][0], bf, acc[r][0], 0, 0, 0);
        acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ky][1], bf, acc[r][1], 0, 0, 0);
      }
    }
  }
  const int j = j0 + 16 * wave + pxl;
  if (a.st_ws) {
    // batch-norm sums of the ROUNDED outputs (lsi_conv_igemm.hip's protocol)
    float ss[2][4], qq[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { ss[c][e] = 0.f; qq[c][e] = 0.f; }
#pragma unroll
    for (int r = 0; r < F_TH; ++r) {
      if (j < a.OW && i0 + r < a.OH) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = (float)(__bf16)acc[r][c][e];
            ss[c][e] += v;
            qq[c][e] = __builtin_fmaf(v, v, qq[c][e]);
          }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { ss[c][e] = row16_sum(ss[c][e]); qq[c][e] = row16_sum(qq[c][e]); }
    if (pxl == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wave * 2 + 0) * 32 + 16 * c + 4 * kg + e] = ss[c][e];
          red[(wave * 2 + 1) * 32 + 16 * c + 4 * kg + e] = qq[c][e];
        }
    }
    __syncthreads();
    const int per_grp = a.N / a.st_groups, grp = n / per_grp;
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && n == grp * per_grp)
      __hip_atomic_store(reinterpret_cast<int*>(a.st_ws + (size_t)grp * LSI_BN_WS_STRIDE) +
                             LSI_BN_WS_TAG,
                         LSI_BN_TAG(32, a.st_groups), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
      const int q = tid >> 5, ch = tid & 31;
      const float v = (red[(0 * 2 + q) * 32 + ch] + red[(1 * 2 + q) * 32 + ch]) +
                      (red[(2 * 2 + q) * 32 + ch] + red[(3 * 2 + q) * 32 + ch]);
      const int f = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
      __hip_atomic_fetch_add(a.st_ws + (size_t)grp * LSI_BN_WS_STRIDE + LSI_BN_WS_ACC +
                                 (f % a.st_ns) * 2 * 32 + q * 32 + ch,
                             v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (j < a.OW) {
#pragma unroll
    for (int r = 0; r < F_TH; ++r) {
      const int i = i0 + r;
      if (i < a.OH) {
        __bf16* const o = a.out + (((size_t)n * a.OH + i) * a.OW + j) * 32 + 4 * kg;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bf16x4 v;
          v[0] = (__bf16)acc[r][c][0]; v[1] = (__bf16)acc[r][c][1];
          v[2] = (__bf16)acc[r][c][2]; v[3] = (__bf16)acc[r][c][3];
          *reinterpret_cast<bf16x4*>(o + 16 * c) = v;
        }
      }
    }
  }
}

// ---- weight gradient --------------------------------------------------------------
constexpr int G_TR = 4, G_TC = 128;                 // tile: output rows (one per wave) x columns
constexpr int G_PR = (G_TR - 1) * ST + KH;          // 13 staged image rows
constexpr int G_PQ = (G_TC - 1) * ST + KW;          // 261 staged image columns
constexpr int G_PWH = 132;                          // columns per parity plane (131 used)
constexpr int G_GP = 34;                            // gy pixel pitch in bf16 (68 bytes)
constexpr int G_NACC = KH * 32 * 32;                // partial: [ky][n = 4 kx + c][co]
constexpr size_t G_XS_BYTES = (size_t)2 * 4 * G_PR * G_PWH * 2;
constexpr size_t G_GS_BYTES = (size_t)G_TR * G_TC * G_GP * 2;
constexpr size_t G_LDS = G_XS_BYTES + G_GS_BYTES;   // 27456 + 34816 = 62272 bytes

struct FirstGwArgs {
  const void* x;
  const __bf16* gy;    // N x OH x OW x 32
  float* part;         // [gridDim.x][G_NACC]
  int N, H, W, Cin, OH, OW;
  int pad_t, pad_l;
  int tiles_y, tiles_x, ntiles;
};

template <bool XBF16>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(FirstGwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];
  unsigned short* const xs = reinterpret_cast<unsigned short*>(g_smem);
  unsigned short* const gs = reinterpret_cast<unsigned short*>(g_smem + G_XS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pxl = lane & 15, kg = lane >> 4;
  f32x4 acc[KH][2][2];   // [ky][nt: 16 columns n = 4 kx + c][ct: 16 channels]
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      acc[ky][nt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[ky][nt][1] = acc[ky][nt][0];
    }
  // this lane's B column n = 16 nt + pxl = 4 kx + c (clamped into the kernel: the
  // products of the clamped columns land in entries nobody reads)
  int bofs[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int nn = 16 * nt + pxl;
    const int kx = min(nn >> 2, KW - 1), c = min(nn & 3, a.Cin - 1);
    bofs[nt] = (((kx & 1) * 4 + c) * G_PR) * G_PWH + (kx >> 1);
  }
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int n = t / (a.tiles_y * a.tiles_x), rem = t - n * (a.tiles_y * a.tiles_x);
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int i0 = ty * G_TR, j0 = tx * G_TC;
    __syncthreads();   // (the previous tile's fragments have been read)
    // ---- the image rows, de-interleaved: xs[parity][c][row][col / 2] ----------
    {
      const int iy0 = i0 * ST - a.pad_t, ix0 = j0 * ST - a.pad_l;
      for (int idx = tid; idx < G_PR * G_PQ; idx += 256) {
        const int py = idx / G_PQ, q = idx - py * G_PQ;
        const int iy = iy0 + py, ix = ix0 + q;
        const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const u32x2 v = load_pixel4<XBF16>(a.x, ((long)n * a.H + iy) * a.W + ix, a.Cin, ok);
        unsigned short* const d = xs + (((q & 1) * 4) * G_PR + py) * G_PWH + (q >> 1);
        d[0 * G_PR * G_PWH] = (unsigned short)(v[0] & 0xffffu);
        d[1 * G_PR * G_PWH] = (unsigned short)(v[0] >> 16);
        d[2 * G_PR * G_PWH] = (unsigned short)(v[1] & 0xffffu);
        d[3 * G_PR * G_PWH] = (unsigned short)(v[1] >> 16);
      }
    }
    // ---- gy: 4 rows x 128 pixels x 32 channels, 68 bytes per pixel ------------
    for (int idx = tid; idx < G_TR * G_TC * 4; idx += 256) {
      const int q = idx & 3, p = (idx >> 2) & (G_TC - 1), r = idx >> 9;
      const int i = i0 + r, j = j0 + p;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (i < a.OH && j < a.OW)
        v = *reinterpret_cast<const u32x4*>(a.gy + (((size_t)n * a.OH + i) * a.OW + j) * 32 + 8 * q);
      unsigned* const d = reinterpret_cast<unsigned*>(gs + (size_t)(r * G_TC + p) * G_GP + 8 * q);
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    // ---- wave = output row i0 + wave; four K steps of 32 pixels ------------------
#pragma unroll 1
    for (int seg = 0; seg < G_TC / 32; ++seg) {
      const int p0 = seg * 32 + 8 * kg;     // this lane's 8 pixels
      bf16x8 af[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const unsigned short* g = gs + (size_t)(wave * G_TC + p0) * G_GP + 16 * ct + pxl;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          af[ct][e] = __builtin_bit_cast(__bf16, g[e * G_GP]);
      }
#pragma unroll
      for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const unsigned short* xr = xs + bofs[nt] + (ST * wave + ky) * G_PWH + p0;
          bf16x8 bf;
#pragma unroll
          for (int e = 0; e < 8; ++e) bf[e] = __builtin_bit_cast(__bf16, xr[e]);
          acc[ky][nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf, acc[ky][nt][0], 0, 0, 0);
          acc[ky][nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf, acc[ky][nt][1], 0, 0, 0);
        }
      }
    }
  }
  // ---- the four waves' sums through LDS, one partial per workgroup --------------
  // D of (ky, nt, ct): lane holds rows co = 16 ct + 4 kg + e, column n = 16 nt + pxl
  float* const red = reinterpret_cast<float*>(g_smem);   // [ky][n][co]: 28 KB
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            float* const d = red + ((ky * 32 + 16 * nt + pxl) * 32 + 16 * ct + 4 * kg);
            f32x4 v = acc[ky][nt][ct];
            if (w > 0) {
              const f32x4 o = *reinterpret_cast<const f32x4*>(d);
              v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            *reinterpret_cast<f32x4*>(d) = v;
          }
    }
  }
  __syncthreads();
  float* const out = a.part + (size_t)blockIdx.x * G_NACC;
  for (int idx = tid; idx < G_NACC / 4; idx += 256)
    reinterpret_cast<f32x4*>(out)[idx] = reinterpret_cast<const f32x4*>(red)[idx];
}

// gW[co][c][ky][kx] = sum over the partials of [ky][4 kx + c][co], in a fixed order.
// Block = 64 entries x 16 groups of partials (group y sums partials y, y + 16,
// ... four at a time, the 16 group sums are added in order through LDS): 112
// workgroups of 8 - 32 rounds of loads.  (One thread per entry walking all the
// partials was 28 workgroups x ~128 dependent rounds: 44 us for a 4704-float
// gradient.)
__global__ __launch_bounds__(1024) void conv_first_wgrad_fold_kernel(
    const float* __restrict__ part, int nparts, float* __restrict__ gw, int cin, long wsco,
    long wsc, long wsky, long wskx) {
  __shared__ float red[16][64];
  const int idx = blockIdx.x * 64 + threadIdx.x, y = threadIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < G_NACC) {
    int b = y;
    for (; b + 48 < nparts; b += 64) {
      s0 += part[(size_t)(b + 0) * G_NACC + idx];
      s1 += part[(size_t)(b + 16) * G_NACC + idx];
      s2 += part[(size_t)(b + 32) * G_NACC + idx];
      s3 += part[(size_t)(b + 48) * G_NACC + idx];
    }
    for (; b < nparts; b += 16) s0 += part[(size_t)b * G_NACC + idx];
  }
  red[y][threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (y != 0 || idx >= G_NACC) return;
  const int co = idx & 31, nn = (idx >> 5) & 31, ky = idx >> 10;
  const int kx = nn >> 2, c = nn & 3;
  if (kx >= KW || c >= cin) return;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
  gw[co * wsco + c * wsc + ky * wsky + kx * wskx] = v;
}

bool first_ok(const LsiConvDesc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0) return false;
  if (d->Cin < 1 || d->Cin > 4 || d->Cout != 32) return false;
  if (d->KH != KH || d->KW != KW || d->stride != ST) return false;
  if (d->pad_t < 0 || d->pad_l < 0 || d->pad_t >= KH || d->pad_l >= KW) return false;
  if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31)) return false;
  if ((int64_t)d->N * d->OH * d->OW * d->Cout >= (1ll << 31)) return false;
  if (d->N > 65535) return false;
  return true;
}

// element strides of the parameter Cout x Cin x KH x KW: contiguous (0) or
// torch's channels-last strides (2: Cout x KH x KW x Cin in memory)
void weight_strides(const LsiConvDesc* d, int layout, long* sco, long* sc, long* sky, long* skx) {
  if (layout == 2) {
    *sc = 1; *skx = d->Cin; *sky = (long)d->KW * d->Cin; *sco = (long)d->KH * d->KW * d->Cin;
  } else {
    *skx = 1; *sky = d->KW; *sc = (long)d->KH * d->KW; *sco = (long)d->Cin * d->KH * d->KW;
  }
}

int gw_grid(const LsiConvDesc* d, int* tiles_y, int* tiles_x) {
  *tiles_y = (d->OH + G_TR - 1) / G_TR;
  *tiles_x = (d->OW + G_TC - 1) / G_TC;
  const long nt = (long)d->N * *tiles_y * *tiles_x;
  return (int)(nt < 512 ? nt : 512);   // two resident workgroups per CU
}

}  // namespace

extern "C" int lsi_conv2d_first_supported(const LsiConvDesc* d) { return first_ok(d) ? 1 : 0; }

extern "C" int lsi_conv2d_first_fwd(const LsiConvDesc* d, const void* x, int32_t x_bf16,
                                    const float* weight, int32_t weight_layout, void* out,
                                    float* bn_workspace, int32_t groups, lsi_stream_t stream_) {
  if (!d || !x || !weight || !out) return LSI_ENULL;
  if (!first_ok(d)) return LSI_EUNSUPPORTED;
  if (weight_layout != 0 && weight_layout != 2) return LSI_EINVAL;
  if (((uintptr_t)out & 7) || ((uintptr_t)x & (x_bf16 ? 1 : 3))) return LSI_EINVAL;
  FirstArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = weight; a.out = (__bf16*)out;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.OH = d->OH; a.OW = d->OW;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  weight_strides(d, weight_layout, &a.wsco, &a.wsc, &a.wsky, &a.wskx);
  if (bn_workspace) {
    if (groups < 1 || d->N % groups) return LSI_EINVAL;
    a.st_ws = bn_workspace; a.st_groups = groups; a.st_ns = lsi_bn_stat_slots(32);
  }
  const dim3 grid((d->OW + F_TW - 1) / F_TW, (d->OH + F_TH - 1) / F_TH, d->N);
  if (grid.y > 65535) return LSI_EINVAL;
  if (x_bf16)
    hipLaunchKernelGGL(conv_first_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  else
    hipLaunchKernelGGL(conv_first_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream_, a);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

extern "C" size_t lsi_conv2d_first_wgrad_workspace_bytes(const LsiConvDesc* d) {
  if (!first_ok(d)) return 0;
  int ty, tx;
  return (size_t)gw_grid(d, &ty, &tx) * G_NACC * sizeof(float);
}

extern "C" int lsi_conv2d_first_wgrad(const LsiConvDesc* d, const void* x, int32_t x_bf16,
                                      const void* gy, float* g_weight, int32_t weight_layout,
                                      void* workspace, size_t workspace_bytes,
                                      lsi_stream_t stream_) {
  if (!d || !x || !gy || !g_weight || !workspace) return LSI_ENULL;
  if (!first_ok(d)) return LSI_EUNSUPPORTED;
  if (weight_layout != 0 && weight_layout != 2) return LSI_EINVAL;
  if (((uintptr_t)gy & 15) || ((uintptr_t)workspace & 15) || ((uintptr_t)x & (x_bf16 ? 1 : 3)))
    return LSI_EINVAL;
  if (workspace_bytes < lsi_conv2d_first_wgrad_workspace_bytes(d)) return LSI_EWORKSPACE;
  FirstGwArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.gy = (const __bf16*)gy; a.part = (float*)workspace;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.OH = d->OH; a.OW = d->OW;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  const int nwg = gw_grid(d, &a.tiles_y, &a.tiles_x);
  a.ntiles = d->N * a.tiles_y * a.tiles_x;
  hipStream_t st = (hipStream_t)stream_;
  const void* fn = x_bf16 ? (const void*)conv_first_wgrad_kernel<true>
                          : (const void*)conv_first_wgrad_kernel<false>;
  if (lsi_ensure_dynamic_lds(fn, G_LDS) != LSI_OK) return LSI_ELAUNCH;
  void* kargs[1] = {&a};
  if (hipLaunchKernel(fn, dim3(nwg), dim3(256), kargs, G_LDS, st) != hipSuccess) return LSI_ELAUNCH;
  long sco, sc, sky, skx;
  weight_strides(d, weight_layout, &sco, &sc, &sky, &skx);
  hipLaunchKernelGGL(conv_first_wgrad_fold_kernel, dim3((G_NACC + 63) / 64), dim3(64, 16), 0, st,
                     (const float*)workspace, nwg, g_weight, d->Cin, sco, sc, sky, skx);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
