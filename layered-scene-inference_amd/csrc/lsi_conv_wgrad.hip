// Weight gradient of the 3x3 stride-1 SAME convolutions of the LDI heads
// (reference nets.py:104-111: `upcnv*b`; TF autodiff of slim.conv2d) on the
// matrix cores (gfx950, v_mfma_f32_16x16x32_bf16):
//     gW[co][ci][ky][kx] = sum over n, y, x of gy[n][y][x][co] * x[n][y+ky-1][x+kx-1][ci]
// -- per tap a GEMM with M = Cout, N = Cin and K = ALL PIXELS.  MIOpen runs
// these at 2 - 8 % of the bf16 peak (profiles/r04/conv_util_bf16.json: the
// backward of `upcnv1b` 330 - 370 us, of `upcnv2b` 420 - 440 us at 8 images).
//
// Both operands are channels-last bf16 (N x H x W x C): a pixel's channels are
// contiguous, but an MFMA operand wants, per lane, 8 consecutive K = 8 PIXELS of
// one channel -- the transpose.  gfx950's LDS transpose read does it:
// `ds_read_b64_tr_b16` hands lane t of a 16-lane group column t of a 4 (rows)
// x 16 (columns) block of 16-bit elements, each lane pointing at 4 consecutive
// elements: row t / 4, columns 4 (t % 4) .. + 3 (tools/tr_probe.hip prints it).
// With rows = pixels and columns = channels, two reads give a lane its 8 pixels
// of one channel: the A fragment (gy: 16 output channels x 32 pixels) and the B
// fragment (x shifted by the tap: 32 pixels x 16 input channels) come out of
// pixel-major LDS rows that are filled with plain 16-byte copies.
//
// Workgroup = (image, strip of SW columns, block of 32 rows) x (block of 32
// input channels) x (block of 32 or 64 output channels); it walks down its rows with three input rows (one-pixel halo
// left and right, zero outside the image) and the current gy row in LDS (a
// fourth x row and a second gy row being filled), the loads of the row after
// next in flight in registers while a row is multiplied: one barrier per row.
// The (tap, 16-channel tile) pairs of the B operand are dealt to the waves;
// every wave multiplies its pairs with all Cout / 16 A fragments:
// accumulators = the wave's share of the Cout x 32 x 9 result, in registers for
// the whole block.  Pixel blocks write partial results, conv_wgrad_reduce_kernel
// sums them into the layer's [Cout][Cin][3][3] fp32 gradient.
//
// The prediction head (32 -> 4 + bias + sigmoid, nets.py:150-158) uses the same
// kernel (PRED): its gy = g y (1 - y) is formed while the row is staged, stored
// as bf16 hi + lo in the tile's spare rows, the bias gradient is one more
// product with a fragment of ones (lsi_conv3x3_pred_bwd, lsi_conv.hip).
//
// LDS bank layout: a transpose read moves 32 bytes per (row, group); the rows a
// pass of 32 lanes touches (two groups x four rows) must fall into eight
// different 32-byte bank groups of the 256-byte bank period, so the pixel
// stride is padded to an ODD multiple of 32 bytes and the K order is permuted:
// group g's two reads take pixels 16 (g / 2) + 4 (g % 2) + 8 h + r (h = 0, 1:
// the read, r = 0 .. 3: the row) -- groups 0 and 1 then cover pixels 0 .. 7 or
// 8 .. 15 of the chunk in one pass.  The K order is free as long as A and B
// agree.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lsi_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#ifndef LSI_WG_ROWS
#define LSI_WG_ROWS 32
#endif
constexpr int WG_ROWS = LSI_WG_ROWS;  // rows per pixel block

struct WgradArgs {
  const __bf16* x;   // N x H x W x Cin
  const __bf16* gy;  // N x H x W x Cout
  float* part;       // [pixel blocks][Cout][Cin][9]
  int N, H, W, Cin, Cout;
  int nstrip, nrowblk;
  int rows;   // rows per pixel block (>= WG_ROWS: the workspace is sized for WG_ROWS)
  // PRED (the prediction head, 32 -> cout <= 4 + bias + sigmoid): gy = g y (1 - y)
  // is formed while the row is staged, from the incoming gradient and the saved
  // output (fp32 RGBD pixels, N x H x W x 4)
  const float* g;
  const float* yf;
};

// elements per pixel in LDS: the channels, padded to an odd multiple of 32 bytes
constexpr int padded(int ch) { return ((ch * 2 / 32) % 2 == 0) ? ch + 16 : ch; }

// The fragment of 16 channels x 32 pixels whose first pixel / channel `p`
// points at: lane (t = lane % 16, g = lane / 16) gets channel t of its 8 pixels.
template <int STRIDE>
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* p, int t, int g) {
  const __bf16* q = p + (16 * (g >> 1) + 4 * (g & 1) + (t >> 2)) * STRIDE + 4 * (t & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 8 * STRIDE));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// CO16: Cout / 16; NW: waves; SW: strip width (pixels, a multiple of 32).
// A workgroup takes 32 input channels (two tiles): blockIdx.y, and 16 CO16
// output channels: blockIdx.z.
template <int CO16, int NW, int SW, bool PRED = false>
__global__ __launch_bounds__(NW * 64) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int T = NW * 64;
  constexpr int CIB = 2;                 // input-channel tiles per workgroup
  constexpr int XS = padded(CIB * 16);   // LDS elements per pixel (x rows)
  constexpr int GS = padded(CO16 * 16);  // (gy row)
  constexpr int XROW = (SW + 2) * XS;
  constexpr int NXP = (SW + 2) * CIB * 2, XP = (NXP + T - 1) / T;  // 16-byte pieces
  constexpr int NGP = SW * CO16 * 2, GP = PRED ? 2 : (NGP + T - 1) / T;
  constexpr int NPAIR = 9 * CIB, PB = (NPAIR + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) __bf16 xs[4 * XROW];    // rows y - 1 .. y + 2: slot row & 3
  __shared__ __attribute__((aligned(16))) __bf16 gs[2 * SW * GS];  // rows y, y + 1: slot row & 1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, g = lane >> 4;
  int pb = blockIdx.x;
  const int rb = pb % a.nrowblk;
  pb /= a.nrowblk;
  const int st = pb % a.nstrip, n = pb / a.nstrip;
  const int x0 = st * SW, y0 = rb * a.rows, y1 = min(a.H, y0 + a.rows);
  const int c0 = blockIdx.y * (CIB * 16);  // first input channel
  const int o0 = blockIdx.z * (CO16 * 16);  // first output channel
  static_assert(!PRED || (CO16 == 1 && SW <= T), "PRED: one tile of output channels, a thread per pixel");
  const int H = a.H, W = a.W;
  const size_t img = (size_t)n * H;

  // global -> registers -> LDS, 16 bytes per piece; piece = (pixel, 8 channels)
  auto load_x = [&](int yy, u32x4 (&r)[XP]) {
    const bool yin = yy >= 0 && yy < H;
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int piece = tid + k * T;
      const int pp = piece / (CIB * 2), part = piece - pp * (CIB * 2);
      const int xx = x0 - 1 + pp;
      r[k] = u32x4{0u, 0u, 0u, 0u};
      if (piece < NXP && yin && xx >= 0 && xx < W)
        r[k] = *reinterpret_cast<const u32x4*>(a.x + ((img + yy) * W + xx) * a.Cin + c0 +
                                                8 * part);
    }
  };
  auto store_x = [&](int yy, const u32x4 (&r)[XP]) {
    __bf16* const row = xs + ((yy + 4) & 3) * XROW;
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int piece = tid + k * T;
      const int pp = piece / (CIB * 2), part = piece - pp * (CIB * 2);
      if (piece < NXP) *reinterpret_cast<u32x4*>(row + pp * XS + 8 * part) = r[k];
    }
  };
  auto load_g = [&](int yy, u32x4 (&r)[GP]) {
    if (PRED) {  // thread = pixel: the gradient and the saved output, fp32 RGBD
      const int xx = x0 + tid;
      r[0] = r[1] = u32x4{0u, 0u, 0u, 0u};
      if (tid < SW && yy < y1 && xx < W) {
        const size_t pix = (img + yy) * W + xx;
        r[0] = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(a.g + 4 * pix));
        r[1] = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(a.yf + 4 * pix));
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < GP; ++k) {
      const int piece = tid + k * T;
      const int pp = piece / (CO16 * 2), part = piece - pp * (CO16 * 2);
      const int xx = x0 + pp;
      r[k] = u32x4{0u, 0u, 0u, 0u};
      if (piece < NGP && yy < y1 && xx < W)
        r[k] = *reinterpret_cast<const u32x4*>(a.gy + ((img + yy) * W + xx) * a.Cout + o0 +
                                                8 * part);
    }
  };
  auto store_g = [&](int yy, const u32x4 (&r)[GP]) {
    __bf16* const grow = gs + (yy & 1) * (SW * GS);
    if (PRED) {
      // gz = g y (1 - y) in fp32, stored as bf16 hi (channels 0 .. 3) + lo
      // (channels 4 .. 7: what the rounding to bf16 left); channels 8 .. 15 zero.
      // The matrix tile has 16 rows for 4 outputs: the second bf16 is free.
      if (tid < SW) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const f32x4 gv = __builtin_bit_cast(f32x4, r[0]), yv = __builtin_bit_cast(f32x4, r[1]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gz = k < a.Cout ? gv[k] * yv[k] * (1.0f - yv[k]) : 0.0f;
          const __bf16 hi = (__bf16)gz;
          v[k] = hi;
          v[4 + k] = (__bf16)(gz - (float)hi);
        }
        *reinterpret_cast<bf16x8*>(grow + tid * GS) = v;
        *reinterpret_cast<u32x4*>(grow + tid * GS + 8) = u32x4{0u, 0u, 0u, 0u};
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < GP; ++k) {
      const int piece = tid + k * T;
      const int pp = piece / (CO16 * 2), part = piece - pp * (CO16 * 2);
      if (piece < NGP) *reinterpret_cast<u32x4*>(grow + pp * GS + 8 * part) = r[k];
    }
  };

  f32x4 acc[PB][CO16];
  f32x4 accb = {0.f, 0.f, 0.f, 0.f};  // PRED: row sums of gz = the bias gradient (last wave)
#pragma unroll
  for (int j = 0; j < PB; ++j)
#pragma unroll
    for (int m = 0; m < CO16; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Rows y - 1, y, y + 1 of x and row y of gy are read from LDS while row y is
  // multiplied; row y + 2 of x / y + 1 of gy (loads issued one row earlier) go
  // into the free slots after the multiplication, the loads of rows y + 3 / y + 2
  // are issued before it: two register sets, one barrier per row.
  u32x4 rxa[XP], rga[GP], rxb[XP], rgb[GP];
  for (int yy = y0 - 1; yy <= y0 + 1; ++yy) {
    load_x(yy, rxa);
    store_x(yy, rxa);
  }
  load_g(y0, rga);
  store_g(y0, rga);
  load_x(y0 + 2, rxa);
  load_g(y0 + 1, rga);
  __syncthreads();

  const int nchunk = min(SW, W - x0 + 31) / 32;  // chunks with pixels inside the image
  auto multiply_row = [&](int y) {
    const __bf16* const grow = gs + (y & 1) * (SW * GS);
    for (int q = 0; q < nchunk; ++q) {
      bf16x8 af[CO16];
#pragma unroll
      for (int m = 0; m < CO16; ++m) af[m] = tr_frag<GS>(grow + (32 * q) * GS + 16 * m, t, g);
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int p = wave + j * NW;  // (tap, input-channel tile), wave-uniform
        if (p < NPAIR) {
          const int tap = p / CIB, c = p - tap * CIB;
          const int ky = tap / 3, kx = tap - 3 * ky;
          const __bf16* const row = xs + ((y + ky + 3) & 3) * XROW;  // input row y + ky - 1
          const bf16x8 bf = tr_frag<XS>(row + (32 * q + kx) * XS + 16 * c, t, g);
#pragma unroll
          for (int m = 0; m < CO16; ++m)
            acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf, acc[j][m], 0, 0, 0);
        }
      }
      if (PRED && wave == NW - 1) {  // B = ones: every column of D is the row sum
        const __bf16 one = (__bf16)1.0f;
        const bf16x8 ones = {one, one, one, one, one, one, one, one};
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], ones, accb, 0, 0, 0);
      }
    }
  };
  for (int y = y0; y < y1; y += 2) {
    load_x(y + 3, rxb);
    load_g(y + 2, rgb);
    multiply_row(y);
    store_x(y + 2, rxa);  // (the slot of row y - 2)
    store_g(y + 1, rga);
    __syncthreads();
    if (y + 1 >= y1) break;
    load_x(y + 4, rxa);
    load_g(y + 3, rga);
    multiply_row(y + 1);
    store_x(y + 3, rxb);
    store_g(y + 2, rgb);
    __syncthreads();
  }

  // accumulator of lane (t, g), register r: co = 16 m + 4 g + r, ci = 16 c + t
  if (PRED) {
    // rows 0 .. 3 (lanes g = 0) hold the hi parts, rows 4 .. 7 (g = 1) the lo
    // parts: out[(co * 32 + ci) * 9 + tap], the bias gradient behind the weights
    const int nout = a.Cout * 288 + a.Cout;
    float* const out = a.part + (size_t)blockIdx.x * nout;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int p = wave + j * NW;
      const int tap = p / CIB, c = p - tap * CIB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[j][0][r] + __shfl_down(acc[j][0][r], 16);
        if (p < NPAIR && g == 0 && r < a.Cout) out[(r * 32 + 16 * c + t) * 9 + tap] = v;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = accb[r] + __shfl_down(accb[r], 16);
      if (wave == NW - 1 && lane == 0 && r < a.Cout) out[a.Cout * 288 + r] = v;
    }
    return;
  }
  float* const out = a.part + (size_t)blockIdx.x * a.Cout * a.Cin * 9;
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int p = wave + j * NW;
    if (p < NPAIR) {
      const int tap = p / CIB, c = p - tap * CIB;
#pragma unroll
      for (int m = 0; m < CO16; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((size_t)(o0 + 16 * m + 4 * g + r) * a.Cin + c0 + 16 * c + t) * 9 + tap] =
              acc[j][m][r];
    }
  }
}

// out[i] = sum over the pixel blocks' partials: 64 outputs x 16 block groups per
// workgroup, the groups folded through LDS.
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float* part, int nblk,
                                                                 int nout, float* out) {
  __shared__ float red[16][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  float s = 0.0f;
  if (o < nout) {
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int w = grp;
    for (; w + 48 < nblk; w += 64) {  // four loads in flight
      s += part[(size_t)w * nout + o];
      s1 += part[(size_t)(w + 16) * nout + o];
      s2 += part[(size_t)(w + 32) * nout + o];
      s3 += part[(size_t)(w + 48) * nout + o];
    }
    for (; w < nblk; w += 16) s += part[(size_t)w * nout + o];
    s += s1 + s2 + s3;
  }
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && o < nout) {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
    out[o] = v;
  }
}

constexpr int WG_SW = 64;  // strip width

int pixel_blocks(int N, int H, int W) {
  return N * ((W + WG_SW - 1) / WG_SW) * ((H + WG_ROWS - 1) / WG_ROWS);
}

}  // namespace

// The prediction head's weight + bias gradient (lsi_conv3x3_pred_bwd,
// lsi_conv.hip): the same kernel with gy = g y (1 - y) formed while a row is
// staged; g_wb = [cout * 288 + cout], written.
size_t lsi_pred_wgrad_workspace_bytes(int N, int H, int W, int cout) {
  return (size_t)pixel_blocks(N, H, W) * (cout * 288 + cout) * sizeof(float);
}
int lsi_pred_wgrad_launch(int N, int H, int W, int cout, const float* g, const float* y,
                          const void* x, float* g_wb, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
  if (workspace_bytes < lsi_pred_wgrad_workspace_bytes(N, H, W, cout)) return LSI_EWORKSPACE;
  WgradArgs a;
  a.x = reinterpret_cast<const __bf16*>(x);
  a.gy = nullptr;
  a.g = g;
  a.yf = y;
  a.part = reinterpret_cast<float*>(workspace);
  a.N = N; a.H = H; a.W = W; a.Cin = 32; a.Cout = cout;
  a.nstrip = (W + WG_SW - 1) / WG_SW;
  a.rows = WG_ROWS;
  a.nrowblk = (H + WG_ROWS - 1) / WG_ROWS;
  const int nblk = pixel_blocks(N, H, W), nout = cout * 288 + cout;
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<1, 4, WG_SW, true>), dim3(nblk), dim3(256), 0, stream,
                     a);
  if (hipGetLastError() != hipSuccess) return LSI_ELAUNCH;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((nout + 63) / 64), dim3(1024), 0, stream,
                     a.part, nblk, nout, g_wb);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

extern "C" size_t lsi_conv3x3_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W,
                                                    int32_t cin, int32_t cout) {
  if (N <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 0;
  return (size_t)pixel_blocks(N, H, W) * cin * cout * 9 * sizeof(float);
}

extern "C" int lsi_conv3x3_wgrad(int32_t N, int32_t H, int32_t W, int32_t cin, int32_t cout,
                                 const void* x, const void* gy, float* g_weight,
                                 void* workspace, size_t workspace_bytes,
                                 lsi_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || N > 65535) return LSI_EINVAL;
  if (cin % 32 != 0 || cin < 32 || cout % 32 != 0 || cout < 32) return LSI_EUNSUPPORTED;
  if (!x || !gy || !g_weight || !workspace) return LSI_ENULL;
  if (((uintptr_t)x & 15) || ((uintptr_t)gy & 15) || ((uintptr_t)workspace & 15))
    return LSI_EINVAL;
  if (workspace_bytes < lsi_conv3x3_wgrad_workspace_bytes(N, H, W, cin, cout))
    return LSI_EWORKSPACE;
  WgradArgs a;
  a.x = reinterpret_cast<const __bf16*>(x);
  a.gy = reinterpret_cast<const __bf16*>(gy);
  a.part = reinterpret_cast<float*>(workspace);
  a.g = a.yf = nullptr;
  a.N = N; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout;
  a.nstrip = (W + WG_SW - 1) / WG_SW;
  // Rows per pixel block: WG_ROWS, or more when that brings the launch down to
  // what is resident at once (the 8-wave build: two workgroups per CU, 512; the
  // 4-wave build: four, 1024) -- `upcnv2b` (96 -> 64 at 128 x 384) was 576
  // workgroups: two rounds of the chip for 1.1 rounds of work.
  {
    const long per_blk = (long)(cin / 32) * (cout % 64 ? cout / 32 : cout / 64);
    const long resident = (cout % 64) ? 1024 : 512;
    int rows = WG_ROWS;
    const long strips = (long)N * a.nstrip;
    while (rows < H && strips * ((H + rows - 1) / rows) * per_blk > resident &&
           strips * per_blk <= resident)
      ++rows;
    // (the smallest number of row blocks that fits, then rows evened out)
    const int nrb = (H + rows - 1) / rows;
    rows = (H + nrb - 1) / nrb;
    if (rows < WG_ROWS) rows = WG_ROWS;
    a.rows = rows;
  }
  a.nrowblk = (H + a.rows - 1) / a.rows;
  const int nblk = N * a.nstrip * a.nrowblk;
  if (cout % 64 != 0)
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<2, 4, WG_SW>), dim3(nblk, cin / 32, cout / 32),
                       dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 8, WG_SW>), dim3(nblk, cin / 32, cout / 64),
                       dim3(512), 0, (hipStream_t)stream, a);
  if (hipGetLastError() != hipSuccess) return LSI_ELAUNCH;
  const int nout = cout * cin * 9;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((nout + 63) / 64), dim3(1024), 0,
                     (hipStream_t)stream, a.part, nblk, nout, g_weight);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
