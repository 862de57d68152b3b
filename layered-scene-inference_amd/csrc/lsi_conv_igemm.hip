// Convolutions of the encoder-decoder and the LDI heads as implicit GEMMs on
// the matrix cores (gfx950, v_mfma_f32_16x16x32_bf16, fp32 accumulation):
// reference nets.py:29-70, 73-114, 244-348 -- slim.conv2d (k x k, stride 1 | 2,
// TF `SAME` padding) and slim.conv2d_transpose (4 x 4, stride 2) on bf16
// channels-last activations, forward and data gradient.
//
// One kernel, `conv_igemm_kernel`, computes
//     out[n][i * os + ooy][j * os + oox][co] =
//         sum over taps t, channels ci of  x[n][i * s + dy_t][j * s + dx_t][ci] * Wp[t][co][ci]
// for a list of taps (dy_t, dx_t) with zero outside the input.  Everything the
// network needs is that with a tap list:
//   * forward, stride s:       dy = ky - pad_t, dx = kx - pad_l, os = 1;
//   * data gradient, stride 1: the same sum over the incoming gradient with
//     dy = pad_t - ky and the channel roles swapped (Wp[t][ci][co]);
//   * data gradient, stride 2 (= the forward of a transposed convolution): the
//     input pixels of one parity class (iy, ix) = (2 i + p, 2 j + q) receive
//     only the taps with ky = p + pad_t (mod 2), from gradient pixel
//     i + (p + pad_t - ky) / 2: four launches, each a stride-1 sum over a
//     sub-kernel (2 x 2 taps for the 4 x 4 transposed convolutions) writing
//     every second pixel (os = 2) -- no zero-stuffed input, no wasted products;
//   * the data gradient of a transposed convolution is the forward with s = 2.
//
// Layout.  One pixel's 32 consecutive channels are 64 contiguous bytes = the
// K of ONE MFMA.  The weights are the A operand (16 output channels x 32 input
// channels), the pixels the B operand (lane l: pixel l & 15, channels
// 8 (l >> 4) .. + 7), so that the accumulator of lane l is four consecutive
// output channels of one pixel: channels-last stores without a transpose.
//
// Workgroup = 4 waves = a tile of (4 RW) rows x 16 columns of output pixels x BN
// output channels (BN = 64 | 32); wave w owns rows w RW .. + RW - 1.  Per chunk
// of 32 input channels the workgroup stages the input patch the tile's taps
// reach ((TH - 1) s + span rows, pixels 80 bytes apart: the 16 lanes of a
// fragment read then fall into 16 different 16-byte bank groups) and, per group
// of <= G taps, the weights [tap][co][32 ci] (rows 80 bytes apart) in LDS;
// every tap is then RW + BN / 16 fragment reads for RW * BN / 16 MFMAs.  Two or
// three workgroups share a CU (<= 80 KB of LDS each): one stages while another
// multiplies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lsi_hip.h"
#include "lsi_splat_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int IG_MAXTAPS = 49;
constexpr int IG_PIX = 80;  // bytes per staged pixel / weight row (64 + 16 of padding)

// One tap list = one class of output pixels: all of them (forward, stride-1
// data gradient) or one parity class of a stride-2 data gradient.
struct IgClass {
  int ntaps, dy0, dx0;   // taps; smallest dy, dx
  int ooy, oox;          // output pixel of (i, j): (i * os + ooy, j * os + oox)
  int OHt, OWt;          // output grid (i, j) of the class
  int wofs;              // first tap of the class in wp
  // patch byte offset of every tap, (dy - dy0) * PW + (dx - dx0) pixels; padded
  // with zeros to whole stages of G taps (their weights are staged as zeros)
  int toff[IG_MAXTAPS + 7];
  signed char tdy[IG_MAXTAPS + 3], tdx[IG_MAXTAPS + 3];
};
struct IgArgs {
  const __bf16* x;    // N x H x W x Cin (C1 channels of it when x2 != nullptr)
  // The input as TWO tensors (the skip connections, nets.py:104-106, 300-345:
  // tf.concat([upcnv, skip], axis=3) in front of a convolution): channels
  // [0, C1) come from x (N x H x W x C1), channels [C1, Cin) from x2
  // (N x H x W x (Cin - C1)) -- the concatenated tensor is never written.
  // C1 a multiple of 32 (a chunk of 32 input channels is one tensor's).
  const __bf16* x2;
  int C1;
  // ... and its data gradient as two tensors: output channels [0, O1) go to out
  // (pitch O1), [O1, Cout) to out2 (pitch Cout - O1); O1 a multiple of the
  // workgroup's block of output channels.
  __bf16* out2;
  int O1;
  const __bf16* wp;   // [taps of all classes][Cout][Cin]
  __bf16* out;        // N x OHF x OWF x Cout
  int N, H, W, Cin, Cout;
  int s;              // input pixels per output step
  int os;             // output pixels per (i, j) step
  int OHF, OWF;
  int G;              // taps per weight stage
  int PH, PW;         // staged patch: rows, pixels per row
  int ncls;
  // Batch-norm statistics of the output, accumulated in the epilogue (st_ws !=
  // nullptr; csrc/lsi_bn.hip's workspace): sums of y and y * y per channel and
  // sub-batch group of the bf16-ROUNDED outputs -- what lsi_bn_relu_fwd's first
  // pass would read the tensor back for --, fp32 device atomics into slot
  // (workgroup index mod st_ns) of the group's accumulators, fire and forget:
  // lsi_bn_relu_norm, the next kernel on the stream, folds and clears them.  (A
  // last-arriver protocol inside this kernel was built first: every workgroup
  // then waits for its adds and a counter's return before it leaves, ~4 us each
  // -- 91 -> 117 us for the 3072 workgroups of `upcnv1`, all that the
  // statistics pass had cost.)
  float* st_ws;
  int st_groups;   // sub-batch groups along N
  int st_ns;       // accumulator slots (lsi_bn_stat_slots)
  // Split over the input channels (ks > 1): workgroup (.., ksi) multiplies the
  // chunks [ksi nch / ks, (ksi + 1) nch / ks) of 32 input channels and leaves its
  // fp32 sums in part[ksi][n][y][x][co]; conv_splitk_fold_kernel adds the ks
  // slabs in a fixed order, rounds, stores (and takes the batch-norm sums).  For
  // the bottleneck maps (2 x 6 ... 16 x 48 pixels, 256 - 1024 channels), where
  // the tiles alone are 64 - 256 workgroups, each a chain of 16 - 32 staged
  // units that wait a load latency per 36 - 72 MFMAs.
  float* part;
  int ks;
  int swz;   // XCD-aware workgroup order (see the kernel's index decode)
  IgClass cls[4];
};

// sum over the 16 lanes of a DPP row (every lane gets it)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// RW: pixel rows per wave; NCT: tiles of 16 output channels (BN = 16 NCT); G:
// taps per weight stage -- a compile-time count, so that a stage is straight-
// line code: the tap offsets come in one batch of scalar loads and the
// scheduler overlaps a tap's fragment reads with the previous tap's MFMAs
template <int RW, int NCT, int G>
__global__ __launch_bounds__(256) void conv_igemm_kernel(IgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ig_smem[];
  constexpr int TH = 4 * RW, BN = 16 * NCT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pxl = lane & 15, kg = lane >> 4;
  const int ncb = a.Cout / BN;
  // Workgroup -> (tile, image, class, channel block, split).  The workgroups that
  // read the same input patch -- the parity classes of a stride-2 data gradient /
  // transposed convolution, the blocks of output channels, the splits -- used to
  // sit gridDim.x * gridDim.y * N apart in dispatch order: never in flight
  // together, on whatever XCD, and the patch came from HBM once per class
  // (`upcnv1`: 4 x 50 MB read for 100 MB written).  a.swz: with block b on XCD
  // b % 8 (observed placement; a speed matter only), the (class, channel block,
  // split) siblings of a tile are consecutive blocks OF ONE XCD -- eight apart in
  // dispatch order: in flight together, sharing that XCD's L2.
  int bx = blockIdx.x, by = blockIdx.y, co0, ksi, ci_, n;
  if (a.swz) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = lin & 7u, r = lin >> 3;
    const unsigned nsib = (unsigned)(a.ks * a.ncls * ncb);
    const unsigned g = r % nsib, tile = (r / nsib) * 8u + xcd;   // tile over gx * gy * N
    bx = (int)(tile % gridDim.x);
    by = (int)((tile / gridDim.x) % gridDim.y);
    n = (int)(tile / (gridDim.x * gridDim.y));
    co0 = (int)(g % ncb) * BN;
    ci_ = (int)((g / ncb) % a.ncls);
    ksi = (int)(g / (ncb * a.ncls));
  } else {
    const int zz = blockIdx.z / ncb;
    co0 = (blockIdx.z - zz * ncb) * BN;
    ksi = zz / (a.ncls * a.N);
    const int zc = zz - ksi * (a.ncls * a.N);
    ci_ = zc / a.N; n = zc - ci_ * a.N;
  }
  const IgClass& k = a.cls[ci_];
  const int i0 = by * TH, j0 = bx * 16;
  if (i0 >= k.OHt || j0 >= k.OWt) return;  // (the grid covers the largest class)
  const int PW = a.PW, npix = a.PH * PW;
  unsigned char* const patch = ig_smem;
  unsigned char* const wts = ig_smem + (size_t)npix * IG_PIX;

  // The patch pieces this thread stages (16 bytes each: pixel, quarter of its
  // 32 channels): pixel index into the input, -1 outside it.  The same for
  // every chunk.
  constexpr int MAXP = 12;  // (<= 768 staged pixels)
  int goff[MAXP];
  const int npiece = npix * 4;
  {
    const int iy0 = i0 * a.s + k.dy0, ix0 = j0 * a.s + k.dx0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int idx = tid + 256 * k;
      const int pix = idx >> 2;
      const int py = pix / PW, px = pix - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = idx < npiece && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      goff[k] = ok ? ((n * a.H + iy) * a.W + ix) : -1;   // (pixel index)
    }
  }
  f32x4 acc[RW][NCT];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // this lane's B fragments: pixel (row wave * RW + r, column pxl) of the tile
  const unsigned b_lane =
      (unsigned)((wave * RW * a.s) * PW + pxl * a.s) * IG_PIX + (unsigned)kg * 16u;
  const unsigned b_row = (unsigned)(a.s * PW) * IG_PIX;
  const int ntaps = k.ntaps;
  const __bf16* const wp = a.wp + (size_t)k.wofs * a.Cout * a.Cin;
  const unsigned a_lane = (unsigned)pxl * IG_PIX + (unsigned)kg * 16u;  // weight row pxl of a tile

  // Software pipeline over the units (chunk of 32 channels, group of G taps):
  // the loads of unit u + 1 -- its weights, and the chunk's patch when the unit
  // opens a chunk -- are issued into registers before unit u is multiplied and
  // written to LDS after it, so a stage's memory latency hides behind the
  // previous stage's MFMAs (one wave per SIMD and workgroup: nobody else would).
  constexpr int NWP = G * BN * 4, WB = (NWP + 255) / 256;
  const int ngrp = (ntaps + G - 1) / G;
  const int nch = a.Cin / 32;
  const int ch_lo = ksi * nch / a.ks, ch_hi = (ksi + 1) * nch / a.ks;   // (this split's chunks)
  const int nunit = (ch_hi - ch_lo) * ngrp;
  u32x4 pv[MAXP], wv[WB];
  auto fetch = [&](int u) {
    const int chl = u / ngrp, gi = u - chl * ngrp;
    const int c0 = (ch_lo + chl) * 32, t0 = gi * G;
    if (gi == 0) {
      // (the chunk's tensor: its channel count is the pixel pitch)
      const bool second = c0 >= a.C1;
      const __bf16* const xb = second ? a.x2 + (c0 - a.C1) : a.x + c0;
      const int pitch = second ? a.Cin - a.C1 : a.C1;
#pragma unroll
      for (int kk = 0; kk < MAXP; ++kk) {
        pv[kk] = zero4;
        if (tid + 256 * kk < npiece && goff[kk] >= 0)
          pv[kk] = *reinterpret_cast<const u32x4*>(xb + (size_t)goff[kk] * pitch +
                                                   8 * ((tid + 256 * kk) & 3));
      }
    }
    const __bf16* const wsrc = wp + ((size_t)t0 * a.Cout + co0) * a.Cin + c0;
    const int nreal = (ntaps - t0) * BN * 4;   // (taps past the class's last one: zeros)
#pragma unroll
    for (int kk = 0; kk < WB; ++kk) {
      const int idx = tid + 256 * kk;
      wv[kk] = zero4;
      if (idx < NWP && idx < nreal) {
        const int q = idx & 3, co = (idx >> 2) & (BN - 1), t = idx / (4 * BN);
        wv[kk] = *reinterpret_cast<const u32x4*>(wsrc + ((size_t)t * a.Cout + co) * a.Cin + 8 * q);
      }
    }
  };
  auto stash = [&](int u) {   // registers -> LDS
    const int gi = u % ngrp;
    if (gi == 0) {
#pragma unroll
      for (int kk = 0; kk < MAXP; ++kk) {
        const int idx = tid + 256 * kk;
        if (idx < npiece)
          *reinterpret_cast<u32x4*>(patch + (size_t)(idx >> 2) * IG_PIX + (idx & 3) * 16) = pv[kk];
      }
    }
#pragma unroll
    for (int kk = 0; kk < WB; ++kk) {
      const int idx = tid + 256 * kk;
      if (idx < NWP) {
        const int q = idx & 3, co = (idx >> 2) & (BN - 1), t = idx / (4 * BN);
        *reinterpret_cast<u32x4*>(wts + (size_t)(t * BN + co) * IG_PIX + q * 16) = wv[kk];
      }
    }
  };
  if (nunit > 0) fetch(0);
  for (int u = 0; u < nunit; ++u) {
    __syncthreads();  // (the previous unit's fragments have been read)
    stash(u);
    __syncthreads();
    if (u + 1 < nunit) fetch(u + 1);   // in flight while this unit is multiplied
    const int t0 = (u % ngrp) * G;
#pragma unroll
    for (int t = 0; t < G; ++t) {
      const unsigned toff = (unsigned)k.toff[t0 + t];
      bf16x8 af[NCT];
#pragma unroll
      for (int c = 0; c < NCT; ++c)
        af[c] = *reinterpret_cast<const bf16x8*>(wts + (size_t)(t * BN + 16 * c) * IG_PIX + a_lane);
      const unsigned char* const bp = patch + b_lane + toff;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bp + r * b_row);
#pragma unroll
        for (int c = 0; c < NCT; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bf, acc[r][c], 0, 0, 0);
      }
    }
  }
  // ---- channels-last stores: lane = 4 output channels of one pixel -----------
  const int j = j0 + pxl;
  if (a.ks > 1) {
    // ---- a split's fp32 sums: part[ksi][n][y][x][co] ----------------------------
    if (j < k.OWt) {
      float* const pb = a.part + (size_t)ksi * a.N * a.OHF * a.OWF * a.Cout;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int i = i0 + wave * RW + r;
        if (i < k.OHt) {
          float* const o = pb + (((size_t)n * a.OHF + (size_t)(i * a.os + k.ooy)) * a.OWF +
                                 (j * a.os + k.oox)) * a.Cout + co0 + 4 * kg;
#pragma unroll
          for (int c = 0; c < NCT; ++c) *reinterpret_cast<f32x4*>(o + 16 * c) = acc[r][c];
        }
      }
    }
    return;
  }
  if (a.st_ws) {
    // ---- batch-norm statistics of this tile (see IgArgs) -----------------------
    float ss[NCT][4], qq[NCT][4];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { ss[c][e] = 0.f; qq[c][e] = 0.f; }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = i0 + wave * RW + r;
      if (j < k.OWt && i < k.OHt) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = (float)(__bf16)acc[r][c][e];
            ss[c][e] += v;
            qq[c][e] = __builtin_fmaf(v, v, qq[c][e]);
          }
      }
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { ss[c][e] = row16_sum(ss[c][e]); qq[c][e] = row16_sum(qq[c][e]); }
    __syncthreads();  // (the last unit's fragments have been read)
    float* const red = reinterpret_cast<float*>(ig_smem);   // [4 waves][2][BN]
    if (pxl == 0) {
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wave * 2 + 0) * BN + 16 * c + 4 * kg + e] = ss[c][e];
          red[(wave * 2 + 1) * BN + 16 * c + 4 * kg + e] = qq[c][e];
        }
    }
    __syncthreads();
    const int C = a.Cout;
    const int per_grp = a.N / a.st_groups, grp = n / per_grp;
    // (the hand-over tag of the group, by the workgroup of its first tile: see
    // lsi_splat_internal.h)
    if (tid == 0 && bx == 0 && by == 0 && co0 == 0 && ci_ == 0 &&
        n == grp * per_grp)
      __hip_atomic_store(reinterpret_cast<int*>(a.st_ws + (size_t)grp * LSI_BN_WS_STRIDE) +
                             LSI_BN_WS_TAG,
                         LSI_BN_TAG(C, a.st_groups), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 2 * BN) {
      const int q = tid / BN, ch = tid - q * BN;
      const float v = (red[(0 * 2 + q) * BN + ch] + red[(1 * 2 + q) * BN + ch]) +
                      (red[(2 * 2 + q) * BN + ch] + red[(3 * 2 + q) * BN + ch]);
      const int f = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
      __hip_atomic_fetch_add(a.st_ws + (size_t)grp * LSI_BN_WS_STRIDE + LSI_BN_WS_ACC +
                                 (f % a.st_ns) * 2 * C + q * C + co0 + ch,
                             v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (j < k.OWt) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = i0 + wave * RW + r;
      if (i < k.OHt) {
        const bool o_second = co0 >= a.O1;
        __bf16* const o = (o_second ? a.out2 + (co0 - a.O1) : a.out + co0) +
            (((size_t)n * a.OHF + (size_t)(i * a.os + k.ooy)) * a.OWF + (j * a.os + k.oox)) *
                (o_second ? a.Cout - a.O1 : a.O1) + 4 * kg;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          bf16x4 v;
          v[0] = (__bf16)acc[r][c][0]; v[1] = (__bf16)acc[r][c][1];
          v[2] = (__bf16)acc[r][c][2]; v[3] = (__bf16)acc[r][c][3];
          *reinterpret_cast<bf16x4*>(o + 16 * c) = v;
        }
      }
    }
  }
}

// The sums of a split convolution (IgArgs::part): out = bf16(slab 0 + slab 1 +
// ...), the order fixed; the batch-norm sums of the rounded values as the
// unsplit kernel's epilogue takes them.  Block = 64 quads of channels x 4
// pixels; grid.x = image x pixel block (a block's pixels belong to one image =
// one sub-batch group), grid.y = blocks of 256 channels.
struct FoldArgs {
  const float* part;
  __bf16* out;
  __bf16* out2;
  int O1, Cout, N, ks, pb;
  long npix;       // pixels per image
  float* st_ws;
  int st_groups, st_ns;
};
__global__ __launch_bounds__(256) void conv_splitk_fold_kernel(FoldArgs a) {
  __shared__ float red[4][2][256];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int co = 4 * ((int)blockIdx.y * 64 + tx);
  const int n = blockIdx.x / a.pb, pbk = blockIdx.x - n * a.pb;
  const long per = (a.npix + a.pb - 1) / a.pb;
  const long p0 = pbk * per, p1 = p0 + per < a.npix ? p0 + per : a.npix;
  const bool live = co < a.Cout;
  float ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const size_t slab = (size_t)a.N * a.npix * a.Cout;
    const bool second = co >= a.O1;
    __bf16* const ob = second ? a.out2 + (co - a.O1) : a.out + co;
    const int pitch = second ? a.Cout - a.O1 : a.O1;
    for (long p = p0 + ty; p < p1; p += 4) {
      const size_t pix = (size_t)n * a.npix + p;
      const float* const src = a.part + pix * a.Cout + co;
      f32x4 v = *reinterpret_cast<const f32x4*>(src);
      for (int k = 1; k < a.ks; ++k) v += *reinterpret_cast<const f32x4*>(src + k * slab);
      bf16x4 o;
      o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
      *reinterpret_cast<bf16x4*>(ob + pix * pitch) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float r = (float)o[e];
        ss[e] += r;
        qq[e] = __builtin_fmaf(r, r, qq[e]);
      }
    }
  }
  if (!a.st_ws) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[ty][0][4 * tx + e] = ss[e]; red[ty][1][4 * tx + e] = qq[e]; }
  __syncthreads();
  const int C = a.Cout;
  const int per_grp = a.N / a.st_groups, grp = n / per_grp;
  const int tid = ty * 64 + tx;
  if (tid == 0 && blockIdx.y == 0 && pbk == 0 && n == grp * per_grp)
    __hip_atomic_store(reinterpret_cast<int*>(a.st_ws + (size_t)grp * LSI_BN_WS_STRIDE) +
                           LSI_BN_WS_TAG,
                       LSI_BN_TAG(C, a.st_groups), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int t = tid; t < 512; t += 256) {
    const int q = t >> 8, ch = t & 255;
    const int cc = (int)blockIdx.y * 256 + ch;
    if (cc < C) {
      const float v = (red[0][q][ch] + red[1][q][ch]) + (red[2][q][ch] + red[3][q][ch]);
      const int f = (int)(blockIdx.y * gridDim.x + blockIdx.x);
      __hip_atomic_fetch_add(a.st_ws + (size_t)grp * LSI_BN_WS_STRIDE + LSI_BN_WS_ACC +
                                 (f % a.st_ns) * 2 * C + q * C + cc,
                             v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Weights into the kernel's operand order: dst[t][o][i] = W[o][i][ky_t][kx_t]
// (tr = 0) or W[i][o][ky_t][kx_t] (tr = 1: the data gradients), rounded to bf16
// as torch.autocast rounds them.  W is the layer's fp32 parameter
// D0 x D1 x KH x KW (both multiples of 32).  A workgroup moves a 32 x 32 tile of
// (d0, d1) per tap through LDS, so that reads and (transposed) writes are both
// contiguous runs.
typedef LsiPackJob PackArgs;   // {w, dst, D0, D1, khw, tr, ntaps, block0, tap[]}
// PACK_TC taps per round: their 4 x PACK_TC loads per thread are in flight
// together and a round costs one pair of barriers (one tap per round was 338 us
// for the 36 M parameters x 2 directions of the networks: a chain of load
// latencies; the trainer packs after every optimiser step).
constexpr int PACK_TC = 8;
__device__ __forceinline__ void pack_tile(const PackArgs& a, int bx, int by,
                                          float (*tile)[32][33]) {
  const int a0 = by * 32, b0 = bx * 32;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  const size_t per = (size_t)a.D0 * a.D1;
  __bf16* const dst = reinterpret_cast<__bf16*>(a.dst);
  for (int t0 = 0; t0 < a.ntaps; t0 += PACK_TC) {
    float v[PACK_TC][4];
#pragma unroll
    for (int tt = 0; tt < PACK_TC; ++tt) {
      const int tap = a.tap[min(t0 + tt, a.ntaps - 1)];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + 8 * j;
        // (the parameter D0 x D1 x KH x KW as torch stores it: contiguous, or --
        // tr & 2 -- channels-last strides, D0 x KH x KW x D1 in memory: what
        // module.to(memory_format=torch.channels_last) leaves)
        v[tt][j] = (a.tr & 2) ? a.w[((size_t)(a0 + r) * a.khw + tap) * a.D1 + b0 + c]
                              : a.w[((size_t)(a0 + r) * a.D1 + b0 + c) * a.khw + tap];
      }
    }
#pragma unroll
    for (int tt = 0; tt < PACK_TC; ++tt)
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[tt][r0 + 8 * j][c] = v[tt][j];
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < PACK_TC; ++tt) {
      const int t = t0 + tt;
      if (t < a.ntaps) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = r0 + 8 * j;
          if (a.tr & 1)   // dst[t][d1][d0]
            dst[(size_t)t * per + (size_t)(b0 + r) * a.D0 + a0 + c] = (__bf16)tile[tt][c][r];
          else        // dst[t][d0][d1]
            dst[(size_t)t * per + (size_t)(a0 + r) * a.D1 + b0 + c] = (__bf16)tile[tt][r][c];
        }
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void conv_pack_kernel(PackArgs a) {
  __shared__ float tile[PACK_TC][32][33];
  pack_tile(a, blockIdx.x, blockIdx.y, tile);
}
// Many layers in one launch: job j owns blocks [block0_j, block0_{j+1}) (the
// table lives in device memory; the last entry's block0 + its blocks = grid).
__global__ __launch_bounds__(256) void conv_pack_many_kernel(const PackArgs* jobs, int njobs) {
  __shared__ float tile[PACK_TC][32][33];
  __shared__ int which;
  if (threadIdx.x == 0) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].block0) ++j;
    which = j;
  }
  __syncthreads();
  const PackArgs a = jobs[which];
  const int lb = (int)blockIdx.x - a.block0, nbx = a.D1 / 32;
  pack_tile(a, lb % nbx, lb / nbx, tile);
}

constexpr size_t IG_LDS_CAP = 80 * 1024;  // two workgroups per CU

// Tile shape for the tap lists: the tallest row block whose patch + one weight
// stage fit the LDS share; G = as many taps per stage as fit next to the patch.
bool ig_shape(IgArgs& k, int* rw_out, int* nct_out, size_t* lds_out) {
  const int nct = (k.Cout % 64 == 0) ? 4 : 2;
  const int bn = 16 * nct;
  int spany = 1, spanx = 1, maxtaps = 0, maxoh = 1;
  for (int c = 0; c < k.ncls; ++c) {
    IgClass& q = k.cls[c];
    int dy1 = -128, dx1 = -128, dy0 = 127, dx0 = 127;
    for (int t = 0; t < q.ntaps; ++t) {
      dy0 = q.tdy[t] < dy0 ? q.tdy[t] : dy0; dy1 = q.tdy[t] > dy1 ? q.tdy[t] : dy1;
      dx0 = q.tdx[t] < dx0 ? q.tdx[t] : dx0; dx1 = q.tdx[t] > dx1 ? q.tdx[t] : dx1;
    }
    if (q.ntaps == 0) { dy0 = dy1 = dx0 = dx1 = 0; }
    q.dy0 = dy0; q.dx0 = dx0;
    spany = dy1 - dy0 + 1 > spany ? dy1 - dy0 + 1 : spany;
    spanx = dx1 - dx0 + 1 > spanx ? dx1 - dx0 + 1 : spanx;
    maxtaps = q.ntaps > maxtaps ? q.ntaps : maxtaps;
    maxoh = q.OHt > maxoh ? q.OHt : maxoh;
  }
  k.PW = 15 * k.s + spanx;
  static const char* rw_env = getenv("LSI_IGEMM_MAXRW");   // experiments
  const int rw_max = rw_env ? atoi(rw_env) : 8;
  for (int rw = rw_max; rw >= 1; rw >>= 1) {
    if (rw > 1 && 4 * (rw / 2) >= maxoh) continue;  // (a shorter block covers the rows)
    // (rows 8 per wave: 12 fragment reads for 32 MFMAs instead of 8 for 16 -- the
    // LDS read rate is what bounds the wide layers -- where the tile count still
    // fills the chip twice)
    if (rw == 8 && (long)((maxoh + 31) / 32) * ((k.cls[0].OWt + 15) / 16) * k.N * k.ncls *
                           (k.Cout / bn) < 1024)
      continue;
    // (... and shorter tiles while the launch would leave CUs without a
    // workgroup: `icnv5` 512 -> 256 at 16 x 48 had 96 workgroups of 16 rows,
    // 58 us; 192 of 8 rows: 40 us)
    static const char* fill_env = getenv("LSI_IGEMM_MINWG");   // experiments
    const long min_wg = fill_env ? atol(fill_env) : 256;
    if (rw > 1 && (long)((maxoh + 4 * rw - 1) / (4 * rw)) * ((k.cls[0].OWt + 15) / 16) * k.N *
                          k.ncls * (k.Cout / bn) < min_wg)
      continue;
    const int th = 4 * rw;
    k.PH = (th - 1) * k.s + spany;
    const size_t patch = (size_t)k.PH * k.PW * IG_PIX;
    if (k.PH * k.PW > 768) continue;
    if (patch + (size_t)bn * IG_PIX > IG_LDS_CAP) continue;
    // taps per weight stage: one of the instantiated counts -- the one that
    // wastes the fewest padded taps, then the fewest stages -- that fits
    static const int GS[4] = {9, 7, 5, 4};
    int g = 0, best_pad = 1 << 30, best_st = 1 << 30;
    for (int gi = 0; gi < 4; ++gi) {
      const int gc = GS[gi];
      if (patch + (size_t)gc * bn * IG_PIX > IG_LDS_CAP) continue;
      int pad = 0, st = 0;
      for (int c = 0; c < k.ncls; ++c) {
        const int n = k.cls[c].ntaps, stages = (n + gc - 1) / gc;
        pad += stages * gc - n; st += stages;
      }
      if (pad < best_pad || (pad == best_pad && st < best_st)) {
        best_pad = pad; best_st = st; g = gc;
      }
    }
    if (!g) continue;
    k.G = g;
    for (int c = 0; c < k.ncls; ++c) {
      IgClass& q = k.cls[c];
      for (int t = 0; t < IG_MAXTAPS + 7; ++t)
        q.toff[t] = t < q.ntaps
            ? ((q.tdy[t] - q.dy0) * k.PW + (q.tdx[t] - q.dx0)) * IG_PIX : 0;
    }
    *rw_out = rw; *nct_out = nct;
    *lds_out = patch + (size_t)g * bn * IG_PIX;
    return true;
  }
  return false;
}

// The split over the input channels (IgArgs::ks) of a launch of `nwg` tiles:
// as many splits as bring the launch to ~2 workgroups per CU, at least two
// chunks each (LSI_IGEMM_SPLITK: the target workgroup count, 0 = never).
int ig_splits(const IgArgs& k, long nwg) {
  static const char* env = getenv("LSI_IGEMM_SPLITK");   // experiments
  const long target = env ? atol(env) : 512;
  const int nch = k.Cin / 32;
  if (target <= 0 || nch < 4 || nwg <= 0 || nwg * 2 > target) return 1;
  long ks = target / nwg;
  if (ks > nch / 2) ks = nch / 2;
  if (ks > 16) ks = 16;
  return ks < 2 ? 1 : (int)ks;
}

static inline int bn_of(int nct) { return 16 * nct; }

struct IgPlan {
  int rw, nct, ks;
  size_t lds;
  dim3 grid;   // (grid.z without the splits)
};

int ig_plan(IgArgs& k, IgPlan* p) {
  if (!ig_shape(k, &p->rw, &p->nct, &p->lds)) return LSI_EUNSUPPORTED;
  const int th = 4 * p->rw, bn = 16 * p->nct;
  int oh = 0, ow = 0;
  for (int c = 0; c < k.ncls; ++c) {
    oh = k.cls[c].OHt > oh ? k.cls[c].OHt : oh;
    ow = k.cls[c].OWt > ow ? k.cls[c].OWt : ow;
  }
  p->ks = 1;
  p->grid = dim3(0, 0, 0);
  if (oh <= 0 || ow <= 0) return LSI_OK;
  p->grid = dim3((ow + 15) / 16, (oh + th - 1) / th, k.ncls * k.N * (k.Cout / bn));
  if (p->grid.z > 65535 || p->grid.y > 65535) return LSI_EINVAL;
  p->ks = ig_splits(k, (long)p->grid.x * p->grid.y * p->grid.z);
  if ((long)p->grid.z * p->ks > 65535) p->ks = 1;
  return LSI_OK;
}

size_t ig_part_bytes(const IgArgs& k, int ks) {
  return ks > 1 ? (size_t)ks * k.N * k.OHF * k.OWF * k.Cout * sizeof(float) : 0;
}

int ig_launch(IgArgs& k, hipStream_t stream, void* workspace = nullptr,
              size_t workspace_bytes = 0) {
  IgPlan pl;
  const int prc = ig_plan(k, &pl);
  if (prc != LSI_OK) return prc;
  if (pl.grid.x == 0) return LSI_OK;
  const int rw = pl.rw, nct = pl.nct;
  const size_t lds = pl.lds;
  dim3 grid = pl.grid;
  k.ks = 1;
  k.part = nullptr;
  if (pl.ks > 1 && workspace && !((uintptr_t)workspace & 15) &&
      workspace_bytes >= ig_part_bytes(k, pl.ks)) {
    k.ks = pl.ks;
    k.part = (float*)workspace;
    grid.z *= pl.ks;
  }
  {
    // (bijective only when the tiles split evenly over the eight XCDs.  Taken for
    // the parity classes of large maps only -- `upcnv1` 88 -> 71 us forward,
    // profiles/r06/conv_bench_swz{0,1}.txt; where the WEIGHTS are the traffic -- the
    // bottleneck maps, their splits and channel blocks -- the workgroups that share
    // weights are neighbours in the plain order and the swizzle costs 3 - 8 us.)
    static const char* env = getenv("LSI_IGEMM_SWZ");   // experiments
    const long tiles = (long)grid.x * grid.y * k.N;
    k.swz = (env ? atoi(env) != 0 : true) && k.ks == 1 && k.ncls > 1 && tiles >= 512 &&
            tiles % 8 == 0;
  }
  const void* fn = nullptr;
#define IG_CASE(R, C, GG) \
  if (rw == R && nct == C && k.G == GG) fn = (const void*)conv_igemm_kernel<R, C, GG>
#define IG_CASES(GG) \
  IG_CASE(8, 4, GG); IG_CASE(8, 2, GG); \
  IG_CASE(4, 4, GG); IG_CASE(2, 4, GG); IG_CASE(1, 4, GG); \
  IG_CASE(4, 2, GG); IG_CASE(2, 2, GG); IG_CASE(1, 2, GG)
  IG_CASES(9); IG_CASES(7); IG_CASES(5); IG_CASES(4);
#undef IG_CASES
#undef IG_CASE
  if (!fn) return LSI_EINVAL;
  {
    static const char* dbg = getenv("LSI_IG_DEBUG");   // (experiments: the plan of every call)
    if (dbg)
      fprintf(stderr, "ig N%d %dx%d cin %d cout %d s%d os%d ncls %d taps %d: RW %d NCT %d G %d grid %u x %u x %u = %u WGs (ks %d), lds %zu\n",
              k.N, k.H, k.W, k.Cin, k.Cout, k.s, k.os, k.ncls, k.cls[0].ntaps, rw, nct, k.G, grid.x, grid.y,
              grid.z, grid.x * grid.y * grid.z, k.ks, lds);
  }
  if (lsi_ensure_dynamic_lds(fn, lds) != LSI_OK) return LSI_ELAUNCH;
  void* kargs[1] = {&k};
  if (hipLaunchKernel(fn, grid, dim3(256), kargs, lds, stream) != hipSuccess) return LSI_ELAUNCH;
  if (hipGetLastError() != hipSuccess) return LSI_ELAUNCH;
  if (k.ks > 1) {
    FoldArgs f;
    f.part = k.part; f.out = k.out; f.out2 = k.out2; f.O1 = k.O1; f.Cout = k.Cout;
    f.N = k.N; f.ks = k.ks;
    f.npix = (long)k.OHF * k.OWF;
    // (pixel blocks per image: ~2 workgroups per CU over the launch, >= 4 pixels each)
    const int cb = (k.Cout + 255) / 256;
    long pb = 512 / ((long)k.N * cb);
    if (pb > (f.npix + 3) / 4) pb = (f.npix + 3) / 4;
    f.pb = pb < 1 ? 1 : (int)pb;
    f.st_ws = k.st_ws; f.st_groups = k.st_groups; f.st_ns = k.st_ns;
    hipLaunchKernelGGL(conv_splitk_fold_kernel, dim3(k.N * f.pb, cb), dim3(64, 4), 0, stream, f);
    if (hipGetLastError() != hipSuccess) return LSI_ELAUNCH;
  }
  return LSI_OK;
}

bool desc_ok(const LsiConvDesc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0) return false;
  if (d->Cin <= 0 || d->Cout <= 0 || d->Cin % 32 || d->Cout % 32) return false;
  if (d->KH < 1 || d->KW < 1 || d->KH > 7 || d->KW > 7) return false;
  if (d->stride != 1 && d->stride != 2) return false;
  if (d->pad_t < 0 || d->pad_l < 0 || d->pad_t >= d->KH || d->pad_l >= d->KW) return false;
  // int32 element offsets inside the kernels
  if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31)) return false;
  if ((int64_t)d->N * d->OH * d->OW * d->Cout >= (1ll << 31)) return false;
  return true;
}

// The tap lists of a call.  mode 0: forward (one class); mode 1: data gradient
// (stride^2 parity classes of input pixels).  `tap` receives ky * KW + kx of
// every tap in class order (the order of the packed weights).
void ig_classes(const LsiConvDesc* d, int mode, IgArgs& k, int8_t* tap) {
  memset(&k, 0, sizeof(k));
  int nt = 0;
  if (mode == 0) {
    IgClass& q = k.cls[0];
    for (int y = 0; y < d->KH; ++y)
      for (int xk = 0; xk < d->KW; ++xk) {
        tap[nt] = (signed char)(y * d->KW + xk);
        q.tdy[q.ntaps] = (signed char)(y - d->pad_t);
        q.tdx[q.ntaps] = (signed char)(xk - d->pad_l);
        ++q.ntaps; ++nt;
      }
    q.OHt = d->OH; q.OWt = d->OW; q.wofs = 0;
    k.ncls = 1;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Cout = d->Cout;
    k.s = d->stride; k.os = 1; k.OHF = d->OH; k.OWF = d->OW;
    return;
  }
  const int s = d->stride;
  // input pixels (iy, ix) = (s i + p, s j + q): the taps with ky = p + pad_t (mod s)
  //   gx[s i + p] += gy[oy] W[ky],   s oy + ky - pad_t = s i + p
  for (int p = 0; p < s; ++p)
    for (int q_ = 0; q_ < s; ++q_) {
      IgClass& q = k.cls[k.ncls++];
      q.wofs = nt;
      for (int y = 0; y < d->KH; ++y) {
        if ((p + d->pad_t - y) % s != 0) continue;
        for (int xk = 0; xk < d->KW; ++xk) {
          if ((q_ + d->pad_l - xk) % s != 0) continue;
          tap[nt] = (signed char)(y * d->KW + xk);
          q.tdy[q.ntaps] = (signed char)((p + d->pad_t - y) / s);
          q.tdx[q.ntaps] = (signed char)((q_ + d->pad_l - xk) / s);
          ++q.ntaps; ++nt;
        }
      }
      q.ooy = p; q.oox = q_;
      q.OHt = (d->H - p + s - 1) / s; q.OWt = (d->W - q_ + s - 1) / s;
    }
  k.N = d->N; k.H = d->OH; k.W = d->OW; k.Cin = d->Cout; k.Cout = d->Cin;
  k.s = 1; k.os = s; k.OHF = d->H; k.OWF = d->W;
}

}  // namespace

extern "C" int lsi_conv2d_supported(const LsiConvDesc* d) { return desc_ok(d) ? 1 : 0; }

extern "C" size_t lsi_conv2d_packed_bytes(const LsiConvDesc* d) {
  if (!desc_ok(d)) return 0;
  return (size_t)d->KH * d->KW * d->Cin * d->Cout * sizeof(__bf16);
}

extern "C" int lsi_conv2d_pack_job(const LsiConvDesc* d, int32_t mode, const float* weight,
                                   void* packed, size_t packed_bytes, LsiPackJob* job,
                                   int32_t* nblocks) {
  if (!d || !weight || !packed || !job || !nblocks) return LSI_ENULL;
  if (!desc_ok(d)) return LSI_EUNSUPPORTED;
  if (mode < 0 || mode > 3) return LSI_EINVAL;
  if ((uintptr_t)packed & 15) return LSI_EINVAL;
  if (packed_bytes < lsi_conv2d_packed_bytes(d)) return LSI_EWORKSPACE;
  IgArgs k;
  memset(job, 0, sizeof(*job));
  ig_classes(d, mode & 1, k, job->tap);
  job->w = weight; job->dst = packed;
  job->D0 = d->Cout; job->D1 = d->Cin; job->khw = d->KH * d->KW; job->tr = mode;
  job->ntaps = d->KH * d->KW;
  job->block0 = 0;
  *nblocks = (d->Cin / 32) * (d->Cout / 32);
  return LSI_OK;
}

extern "C" int lsi_conv2d_pack(const LsiConvDesc* d, int32_t mode, const float* weight,
                               void* packed, size_t packed_bytes, lsi_stream_t stream_) {
  LsiPackJob p;
  int32_t nb;
  const int rc = lsi_conv2d_pack_job(d, mode, weight, packed, packed_bytes, &p, &nb);
  if (rc != LSI_OK) return rc;
  hipLaunchKernelGGL(conv_pack_kernel, dim3(d->Cin / 32, d->Cout / 32), dim3(256), 0,
                     (hipStream_t)stream_, p);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

extern "C" int lsi_conv2d_pack_many(const LsiPackJob* jobs_device, int32_t njobs,
                                    int32_t total_blocks, lsi_stream_t stream_) {
  if (!jobs_device) return LSI_ENULL;
  if (njobs <= 0 || total_blocks <= 0) return LSI_EINVAL;
  hipLaunchKernelGGL(conv_pack_many_kernel, dim3(total_blocks), dim3(256), 0,
                     (hipStream_t)stream_, jobs_device, njobs);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

struct IgStats {
  float* ws; int groups;
};

static int ig_run(const LsiConvDesc* d, int mode, const void* src, const void* packed,
                  void* dst, lsi_stream_t stream_, const IgStats* st = nullptr,
                  const void* src2 = nullptr, int c1 = 0, void* dst2 = nullptr,
                  void* workspace = nullptr, size_t workspace_bytes = 0) {
  if (!d || !src || !packed || !dst) return LSI_ENULL;
  if (!desc_ok(d)) return LSI_EUNSUPPORTED;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7) || ((uintptr_t)packed & 15)) return LSI_EINVAL;
  IgArgs k;
  int8_t tap[56];
  ig_classes(d, mode, k, tap);
  k.x = (const __bf16*)src; k.wp = (const __bf16*)packed; k.out = (__bf16*)dst;
  k.C1 = k.Cin;
  k.O1 = k.Cout;
  k.ks = 1;
  if (dst2) {   // the data gradient as two tensors
    const int bn = (k.Cout % 64 == 0) ? 64 : 32;
    if (mode != 1 || ((uintptr_t)dst2 & 7)) return LSI_EINVAL;
    if (c1 <= 0 || c1 >= k.Cout || c1 % bn) return LSI_EINVAL;
    k.out2 = (__bf16*)dst2;
    k.O1 = c1;
  }
  if (src2) {   // the input as two tensors (forward only)
    if (mode != 0 || ((uintptr_t)src2 & 15)) return LSI_EINVAL;
    if (c1 <= 0 || c1 >= k.Cin || c1 % 32) return LSI_EINVAL;
    k.x2 = (const __bf16*)src2;
    k.C1 = c1;
  }
  if (st) {
    if (!st->ws) return LSI_ENULL;
    // (the accumulators of a group are 2 x 2048 floats; whole images per group)
    if (st->groups < 1 || k.N % st->groups || k.Cout > 2048) return LSI_EINVAL;
    k.st_ws = st->ws; k.st_groups = st->groups; k.st_ns = lsi_bn_stat_slots(k.Cout);
  }
  return ig_launch(k, (hipStream_t)stream_, workspace, workspace_bytes);
}

extern "C" size_t lsi_conv2d_workspace_bytes(const LsiConvDesc* d, int32_t mode) {
  if (!desc_ok(d) || mode < 0 || mode > 1) return 0;
  IgArgs k;
  int8_t tap[56];
  ig_classes(d, mode, k, tap);
  k.C1 = k.Cin; k.O1 = k.Cout; k.ks = 1;
  IgPlan pl;
  if (ig_plan(k, &pl) != LSI_OK || pl.grid.x == 0) return 0;
  return ig_part_bytes(k, pl.ks);
}

extern "C" int lsi_conv2d_run(const LsiConvDesc* d, int32_t mode, const LsiConvIO* io,
                              lsi_stream_t stream) {
  if (!io) return LSI_ENULL;
  if (mode < 0 || mode > 1) return LSI_EINVAL;
  if (io->workspace_bytes && !io->workspace) return LSI_ENULL;
  const IgStats st = {io->bn_workspace, io->groups};
  if (mode == 0 && io->out2) return LSI_EINVAL;
  if (mode == 1 && io->x2) return LSI_EINVAL;
  return ig_run(d, mode, io->x, io->packed, io->out, stream, io->bn_workspace ? &st : nullptr,
                io->x2, io->c1, io->out2, io->workspace, io->workspace_bytes);
}

extern "C" int lsi_conv2d_fwd(const LsiConvDesc* d, const void* x, const void* packed,
                              void* out, lsi_stream_t stream) {
  return ig_run(d, 0, x, packed, out, stream);
}

extern "C" int lsi_conv2d_bwd_data(const LsiConvDesc* d, const void* gy, const void* packed,
                                   void* gx, lsi_stream_t stream) {
  return ig_run(d, 1, gy, packed, gx, stream);
}

extern "C" int lsi_conv2d_fwd_bnstats(const LsiConvDesc* d, const void* x, const void* packed,
                                      void* out, float* bn_workspace, int32_t groups,
                                      lsi_stream_t stream) {
  const IgStats st = {bn_workspace, groups};
  return ig_run(d, 0, x, packed, out, stream, &st);
}

extern "C" int lsi_conv2d_fwd_cat(const LsiConvDesc* d, const void* x1, const void* x2,
                                  int32_t c1, const void* packed, void* out,
                                  float* bn_workspace, int32_t groups, lsi_stream_t stream) {
  if (!x2) return LSI_ENULL;
  const IgStats st = {bn_workspace, groups};
  return ig_run(d, 0, x1, packed, out, stream, bn_workspace ? &st : nullptr, x2, c1);
}

extern "C" int lsi_conv2d_bwd_data_cat(const LsiConvDesc* d, const void* gy, const void* packed,
                                       void* gx1, void* gx2, int32_t c1, lsi_stream_t stream) {
  if (!gx2) return LSI_ENULL;
  return ig_run(d, 1, gy, packed, gx1, stream, nullptr, nullptr, c1, gx2);
}

extern "C" int lsi_conv2d_bwd_data_bnstats(const LsiConvDesc* d, const void* gy,
                                           const void* packed, void* gx, float* bn_workspace,
                                           int32_t groups, lsi_stream_t stream) {
  const IgStats st = {bn_workspace, groups};
  return ig_run(d, 1, gy, packed, gx, stream, &st);
}
