// Weight gradient of any k x k, stride 1 | 2 convolution of the encoder-decoder
// and the heads (reference nets.py:29-114, 244-348; TF autodiff of slim.conv2d /
// slim.conv2d_transpose) on the matrix cores (gfx950, v_mfma_f32_16x16x32_bf16):
//     gW[co][ci][ky][kx] = sum over n, oy, ox of
//         gy[n][oy][ox][co] * x[n][oy * s + ky - pad_t][ox * s + kx - pad_l][ci]
// -- per tap a GEMM with M = Cout, N = Cin and K = ALL OUTPUT PIXELS.  The
// 3 x 3 stride-1 layers at full and half resolution keep their own kernel
// (lsi_conv_wgrad.hip: a ring of rows, every activation staged once); this one
// takes every other shape with the patch staging of the forward kernel
// (lsi_conv_igemm.hip) and the operand transposition of lsi_conv_wgrad.hip:
//
// * Both operands are channels-last: a pixel's channels are contiguous, an MFMA
//   operand wants 8 consecutive K = 8 pixels of one channel per lane.  The LDS
//   transpose read (`ds_read_b64_tr_b16`, tr_frag below) hands every lane its 8
//   pixels of one channel out of pixel-major LDS rows; every lane points at its
//   own rows, so the shifted (tap) and strided (stride 2) pixels of the input
//   patch are plain address arithmetic.
// * Workgroup = (image, strip of 32 output columns, block of RB output rows) x
//   (32 input channels, group of <= 9 taps) x (BN = 64 | 32 output channels); it
//   walks down its rows TH at a time: the input patch the taps reach and the gy
//   tile (TH x 32 pixels) are staged in LDS, then every K step (one tile row =
//   32 pixels) is NCT A fragments (gy) times the wave's (tap, 16-channel tile)
//   pairs.  The accumulators -- the wave's share of taps x BN x 32 -- stay in
//   registers for the whole block; pixel blocks write partial sums, the fold
//   kernel of lsi_conv_wgrad.hip's design sums them (deterministic).
// * A transposed convolution's weight gradient is the same sum with the roles
//   of the tensors swapped (see lsi_hip.h: it is the forward convolution
//   {2h x 2w x Cout_T -> h x w x Cin_T} whose "input" is the transposed
//   convolution's output gradient and whose "output gradient" is its input).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lsi_hip.h"
#include "lsi_splat_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int GW_XS = 48;     // LDS elements per staged input pixel: 32 channels + 16 (96 bytes, odd x 32)
constexpr int GW_GT = 9;      // taps per workgroup
constexpr int GW_PB = 5;      // (tap, 16-channel tile) pairs per wave: 4 waves x 5 >= 2 x 9
constexpr int GW_MAXTAPS = 49;
constexpr size_t GW_LDS_CAP = 80 * 1024;
constexpr size_t GW_PART_CAP = 96u << 20;  // partial sums: at most this many bytes

struct GwArgs {
  const __bf16* x2;   // the input as two tensors (lsi_conv_igemm.hip: IgArgs::x2): channels [C1, Cin)
  int C1;
  const __bf16* x;    // N x H x W x Cin (x C1 with x2)
  const __bf16* gy;   // N x OH x OW x Cout
  float* part;        // [pixel blocks][khw][Cout][Cin]
  int N, H, W, Cin, OH, OW, Cout;
  int khw, ntaps, ntg;        // kernel taps; tap groups of GW_GT
  int dy0, dx0, PH, PW, TH;   // patch origin (smallest dy, dx), size; tile rows per stage
  int nstrip;
  // Stages = (image, TH output rows) of a 32-column strip; the strip's N * nrs
  // stages are dealt to its PS workgroups round-robin (workgroup p: stages p,
  // p + PS, ...), PS chosen so that ALL workgroups of the launch are resident at
  // once (two per CU by registers: 512) -- a grid of 576 or 768 workgroups was two
  // rounds of the chip for 1.1 - 1.5 rounds of work.  Every workgroup writes one
  // partial sum (nstrip * PS of them to fold).
  int nrs, PS;
  // XCD-aware workgroup order: the workgroups of one pixel block -- the (input
  // chunk, tap group, output block) siblings, which read the same input rows and
  // the same gy tile -- are consecutive blocks of ONE XCD (block b runs on XCD
  // b % 8: observed, a speed matter only) instead of gridDim.x apart: in flight
  // together, sharing that XCD's L2.  Needs gridDim.x % 8 == 0.
  int swz;
  signed char tdy[GW_MAXTAPS + 3], tdx[GW_MAXTAPS + 3];
};

// 16 channels x 32 pixels at `p` (first pixel / first channel), pixel pitch
// STRIDE elements: lane (t = lane % 16, g = lane / 16) gets channel t of its 8
// pixels 16 (g / 2) + 4 (g % 2) + 8 h + r (the K order, the same for A and B).
template <int STRIDE>
__device__ __forceinline__ bf16x8 gw_tr_frag(const __bf16* p, int t, int g) {
  const __bf16* q = p + (16 * (g >> 1) + 4 * (g & 1) + (t >> 2)) * STRIDE + 4 * (t & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 8 * STRIDE));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int gw_gs(int bn) { return bn == 64 ? 80 : 48; }  // gy pixel pitch (odd x 32 bytes)

// NCT: tiles of 16 output channels (BN = 16 NCT); S: stride
template <int NCT, int S>
__global__ __launch_bounds__(256) void conv_wgrad_igemm_kernel(GwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gw_smem[];
  constexpr int BN = 16 * NCT, XS = GW_XS, GS = gw_gs(BN), NW = 4, PB = GW_PB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, g = lane >> 4;
  const int PW = a.PW, npix = a.PH * PW, TH = a.TH;
  __bf16* const xs = reinterpret_cast<__bf16*>(gw_smem);
  __bf16* const gs = xs + (size_t)npix * XS;

  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.swz) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = lin & 7u, r = lin >> 3;
    const unsigned nsib = gridDim.y * gridDim.z, g = r % nsib;
    bx = (int)((r / nsib) * 8u + xcd);
    by = (int)(g % gridDim.y);
    bz = (int)(g / gridDim.y);
  }
  const int st = bx % a.nstrip, slot = bx / a.nstrip;
  const int n = 0;   // (goff below is relative to image 0; `shift` adds the image)
  const int tg = by % a.ntg, c0 = (by / a.ntg) * 32;
  const int o0 = bz * BN;
  const int t0 = tg * GW_GT, nt = min(GW_GT, a.ntaps - t0);
  const int i_end = a.OH, j0 = st * 32;
  const int nstage = a.N * a.nrs;

  // patch pieces of this thread (pixel, quarter of the 32 channels): row of the
  // patch and element offset for the block's first stage; -1: column outside
  constexpr int MAXP = 12;
  int prow[MAXP], goff[MAXP];
  const int npiece = npix * 4;
  // (this workgroup's 32 input channels: one tensor's; its channel count is the pitch)
  const bool second = c0 >= a.C1;
  const __bf16* const xsrc = second ? a.x2 + (c0 - a.C1) : a.x + c0;
  const int xpitch = second ? a.Cin - a.C1 : a.C1;
  {
    const int ix0 = j0 * S + a.dx0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int idx = tid + 256 * k;
      const int pix = idx >> 2, q = idx & 3;
      const int py = pix / PW, px = pix - py * PW;
      const int ix = ix0 + px;
      const bool ok = idx < npiece && ix >= 0 && ix < a.W;
      prow[k] = ok ? py : -1;
      goff[k] = ((n * a.H + py) * a.W + ix) * xpitch + 8 * q;
    }
  }
  f32x4 acc[PB][NCT];
#pragma unroll
  for (int j = 0; j < PB; ++j)
#pragma unroll
    for (int m = 0; m < NCT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // the wave's pairs: LDS element offset of the tap inside the patch + the tile
  int poff[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int p = wave + j * NW;
    const int tl = p >> 1, c = p & 1;
    poff[j] = -1;
    if (tl < nt)
      poff[j] = ((a.tdy[t0 + tl] - a.dy0) * PW + (a.tdx[t0 + tl] - a.dx0)) * XS + 16 * c;
  }

  for (int sg = slot; sg < nstage; sg += a.PS) {
    const int ni = sg / a.nrs, i0 = (sg - ni * a.nrs) * TH;
    __syncthreads();  // (the previous stage's fragments have been read)
    // ---- input patch: rows i0 * S + dy0 + py ---------------------------------
    {
      const int iy0 = i0 * S + a.dy0;
      const int shift = (iy0 + (ni - n) * a.H) * a.W * xpitch;
      u32x4 pv[MAXP];
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        pv[k] = zero4;
        const int iy = iy0 + prow[k];
        if (prow[k] >= 0 && iy >= 0 && iy < a.H)
          pv[k] = *reinterpret_cast<const u32x4*>(xsrc + (size_t)(goff[k] + shift));
      }
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        const int idx = tid + 256 * k;
        if (idx < npiece)
          *reinterpret_cast<u32x4*>(xs + (size_t)(idx >> 2) * XS + (idx & 3) * 8) = pv[k];
      }
    }
    // ---- gy tile: TH x 32 pixels x BN channels ----------------------------------
    {
      constexpr int PPX = BN / 8;        // 16-byte pieces per pixel
      const int ngp = TH * 32 * PPX;
      for (int base = 0; base < ngp; base += 256 * 8) {
        u32x4 gv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int piece = base + tid + 256 * k;
          const int pp = piece / PPX, part = piece - pp * PPX;
          const int r = pp >> 5, c = pp & 31;
          const int oy = i0 + r, ox = j0 + c;
          gv[k] = zero4;
          if (piece < ngp && oy < i_end && ox < a.OW)
            gv[k] = *reinterpret_cast<const u32x4*>(
                a.gy + (((size_t)ni * a.OH + oy) * a.OW + ox) * a.Cout + o0 + 8 * part);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int piece = base + tid + 256 * k;
          const int pp = piece / PPX, part = piece - pp * PPX;
          if (piece < ngp) *reinterpret_cast<u32x4*>(gs + (size_t)pp * GS + 8 * part) = gv[k];
        }
      }
    }
    __syncthreads();
    const int nrow = min(TH, i_end - i0);
    for (int r = 0; r < nrow; ++r) {
      bf16x8 af[NCT];
#pragma unroll
      for (int m = 0; m < NCT; ++m) af[m] = gw_tr_frag<GS>(gs + (size_t)(r * 32) * GS + 16 * m, t, g);
      const __bf16* const xrow = xs + (size_t)(r * S) * PW * XS;
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        if (poff[j] >= 0) {   // (wave-uniform)
          const bf16x8 bf = gw_tr_frag<XS * S>(xrow + poff[j], t, g);
#pragma unroll
          for (int m = 0; m < NCT; ++m)
            acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf, acc[j][m], 0, 0, 0);
        }
      }
    }
  }
  // accumulator of lane (t, g), register r: co = 16 m + 4 g + r, ci = 16 c + t
  const size_t nout = (size_t)a.Cout * a.Cin * a.khw;
  float* const out = a.part + (size_t)bx * nout;
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int p = wave + j * NW;
    const int tl = p >> 1, c = p & 1;
    if (tl < nt) {
      // (the tap's place in the layer's kernel: ky * KW + kx, kept in tdy/tdx order)
      const int tap = t0 + tl;
#pragma unroll
      for (int m = 0; m < NCT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((size_t)tap * a.Cout + o0 + 16 * m + 4 * g + r) * a.Cin + c0 + 16 * c + t] =
              acc[j][m][r];   // [tap][co][ci]: the 16 lanes of a row write 64 contiguous bytes
    }
  }
}

// out[co][ci][tap] = sum over the pixel blocks' partials [blk][tap][co][ci]
// (lsi_conv_wgrad.hip's fold; the transposition happens on the small result)
// cl: the parameter (and so its gradient) has torch's channels-last strides --
// [co][tap][ci] in memory -- instead of [co][ci][tap]
__global__ __launch_bounds__(1024) void conv_wgrad_fold_kernel(const float* part, int nblk,
                                                               int nout, float* out, int khw,
                                                               int Cin, int cl) {
  __shared__ float red[16][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  float s = 0.0f;
  if (o < nout) {
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int w = grp;
    for (; w + 48 < nblk; w += 64) {
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
    const int per = nout / khw;            // Cout * Cin
    const int tap = o / per, rest = o - tap * per;
    if (cl) {
      const int co = rest / Cin, ci = rest - co * Cin;
      out[((size_t)co * khw + tap) * Cin + ci] = v;
    } else {
      out[(size_t)rest * khw + tap] = v;
    }
  }
}

// The same fold for layers with many channels (Cout * Cin / 64 workgroups fill
// the chip): workgroup = (co, 64 input channels), all taps.  The partials
// [blk][tap][co][ci] are read in 256-byte runs per (block, tap); the sums go
// through LDS and leave as ONE contiguous run of 64 * khw floats of the
// parameter's layout [co][ci][tap] (the fold above writes 4-byte pieces khw
// floats apart: 9 - 16 visits per cache line, ~70 us for a 512 x 512 x 3 x 3
// layer whatever its map size; this one: a plain stream).
__global__ __launch_bounds__(256) void conv_wgrad_fold_t_kernel(const float* part, int nblk,
                                                                int Cout, int Cin, int khw,
                                                                float* out, int cl) {
  extern __shared__ float ft_tile[];  // [64][khw + 1]
  const int ncc = Cin / 64;
  const int co = blockIdx.x / ncc, ci0 = (blockIdx.x - co * ncc) * 64;
  const int l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t nout = (size_t)khw * Cout * Cin;
  for (int tap = grp; tap < khw; tap += 4) {
    const float* p = part + ((size_t)tap * Cout + co) * Cin + ci0 + l;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = 0;
    for (; w + 3 < nblk; w += 4) {
      s0 += p[(size_t)w * nout];
      s1 += p[(size_t)(w + 1) * nout];
      s2 += p[(size_t)(w + 2) * nout];
      s3 += p[(size_t)(w + 3) * nout];
    }
    for (; w < nblk; ++w) s0 += p[(size_t)w * nout];
    ft_tile[l * (khw + 1) + tap] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (cl) {   // [co][tap][ci]: a 256-byte run per tap
    for (int i = threadIdx.x; i < 64 * khw; i += 256) {
      const int tap = i >> 6, c = i & 63;
      out[((size_t)co * khw + tap) * Cin + ci0 + c] = ft_tile[c * (khw + 1) + tap];
    }
    return;
  }
  float* o = out + ((size_t)co * Cin + ci0) * khw;
  for (int i = threadIdx.x; i < 64 * khw; i += 256) {
    const int c = i / khw, tap = i - c * khw;
    o[i] = ft_tile[c * (khw + 1) + tap];
  }
}

bool gw_desc_ok(const LsiConvDesc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0) return false;
  if (d->Cin <= 0 || d->Cout <= 0 || d->Cin % 32 || d->Cout % 32) return false;
  if (d->KH < 1 || d->KW < 1 || d->KH > 7 || d->KW > 7) return false;
  if (d->stride != 1 && d->stride != 2) return false;
  if (d->pad_t < 0 || d->pad_l < 0 || d->pad_t >= d->KH || d->pad_l >= d->KW) return false;
  if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31)) return false;
  if ((int64_t)d->N * d->OH * d->OW * d->Cout >= (1ll << 31)) return false;
  return true;
}

// Tile rows per stage, rows per pixel block, LDS bytes; false: not taken
// (partial sums too large: small maps with many channels stay on the library).
bool gw_plan(const LsiConvDesc* d, GwArgs& k, int* nct_out, size_t* lds_out, int* nblk_out) {
  const int nct = (d->Cout % 64 == 0) ? 4 : 2;
  const int bn = 16 * nct, gs = gw_gs(bn), s = d->stride;
  k.ntaps = d->KH * d->KW;
  k.khw = k.ntaps;
  k.ntg = (k.ntaps + GW_GT - 1) / GW_GT;
  int nt = 0;
  for (int y = 0; y < d->KH; ++y)
    for (int x = 0; x < d->KW; ++x) {
      k.tdy[nt] = (signed char)(y - d->pad_t);
      k.tdx[nt] = (signed char)(x - d->pad_l);
      ++nt;
    }
  k.dy0 = -d->pad_t; k.dx0 = -d->pad_l;
  k.PW = 31 * s + d->KW;
  int th = 0;
  for (int cand = 8; cand >= 1; cand >>= 1) {
    const int ph = (cand - 1) * s + d->KH;
    if (ph * k.PW > 768) continue;
    const size_t lds = (size_t)ph * k.PW * GW_XS * 2 + (size_t)cand * 32 * gs * 2;
    if (lds > GW_LDS_CAP) continue;
    th = cand; k.PH = ph; *lds_out = lds;
    break;
  }
  if (!th) return false;
  k.TH = th;
  k.nstrip = (d->OW + 31) / 32;
  // workgroups per strip: as many as keep the whole launch resident (512 = two per
  // CU), at most one per stage, partial sums bounded
  const long chan_wgs = (long)(d->Cin / 32) * k.ntg * (d->Cout / bn);
  const size_t wbytes = (size_t)d->Cout * d->Cin * k.khw * sizeof(float);
  k.nrs = (d->OH + th - 1) / th;
  const long nstage = (long)d->N * k.nrs;
  long ps = 512 / (chan_wgs * k.nstrip);
  if (ps < 1) ps = 1;
  if (ps > nstage) ps = nstage;
  while (ps > 1 && (size_t)(ps * k.nstrip) * wbytes > GW_PART_CAP) --ps;
  // (the sibling swizzle needs ps * nstrip % 8 == 0: the largest such ps, if it
  // keeps at least three quarters of the workgroups)
  k.swz = 0;
  {
    static const char* env = getenv("LSI_WGRAD_SWZ");   // experiments
    const int mode = env ? atoi(env) : 1;   // 0 off, 1 on, 2 stride-1 layers only
    const bool want = mode != 0 && chan_wgs > 1 && (mode != 2 || s == 1);
    if (want) {
      long q = ps;
      while (q >= 1 && (q * k.nstrip) % 8) --q;
      if (q >= 1 && 4 * q >= 3 * ps) { ps = q; k.swz = 1; }
    }
  }
  k.PS = (int)ps;
  const long nblk = ps * k.nstrip;
  if ((size_t)nblk * wbytes > GW_PART_CAP || nblk > 65535 * 32L) return false;
  *nblk_out = (int)nblk;
  *nct_out = nct;
  return true;
}

}  // namespace

extern "C" size_t lsi_conv2d_wgrad_workspace_bytes(const LsiConvDesc* d) {
  if (!gw_desc_ok(d)) return 0;
  GwArgs k;
  int nct, nblk;
  size_t lds;
  if (!gw_plan(d, k, &nct, &lds, &nblk)) return 0;
  return (size_t)nblk * d->Cout * d->Cin * d->KH * d->KW * sizeof(float);
}

static int gw_run(const LsiConvDesc* d, const void* x, const void* x2, int c1, const void* gy,
                  float* g_weight, void* workspace, size_t workspace_bytes,
                  lsi_stream_t stream_, int cl = 0);

extern "C" int lsi_conv2d_wgrad(const LsiConvDesc* d, const void* x, const void* gy,
                                float* g_weight, void* workspace, size_t workspace_bytes,
                                lsi_stream_t stream_) {
  return gw_run(d, x, nullptr, 0, gy, g_weight, workspace, workspace_bytes, stream_);
}

extern "C" int lsi_conv2d_wgrad_cat(const LsiConvDesc* d, const void* x1, const void* x2,
                                    int32_t c1, const void* gy, float* g_weight,
                                    int32_t weight_layout, void* workspace,
                                    size_t workspace_bytes, lsi_stream_t stream_) {
  if (weight_layout != 0 && weight_layout != 2) return LSI_EINVAL;
  if (x2 && (!d || c1 <= 0 || c1 >= d->Cin || c1 % 32 || ((uintptr_t)x2 & 15))) return LSI_EINVAL;
  return gw_run(d, x1, x2, x2 ? c1 : 0, gy, g_weight, workspace, workspace_bytes, stream_,
                weight_layout);
}

static int gw_run(const LsiConvDesc* d, const void* x, const void* x2, int c1, const void* gy,
                  float* g_weight, void* workspace, size_t workspace_bytes,
                  lsi_stream_t stream_, int cl) {
  if (!d || !x || !gy || !g_weight || !workspace) return LSI_ENULL;
  if (!gw_desc_ok(d)) return LSI_EUNSUPPORTED;
  if (((uintptr_t)x & 15) || ((uintptr_t)gy & 15) || ((uintptr_t)workspace & 15)) return LSI_EINVAL;
  GwArgs k;
  memset(&k, 0, sizeof(k));
  int nct, nblk;
  size_t lds;
  if (!gw_plan(d, k, &nct, &lds, &nblk)) return LSI_EUNSUPPORTED;
  const size_t nout = (size_t)d->Cout * d->Cin * d->KH * d->KW;
  if (workspace_bytes < (size_t)nblk * nout * sizeof(float)) return LSI_EWORKSPACE;
  if (nout >= (1u << 31)) return LSI_EUNSUPPORTED;
  k.x = (const __bf16*)x; k.gy = (const __bf16*)gy; k.part = (float*)workspace;
  k.x2 = (const __bf16*)x2; k.C1 = x2 ? c1 : d->Cin;
  k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout;
  hipStream_t stream = (hipStream_t)stream_;
  const void* fn = nullptr;
  if (nct == 4 && d->stride == 1) fn = (const void*)conv_wgrad_igemm_kernel<4, 1>;
  if (nct == 2 && d->stride == 1) fn = (const void*)conv_wgrad_igemm_kernel<2, 1>;
  if (nct == 4 && d->stride == 2) fn = (const void*)conv_wgrad_igemm_kernel<4, 2>;
  if (nct == 2 && d->stride == 2) fn = (const void*)conv_wgrad_igemm_kernel<2, 2>;
  if (!fn) return LSI_EINVAL;
  if (lsi_ensure_dynamic_lds(fn, lds) != LSI_OK) return LSI_ELAUNCH;
  const dim3 grid(nblk, (d->Cin / 32) * k.ntg, d->Cout / (16 * nct));
  if (grid.y > 65535 || grid.z > 65535) return LSI_EINVAL;
  void* kargs[1] = {&k};
  if (hipLaunchKernel(fn, grid, dim3(256), kargs, lds, stream) != hipSuccess) return LSI_ELAUNCH;
  if (hipGetLastError() != hipSuccess) return LSI_ELAUNCH;
  const int khw = d->KH * d->KW;
  if (d->Cin % 64 == 0 && (long)d->Cout * (d->Cin / 64) >= 1024)
    hipLaunchKernelGGL(conv_wgrad_fold_t_kernel, dim3((unsigned)(d->Cout * (d->Cin / 64))), dim3(256),
                       (size_t)64 * (khw + 1) * sizeof(float), stream, k.part, nblk, d->Cout,
                       d->Cin, khw, g_weight, cl);
  else
    hipLaunchKernelGGL(conv_wgrad_fold_kernel, dim3((unsigned)((nout + 63) / 64)), dim3(1024), 0,
                       stream, k.part, nblk, (int)nout, g_weight, khw, d->Cin, cl);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
