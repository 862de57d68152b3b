// LSI_PATH_TILE: the any-pose SWEEP kernel
// (general 3-D poses, BASELINE config 4).  Reference semantics: ldi.py:97-184
// (forward_splat) with sampling.py:161-241 (splat) for an arbitrary 4x4 matrix.
//
// The gather kernel (lsi_splat_tile.hip) bins records and walks bin lists
// between chunk barriers; it stays for > 16 composed layers.  Here nothing is binned
// and nothing waits at a barrier while pixels are in flight:
//
// * Workgroup = (tile of TH x TW target cells, batch element[, layer]); the
//   tile's r/g/b/w sums live in LDS (64 KB at 32 x 128 cells).
// * Work item = (source row, layer, group of four 64-pixel segments) with a
//   4-bit mask of the segments whose target rows can intersect the tile (Y is
//   linear-fractional in x and d: extremes at a segment's four (x,
//   disparity-range) corners; the per-layer disparity range comes from
//   disp_range_kernel).  A prologue writes the items that have any such
//   segment to an LDS list in (row, layer) order.  Short segments follow a
//   tilted / keystoned row closely: ~5 % more pixels are projected than land
//   in the tile (whole rows: 20 %).
// * The 16 waves draw items by ticket (no barriers; the hardware favours the
//   oldest waves of a SIMD, so a static split leaves the last waves running
//   long after the first).  16 lanes own a segment, each lane 4 consecutive
//   pixels (16-byte loads, unconditional, the next item's in flight while this
//   one is processed).  Ticket t maps to item (t % 16) * n/16 + t / 16: the 16
//   most recent tickets are in 16 bands of source rows several rows apart, so
//   concurrent waves rarely meet in the tile.
// * Every pixel is projected with the exact index arithmetic of the other
//   paths (the two exact quotients share one refined reciprocal: div2_rn) and
//   each corner is added to its tile cell by a plain LDS read-modify-write of a
//   cell the lane has TAKEN: the cell is its own lock (take2 below: an 8-byte
//   exchange that swaps the cell's (b, w) half with a mark and returns what was
//   there, one LDS round trip for lock + read; the 16-byte store of the sums
//   gives the cell back).  The two left corners (x0, y0) and (x0, y0+1) are
//   taken by ONE `ds_wrxchg2st64_rtn_b64`, then the two right ones: lanes of a
//   wave are >= 1 cell apart in x, so lanes do not collide with themselves.
//   Integer LDS exchanges retire ~20x faster than `ds_add_f32`
//   (tools/microbench2.hip).  A lane never waits while it holds a cell (take
//   both, add where taken, give back, retry what failed): no deadlock for any
//   input, any collision pattern is merely slower.
// * Tile cells are stored even/odd interleaved within a row: lanes two cells
//   apart (trg_downsampling 0.5 x 4 pixels) hit consecutive 16-byte slots.
// * Epilogue per cell: background, normalisation (ldi.py:122-125, 157-182).
// Summation order within a cell is not run-to-run deterministic (like ATOMIC
// and the gather kernel).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"
#include "lsi_splat_internal.h"

#pragma clang fp contract(off)

#ifndef LSI_STREAM_HOOKS
#define LSI_STREAM_HOOKS 0
#endif

using namespace lsi;

namespace {

constexpr int SWEEP_CAP = 4096; // item-list entries in LDS (candidates per chunk)
#ifndef LSI_SWEEP_T
#define LSI_SWEEP_T 1024
#endif
#ifndef LSI_SWEEP_WGS
#define LSI_SWEEP_WGS 1
#endif
constexpr int SWEEP_T = LSI_SWEEP_T;   // threads per workgroup
constexpr int SWEEP_NW = SWEEP_T / 64;  // waves
constexpr int SWEEP_BPW = (SWEEP_CAP / 64 + SWEEP_NW - 1) / SWEEP_NW;  // blocks per wave
constexpr int SEGW = 64;        // source pixels per item (16 lanes x 4)
// Tile rows in LDS: one dummy row above the tile's first row, one below its
// last (row -1 takes the upper corners of pixels whose lower corners are the
// tile's first row).
constexpr int ROW0 = 1;

struct SweepCfg {
  int th;          // nominal tile height (cells): equal-height tiles
  int thmax;       // tile rows allocated in LDS (adaptive tiles may be taller)
  int nty;         // tiles per column of tiles
  int adaptive;    // 1: tile rows cut by source-row count (see the kernel)
  int tiles_x;
  int nq4;         // groups of four 64-pixel segments per source row
  int all_layers;  // 1: compose, every layer sums into the tile; 0: grid.z = layer
  int stream_out;  // 1: 16-byte aligned outputs, Wt % 4 == 0: whole-line streaming stores
  float inv_nq4, inv_nlw;
};

// n / d for 0 <= n < 2^22, d > 0, rcp = fl(1/d)
__device__ __forceinline__ int div_small(int n, int d, float rcp) {
  int q = (int)((float)n * rcp);
  const int r = n - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// A tile cell is its own lock.  A cell is 16 bytes (r, g | b, w); the w word of a
// cell that a lane has taken holds the mark 0xffffffff (never a stored sum: see
// keep_w).  TAKE = one 8-byte exchange of the (b, w) half with the mark -- the
// returned w tells whether the lane got the cell, and the old b, w come with it
// -- followed by a read of the (r, g) half (the LDS executes a wave's
// instructions in order, so the read sees what the exchange saw).  GIVE BACK =
// the 16-byte store of the updated cell.  One LDS round trip and two
// instructions per cell pair less than lock words next to the cells (round 2-4:
// exchange on a lock word, wait, read, wait, write, unlock), and no lock array.
constexpr unsigned CELL_TAKEN = 0xffffffffu;
// (the wave's lane mask of a condition without materialising it in a VGPR)
__device__ __forceinline__ unsigned long long any_lane(bool x) {
  return __builtin_amdgcn_ballot_w64(x);
}

// Takes the cells at LDS byte addresses `a` and `a + 512 * OFF` (the cell one
// tile row below).  bw = {b0, w0, b1, w1}, rg = {r0, g0, r1, g1}.
template <int OFF>
__device__ __forceinline__ void take2(unsigned a, f32x4& bw, f32x4& rg) {
  const unsigned long long mark = ~0ull;
  const unsigned a8 = a + 8u;
  asm volatile(
      "ds_wrxchg2st64_rtn_b64 %0, %2, %4, %4 offset1:%5\n\t"
      "ds_read2st64_b64 %1, %3 offset1:%5\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(bw), "=&v"(rg)
      : "v"(a8), "v"(a), "v"(mark), "n"(OFF)
      : "memory");
}
__device__ __forceinline__ void take1(unsigned a, f32x2& bw, f32x2& rg) {
  const unsigned long long mark = ~0ull;
  asm volatile(
      "ds_wrxchg_rtn_b64 %0, %2, %3 offset:8\n\t"
      "ds_read_b64 %1, %2\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(bw), "=&v"(rg)
      : "v"(a), "v"(mark)
      : "memory");
}
// The common case of a cell pair as ONE straight-line block (no branch, no
// compiler-made exec juggling: a wave issues in order, four waves per SIMD, so
// every instruction of the loop is on some wave's critical path): the lanes of
// `need` take the cell at `a` and the one `ROWB` bytes below it, add V * wa /
// V * wb where they got the cell, and give it back -- (r, g) first, then (b, w),
// whose store is what releases the cell.  Returns the lane masks of the cells
// NOT got (left to the caller's retry loop: rare).  A taken cell reads back as
// the all-ones mark in b and w: one 64-bit compare.
template <int ROWB>
__device__ __forceinline__ void pair_add(unsigned long long need, unsigned a, f32x2 vxy,
                                         f32x2 vzw, f32x2 wa, f32x2 wb,
                                         unsigned long long& fa, unsigned long long& fb) {
  const unsigned long long mark = ~0ull;
  const unsigned a8 = a + 8u;
  f32x2 bwa, bwb, rga, rgb;
  unsigned long long sv, nx;
  asm volatile(
      "s_and_saveexec_b64 %[sv], %[need]\n\t"
      "ds_wrxchg_rtn_b64 %[bwa], %[a8], %[mk]\n\t"
      "ds_wrxchg_rtn_b64 %[bwb], %[a8], %[mk] offset:%[rowb]\n\t"
      "ds_read_b64 %[rga], %[a]\n\t"
      "ds_read_b64 %[rgb], %[a] offset:%[rowb]\n\t"
      "s_mov_b64 %[nx], exec\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_cmp_eq_u64_e64 %[fa], -1, %[bwa]\n\t"
      "v_cmp_eq_u64_e64 %[fb], -1, %[bwb]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_andn2_b64 exec, %[nx], %[fa]\n\t"
      "v_pk_fma_f32 %[rga], %[vxy], %[wa], %[rga] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[bwa], %[vzw], %[wa], %[bwa] op_sel_hi:[1,0,1]\n\t"
      "ds_write_b64 %[a], %[rga]\n\t"
      "ds_write_b64 %[a8], %[bwa]\n\t"
      "s_andn2_b64 exec, %[nx], %[fb]\n\t"
      "v_pk_fma_f32 %[rgb], %[vxy], %[wb], %[rgb] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[bwb], %[vzw], %[wb], %[bwb] op_sel_hi:[1,0,1]\n\t"
      "ds_write_b64 %[a], %[rgb] offset:%[rowb]\n\t"
      "ds_write_b64 %[a8], %[bwb] offset:%[rowb]\n\t"
      "s_mov_b64 exec, %[sv]"
      : [bwa] "=&v"(bwa), [bwb] "=&v"(bwb), [rga] "=&v"(rga), [rgb] "=&v"(rgb),
        [sv] "=&s"(sv), [nx] "=&s"(nx), [fa] "=&s"(fa), [fb] "=&s"(fb)
      : [need] "s"(need), [a] "v"(a), [a8] "v"(a8), [mk] "v"(mark), [vxy] "v"(vxy),
        [vzw] "v"(vzw), [wa] "v"(wa), [wb] "v"(wb), [rowb] "n"(ROWB)
      : "memory", "vcc", "scc");
}

__device__ __forceinline__ bool got_cell(float w) { return __float_as_uint(w) != CELL_TAKEN; }
// A stored w never carries the mark: only a NaN with that payload could, and
// only from a NaN mask value with it (every other input that is not finite is
// dropped or lands in r, g, b) -- the pixel weight is passed through this (it
// stays a NaN, with the next payload), and so are the sums of the C++ path.
template <bool HAS_MASK>
__device__ __forceinline__ float keep_w(float w) {
  return HAS_MASK ? __uint_as_float(min(__float_as_uint(w), CELL_TAKEN - 1u)) : w;
}

// The two exact quotients q0 / n and q1 / n (IEEE, round to nearest: they feed
// floorf) and the refined reciprocal of n.  When |n|, |q0|, |q1| all lie in
// [2^-60, 2^60] -- every lane of the wave: one branch -- v_div_scale_f32 scales
// nothing and v_div_fixup_f32 passes the quotient through, so the compiler's
// division is this sequence with one reciprocal per quotient; sharing the
// reciprocal gives the same bits with 13 instead of 22 instructions and one
// v_rcp_f32 instead of three (the weight's reciprocal was a third).
__device__ __forceinline__ void div2_rn(float q0, float q1, float n, bool active,
                                        float& o0, float& o1, float& rcp_n) {
  const float hi = fmaxf(fmaxf(fabsf(n), fabsf(q0)), fabsf(q1));
  const float lo = fminf(fminf(fabsf(n), fabsf(q0)), fabsf(q1));
  const bool plain = hi < 0x1p60f && lo > 0x1p-60f;  // (false for NaN)
  float r = __builtin_amdgcn_rcpf(n);
  r = __fmaf_rn(__fmaf_rn(-n, r, 1.0f), r, r);
  rcp_n = r;
  if (any_lane(active && !plain) == 0ull) {
    float a = q0 * r;
    a = __fmaf_rn(__fmaf_rn(-n, a, q0), r, a);
    o0 = __fmaf_rn(__fmaf_rn(-n, a, q0), r, a);
    float b = q1 * r;
    b = __fmaf_rn(__fmaf_rn(-n, b, q1), r, b);
    o1 = __fmaf_rn(__fmaf_rn(-n, b, q1), r, b);
  } else {
    o0 = div_rn(q0, n);
    o1 = div_rn(q1, n);
  }
}

template <int TWL, bool VEC4, bool HAS_MASK, bool WANT_DISP>
__global__ __launch_bounds__(SWEEP_T, LSI_SWEEP_WGS) void splat_sweep_kernel(
    SplatArgs a, SweepCfg c, const float2* __restrict__ range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TW = 1 << TWL;
  const LsiSplatDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.y;
  const int l_begin = c.all_layers ? 0 : (int)blockIdx.z;
  const int NLW = c.all_layers ? d.L : 1;  // layers this workgroup sums
  // c.thmax rows of cells are allocated; the tile's own height is decided below
  const int NCP = (c.thmax + ROW0 + 1) << TWL;
  const int tile_y = blockIdx.x / c.tiles_x, tile_x = blockIdx.x - tile_y * c.tiles_x;
  const int tx0 = tile_x * TW;
  const int Ht = d.Ht, Wt = d.Wt, H = d.H, W = d.W;

  // r g b w sums per cell, rows -1 .. thmax (slot row r + 1); a cell is its own
  // lock (take2 above)
  float4* tile = reinterpret_cast<float4*>(smem);   // [thmax + ROW0 + 1][TW]
  int* list = reinterpret_cast<int*>(tile + NCP);   // [SWEEP_CAP] accepted items
  int* cnt = list + SWEEP_CAP;                      // [64] per-block counts
  int* ctl = cnt + 64;                              // [1] list fill
  float2* lrange = reinterpret_cast<float2*>(ctl + 4);  // [LSI_SWEEP_MAXL]
  // WANT_DISP: sum of (target disparity * weight) per cell, next to the tile
  float* dsum = reinterpret_cast<float*>(lrange + LSI_SWEEP_MAXL);  // [thmax + ROW0 + 1][TW]
  const unsigned tile_addr = (unsigned)(uintptr_t)tile;

  float m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = a.M[16 * b + k];
  const float s = d.trg_downsampling, zscale = d.zbuf_scale;
  const float inv_md = div_rn(1.0f, d.max_disp);
  // exp((c - 0.5) * scale) = exp2(c * zA + zB)
  const float zA = zscale * 1.44269504f, zB = -0.5f * zscale * 1.44269504f;
  const float tx0f = (float)tx0;
  const int tw_eff = min(TW, Wt - tx0);
  const float ax_lo = (float)(tx0 - 1), ax_hi = (float)(tx0 + tw_eff - 1);

  for (int i = tid; i < NCP; i += SWEEP_T) {
    tile[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (WANT_DISP) dsum[i] = 0.0f;
  }
  if (tid < NLW) {  // the layer's disparity range: fold the row slices
    float2 dr = make_float2(__builtin_inff(), -__builtin_inff());
    for (int k = 0; k < LSI_RANGE_SLICES; ++k) {
      const float2 r = range[((size_t)(l_begin + tid) * d.B + b) * LSI_RANGE_SLICES + k];
      dr.x = fminf(dr.x, r.x);
      dr.y = fmaxf(dr.y, r.y);
    }
    lrange[tid] = dr;
  }
  __syncthreads();

  // ---- the tile's target rows [ty0, ty0 + th_eff).  Equal-height tiles get
  // unequal numbers of source pixels under a keystone (config 4: the heaviest
  // tile 1.37x the mean, and the heaviest tile is the kernel's time), so the
  // nty tiles of a column split the rows by SOURCE rows instead: a histogram of
  // where the centre pixel of every source row lands (mid disparity), cut into
  // nty equal parts, each tile at most c.thmax rows (config 4: 135 -> 122 us).  Every workgroup of the
  // batch element computes the same cuts from the same data.
  int ty0 = tile_y * c.th, ty1 = min(ty0 + c.th, Ht);
  if (c.adaptive) {
    int* hist = list;                 // [Ht] (Ht <= SWEEP_CAP), then its prefix
    int* cut = cnt;                   // [nty + 1] (nty <= 63)
    for (int i = tid; i < Ht; i += SWEEP_T) hist[i] = 0;
    if (tid == 0) ctl[2] = 0;
    __syncthreads();
    float dlo = __builtin_inff(), dhi = -__builtin_inff();
    for (int k = 0; k < NLW; ++k) {
      dlo = fminf(dlo, lrange[k].x);
      dhi = fmaxf(dhi, lrange[k].y);
    }
    const float dm = 0.5f * (dlo + dhi);
    if (finite_f(dm)) {
      // 8 sample columns per source row; a sample counts where it lands if it
      // lands inside the target image (what falls outside costs almost nothing)
      for (int i = tid; i < H * 8; i += SWEEP_T) {
        const int y = i >> 3, k = i & 7;
        const float py = (float)y + 0.5f;
        const float pxc = ((float)k + 0.5f) * (float)W * 0.125f;
        const float n = safe_den(mrow(m, 2, pxc, py, dm));
        const float t = floorf(div_rn(mrow(m, 1, pxc, py, dm), n) * s - 0.5f);
        const float u = floorf(div_rn(mrow(m, 0, pxc, py, dm), n) * s - 0.5f);
        if (t >= 0.0f && t < (float)Ht && u >= -1.0f && u < (float)Wt)
          atomicAdd(&hist[(int)t], 1);
      }
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: inclusive prefix, lane = a run of rows
      const int per = (Ht + 63) >> 6;
      const int r0 = min(lane * per, Ht), r1 = min(r0 + per, Ht);
      int sum = 0;
      for (int r = r0; r < r1; ++r) sum += hist[r];
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      int run = incl - sum;
      for (int r = r0; r < r1; ++r) { run += hist[r]; hist[r] = run; }
      if (lane == 63) ctl[2] = incl;
    }
    __syncthreads();
    const int total = ctl[2];
    if (tid <= c.nty) {
      // first row whose prefix reaches k * total / nty (binary search), then
      // clamped so that every tile has 1 .. thmax rows and the rest still fits
      int want = 0;
      if (tid > 0 && tid < c.nty && total > 0) {
        const long target = ((long)total * tid + c.nty - 1) / c.nty;
        int lo = 0, hi = Ht - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (hist[mid] >= target) hi = mid; else lo = mid + 1;
        }
        want = lo + 1;
      } else if (tid > 0) {
        want = tid < c.nty ? min(tid * c.th, Ht) : Ht;
      }
      cut[tid] = want;
    }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < c.nty; ++k) {
        const int lo = max(cut[k - 1] + 1, Ht - (c.nty - k) * c.thmax);
        const int hi = min(cut[k - 1] + c.thmax, Ht - (c.nty - k));
        cut[k] = total > 0 ? min(max(cut[k], lo), hi)
                           : min(k * c.th, Ht);
      }
      cut[c.nty] = Ht;
    }
    __syncthreads();
    ty0 = cut[tile_y];
    ty1 = cut[tile_y + 1];
    __syncthreads();  // (list / cnt are reused below)
  }
  const int th_eff = ty1 - ty0;
  const float ty0f = (float)ty0;
  // acceptance window of the top-left cell (x0, y0) around the tile
  const float ay_lo = (float)(ty0 - 1), ay_hi = (float)(ty0 + th_eff - 1);

#if LSI_STREAM_HOOKS
  // reserved & 4: cycles per wave in [0] issue [1] projection + weights
  // [2] left pair [3] right pair [4] whole loop [5] prologue, written behind
  // the range slices in the workspace (tools/sweep_probe.py)
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tprev = __builtin_readcyclecounter();
  const bool prof = (d.reserved & 4) != 0;
#define SWEEP_STAMP(i) do { if (prof) { const long long tn = __builtin_readcyclecounter(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
#else
#define SWEEP_STAMP(i)
#endif
  struct Px {
    float4 dv, ta, tb, tc, mv;  // 4 disparities, 4 x rgb, 4 masks
    int y, x;  // source row, first pixel of this lane (x < 0: nothing)
  };

  const int wave = tid >> 6;
  // Composition with the disparity output needs every layer's own normalised
  // disparity (ldi.py:157-182): the layers are then swept one after the other
  // and folded into per-thread totals (a thread owns cells tid + k * 1024).
  constexpr bool LAYER_PASSES = WANT_DISP;
  const int npass = (LAYER_PASSES && c.all_layers) ? d.L : 1;
  const int PLW = (LAYER_PASSES && c.all_layers) ? 1 : NLW;  // layers per pass
  float4 Tsum[4];
  float Tdmax[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { Tsum[k] = make_float4(0.f, 0.f, 0.f, 0.f); Tdmax[k] = 0.0f; }
  for (int lp = 0; lp < npass; ++lp) {
  const int lofs = npass > 1 ? lp : 0;  // first layer of this pass (in lrange)
  const int ncand = PLW * H * c.nq4;
  for (int cand0 = 0; cand0 < ncand; cand0 += SWEEP_CAP) {
    const int cand1 = min(ncand, cand0 + SWEEP_CAP);
    if (tid == 0) ctl[0] = 0;  // the ticket counter (barriers below)
    // ---- work items of this chunk that can reach the tile, in candidate order
    // (row, layer, group of 4 segments): blocks of 64 candidates, 4 per wave ----
    unsigned long long tmask[SWEEP_BPW];
    int tpacked[SWEEP_BPW];
#pragma unroll
    for (int k = 0; k < SWEEP_BPW; ++k) {
      const int blk = wave + SWEEP_NW * k;
      const int cand = blk < SWEEP_CAP / 64 ? cand0 + blk * 64 + lane : cand1;
      int segs = 0;  // segments of the group that can reach the tile
      int packed = 0;
      if (cand < cand1) {
        const int yl = c.nq4 == 1 ? cand : div_small(cand, c.nq4, c.inv_nq4);
        const int q4 = cand - yl * c.nq4;
        const int y = PLW == 1 ? yl : div_small(yl, PLW, c.inv_nlw);
        const int li = lofs + yl - y * PLW;
        const float2 dr = lrange[li];
        const float py = (float)y + 0.5f;
        const bool unbounded = !(dr.x <= dr.y);  // no finite disparity seen
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int x0s = (4 * q4 + g) * SEGW;
          if (x0s >= W) continue;
          const float xa = (float)x0s + 0.5f;
          const float xb = (float)min(x0s + SEGW, W) - 0.5f;
          float vmin = __builtin_inff(), vmax = -__builtin_inff();
          bool wild = unbounded;
          float nsign = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float px = (q & 1) ? xb : xa;
            const float dv = (q & 2) ? dr.y : dr.x;
            const float q1 = mrow(m, 1, px, py, dv);
            const float n = mrow(m, 2, px, py, dv);
            const float v = div_rn(q1, safe_den(n)) * s - 0.5f;
            // Y is linear-fractional in x and d: monotone along both while the
            // denominator keeps its sign on the segment; else no bound holds
            if (!finite_f(v) || !finite_f(n) || n == 0.0f) wild = true;
            if (q == 0) nsign = n;
            else if ((n > 0.0f) != (nsign > 0.0f)) wild = true;
            vmin = fminf(vmin, v);
            vmax = fmaxf(vmax, v);
          }
          // one cell of slack on either side for the rounding of the corners
          if (wild || (floorf(vmax) + 1.0f >= ay_lo && floorf(vmin) - 1.0f <= ay_hi))
            segs |= 1 << g;
        }
        packed = (li << 28) | (q4 << 20) | (segs << 16) | y;
      }
      tmask[k] = __ballot(segs != 0);
      tpacked[k] = packed;
      if (lane == 0 && blk < SWEEP_CAP / 64) cnt[blk] = __popcll(tmask[k]);
    }
    __syncthreads();
    if (wave == 0) {  // exclusive prefix of the 64 block counts
      const int mine = cnt[lane];
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      cnt[lane] = incl - mine;
      if (lane == 63) ctl[1] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SWEEP_BPW; ++k) {
      if ((tmask[k] >> lane) & 1ull)
        list[cnt[wave + SWEEP_NW * k] +
             __builtin_amdgcn_mbcnt_hi((unsigned)(tmask[k] >> 32),
                                       __builtin_amdgcn_mbcnt_lo((unsigned)tmask[k], 0u))] =
            tpacked[k];
    }
    __syncthreads();
    // ---- the waves draw items by ticket (the hardware favours the oldest
    // waves of a SIMD: with a static split the last waves run on long after
    // the first are done).  Ticket t maps to item (t % 16) * zc + t / 16: the
    // list is sorted by row, so the 16 most recent tickets are items in 16
    // bands of source rows several rows apart, and the four lane groups take
    // the four segments of an item: what is in flight lands in different cells
    // and lock collisions stay rare (a collision costs a wave a retry round).
    const int nitem = ctl[1];
    const int zc = (nitem + 15) >> 4;  // items per band
    const int nticket = zc << 4;
    auto next_ticket = [&]() {
      int t = 0;
      if (lane == 0) t = atomicAdd(&ctl[0], 1);
      return __builtin_amdgcn_readfirstlane(t);
    };

    // Decodes the lane's item of step `st` and starts its loads.  The loads are
    // unconditional (lanes without work read the item's first pixels) and their
    // results are not touched until process(): the compiler then waits for
    // them only there, one step later.
    auto issue = [&](Px& p, int t) {
      const int it = (t & 15) * zc + (t >> 4);
      const unsigned item = (unsigned)list[min(it, nitem - 1)];
      const int g = lane >> 4;
      const int l = l_begin + (int)(item >> 28);
      const int y = (int)(item & 0xffffu);
      const int x = ((int)((item >> 20) & 0xffu) * 4 + g) * SEGW + (lane & 15) * 4;
      const bool valid = t < nticket && it < nitem && ((item >> (16 + g)) & 1u) && x < W;
      const int xs = valid ? x : 0;
      p.y = y;
      p.x = valid ? x : -1;
      const float* dp = a.disp + (long)l * d.disp_sl + (long)b * d.disp_sb +
                        (long)y * d.disp_sy + (long)xs * d.disp_sx;
      const float* tp = a.tex + (long)l * d.tex_sl + (long)b * d.tex_sb +
                        (long)y * d.tex_sy + (long)xs * d.tex_sx;
      const float* mp = HAS_MASK ? a.mask + (long)l * d.mask_sl + (long)b * d.mask_sb +
                                       (long)y * d.mask_sy + (long)xs * d.mask_sx
                                 : nullptr;
      if (VEC4) {  // unit strides, channels last, 16-byte aligned rows, W % 4 == 0
        p.dv = *reinterpret_cast<const float4*>(dp);
        p.ta = *reinterpret_cast<const float4*>(tp);
        p.tb = *reinterpret_cast<const float4*>(tp + 4);
        p.tc = *reinterpret_cast<const float4*>(tp + 8);
        if (HAS_MASK) p.mv = *reinterpret_cast<const float4*>(mp);
      } else {
        float dv[4], t[12], mk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = xs + j < W;  // past the row end: mask 0, dropped below
          const int jj = in ? j : 0;
          dv[j] = dp[(long)jj * d.disp_sx];
          const float* tq = tp + (long)jj * d.tex_sx;
          t[3 * j] = tq[0]; t[3 * j + 1] = tq[d.tex_sc]; t[3 * j + 2] = tq[2 * d.tex_sc];
          mk[j] = HAS_MASK ? mp[(long)jj * d.mask_sx] : 1.0f;
          if (!in) mk[j] = 0.0f;
        }
        p.dv = make_float4(dv[0], dv[1], dv[2], dv[3]);
        p.ta = make_float4(t[0], t[1], t[2], t[3]);
        p.tb = make_float4(t[4], t[5], t[6], t[7]);
        p.tc = make_float4(t[8], t[9], t[10], t[11]);
        p.mv = make_float4(mk[0], mk[1], mk[2], mk[3]);
      }
    };

    // Adds V * wa to the cell at slot `cell` (= (row + ROW0) * TW + slot, row in
    // -1 .. TH-1) and V * wb to the cell below it, for the lanes with need ==
    // true.  Both cells are taken and updated together whatever their
    // weights (the reference, too, adds all four corners, zero weights
    // included); rows -1 and TH exist so that no lane needs a special case --
    // what lands there is never read.  A lane never waits while it holds a
    // cell: it updates and gives back what it got, then retries what it did not
    // get, one cell at a time.
    auto locked_pair = [&](bool need, int cell, float wa, float wb,
                           const float4& V, float vd) {
#if LSI_STREAM_HOOKS
      if (d.reserved & 512) return;  // timing experiment: no accumulation
#endif
      const unsigned ca = tile_addr + (unsigned)cell * 16u;
      auto put = [&](int c1, float w1, float r, float g, float bb, float ww) {
        if (WANT_DISP) {
          dsum[c1] = __fmaf_rn(vd, w1, dsum[c1]);
          asm volatile("" ::: "memory");  // before the cell is given back
        }
        // (explicit FMAs: a sum, not an index or a threshold)
        tile[c1] = make_float4(__fmaf_rn(V.x, w1, r), __fmaf_rn(V.y, w1, g),
                               __fmaf_rn(V.z, w1, bb),
                               keep_w<HAS_MASK>(__fmaf_rn(V.w, w1, ww)));
      };
      bool na, nb;
      if (!WANT_DISP) {
        unsigned long long fa, fb;
        pair_add<TW * 16>(any_lane(need), ca, f32x2{V.x, V.y}, f32x2{V.z, V.w},
                          f32x2{wa, wa}, f32x2{wb, wb}, fa, fb);
        if ((fa | fb) == 0ull) return;  // (nearly always)
        na = (fa >> lane) & 1ull;
        nb = (fb >> lane) & 1ull;
      } else {
        f32x4 bw, rg;
        bool ga = false, gb = false;
        if (need) {
          take2<TW / 32>(ca, bw, rg);
          ga = got_cell(bw.y);
          gb = got_cell(bw.w);
        }
        if (ga) put(cell, wa, rg.x, rg.y, bw.x, bw.y);
        if (gb) put(cell + TW, wb, rg.z, rg.w, bw.z, bw.w);
        na = need && !ga;
        nb = need && !gb;
      }
      // Lost a cell to another lane: one cell at a time, only cells still needed
      while (any_lane(na || nb) != 0ull) {
        if (na || nb) {
          const int c1 = na ? cell : cell + TW;
          const float w1 = na ? wa : wb;
          f32x2 bw1, rg1;
          take1(tile_addr + (unsigned)c1 * 16u, bw1, rg1);
          if (got_cell(bw1.y)) {
            put(c1, w1, rg1.x, rg1.y, bw1.x, bw1.y);
            if (na) na = false; else nb = false;
          }
        }
      }
    };

    auto process = [&](const Px& p) {
      if (any_lane(p.x >= 0) == 0ull) return;
#if LSI_STREAM_HOOKS
      if (d.reserved & 2048) return;  // timing experiment: loads only
#endif
      const float py = (float)p.y + 0.5f;
      // (the row's share of q = ((px*m0 + py*m1) + m2) + d*m3 -- the product
      // py*m1 rounds the same wherever it is formed: once per item)
      const float pm0 = py * m[1], pm1 = py * m[5], pm2 = py * m[9], pm3 = py * m[13];
      const float dvs[4] = {p.dv.x, p.dv.y, p.dv.z, p.dv.w};
      const float t0s[4] = {p.ta.x, p.ta.w, p.tb.z, p.tc.y};
      const float t1s[4] = {p.ta.y, p.tb.x, p.tb.w, p.tc.z};
      const float t2s[4] = {p.ta.z, p.tb.y, p.tc.x, p.tc.w};
      const float mks[4] = {p.mv.x, p.mv.y, p.mv.z, p.mv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float px = (float)(p.x + j) + 0.5f;
        const float dv = dvs[j];
        const float q1 = ((px * m[4] + pm1) + m[6]) + dv * m[7];
        const float nden = safe_den(((px * m[8] + pm2) + m[10]) + dv * m[11]);
        const float q0 = ((px * m[0] + pm0) + m[2]) + dv * m[3];
        float qx, qy, rn;
        div2_rn(q0, q1, nden, p.x >= 0, qx, qy, rn);
        const float Y = qy * s - 0.5f;
        const float y0 = floorf(Y);
        const float X = qx * s - 0.5f;
        const float x0 = floorf(X);
        // (non-finite X / Y fail the comparisons: dropped, like every path)
        bool ok = p.x >= 0 && y0 >= ay_lo && y0 <= ay_hi && x0 >= ax_lo && x0 <= ax_hi;
        if (any_lane(ok) == 0ull) continue;
#if LSI_STREAM_HOOKS
        if (prof) tacc[5] += 1 + ((long long)__popcll(__ballot(ok)) << 32);  // px-iters, ok lanes
#endif
        const float q3 = ((px * m[12] + pm3) + m[14]) + dv * m[15];
        // The target disparity feeds the weight and the disparity output
        // (1e-4 relative), no index or threshold: the refined reciprocal of the
        // normaliser instead of the IEEE division (the disparity output keeps
        // the exact quotient)
        const float dd = WANT_DISP ? div_rn(q3, nden) : q3 * rn;
        // exp((clip(D/max,0,1) - 0.5) * scale) [D > 0] as exp2 of one fma
        const float xn = dd * inv_md;
        const float ez = __builtin_amdgcn_exp2f(
            __fmaf_rn(__builtin_amdgcn_fmed3f(xn, 0.0f, 1.0f), zA, zB));
        const float zw = xn > 0.0f ? ez : 0.0f;
        const float pw = (VEC4 && !HAS_MASK) ? zw : keep_w<true>(zw * mks[j]);
        ok = ok && pw != 0.0f;  // contributes exactly +0 everywhere
        // Corner weights (sampling.py:193-222).  The border masks are implied:
        // a corner outside the image is outside every tile and never read.
        const float wx0 = (x0 + 1.0f) - X, wx1 = X - x0;
        const float wy0 = (y0 + 1.0f) - Y, wy1 = Y - y0;
        const float p00 = wx0 * wy0, p01 = wx1 * wy0, p10 = wx0 * wy1, p11 = wx1 * wy1;
        const float w00 = p00 > 1e-3f ? p00 : 0.0f, w01 = p01 > 1e-3f ? p01 : 0.0f;
        const float w10 = p10 > 1e-3f ? p10 : 0.0f, w11 = p11 > 1e-3f ? p11 : 0.0f;
        const float4 V = make_float4(t0s[j] * pw, t1s[j] * pw, t2s[j] * pw, pw);
        const int iy = ok ? (int)(y0 - ty0f) : 0;  // -1 .. th_eff - 1
        const int ix = ok ? (int)(x0 - tx0f) : 0;  // -1 .. tw_eff - 1
        // even cells of a row first, odd cells in its second half
        const int ixr = ix + 1;
        const int sl = (ix >> 1) + ((ix & 1) << (TWL - 1));
        const int sr = (ixr >> 1) + ((ixr & 1) << (TWL - 1));
        const int rowb = (iy + ROW0) << TWL;
        SWEEP_STAMP(1);
        locked_pair(ok && ix >= 0, rowb + sl, w00, w10, V, dd * pw);
        SWEEP_STAMP(2);
        locked_pair(ok && ixr < tw_eff, rowb + sr, w01, w11, V, dd * pw);
        SWEEP_STAMP(3);
      }
    };

    if (nitem > 0) {
      Px pa, pb;
      pa.mv = make_float4(1.f, 1.f, 1.f, 1.f);
      pb.mv = pa.mv;
      SWEEP_STAMP(5);
#if LSI_STREAM_HOOKS
      const long long tloop = tprev;
#endif
      int ta = next_ticket();
      issue(pa, ta);
      for (;;) {
        const int tb = next_ticket();
        issue(pb, tb);
        SWEEP_STAMP(0);
        if (ta >= nticket) break;
        process(pa);
        ta = next_ticket();
        issue(pa, ta);
        SWEEP_STAMP(0);
        if (tb >= nticket) break;
        process(pb);
      }
#if LSI_STREAM_HOOKS
      tacc[4] += tprev - tloop;
#endif
    }
    __syncthreads();
  }
  if (npass > 1) {
    // layer lp is complete in the tile: fold it into the totals of the cells
    // this thread owns (as the reference: per-layer canvases bg + sums, the
    // layer's disparity normalised by its own weight, then sum / max over
    // the layers), and clear the tile for the next layer
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cell = tid + k * SWEEP_T;
      if (cell >= (th_eff << TWL)) continue;
      const int cy = cell >> TWL, cx = cell & (TW - 1);
      const int slot = ((cy + ROW0) << TWL) + (cx >> 1) + ((cx & 1) << (TWL - 1));
      const float4 t = tile[slot];
      const float bg = d.bg_wt;
      const float l0 = bg + t.x, l1 = bg + t.y, l2 = bg + t.z, lw = bg + t.w;
      const float dl = div_rn(dsum[slot], safe_den(lw));
      if (lp == 0) {
        Tsum[k] = make_float4(l0, l1, l2, lw);
        Tdmax[k] = dl;
      } else {
        Tsum[k].x += l0; Tsum[k].y += l1; Tsum[k].z += l2; Tsum[k].w += lw;
        Tdmax[k] = fmaxf(Tdmax[k], dl);
      }
    }
    __syncthreads();
    if (lp + 1 < npass) {
      for (int i = tid; i < NCP; i += SWEEP_T) {
        tile[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dsum[i] = 0.0f;
      }
      __syncthreads();
    }
  }
  }  // layer passes
#if LSI_STREAM_HOOKS
  if (prof && lane == 0) {
    long long* o = reinterpret_cast<long long*>(
                       const_cast<float2*>(range) +
                       (size_t)d.B * d.L * LSI_RANGE_SLICES) +
                   (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SWEEP_NW + wave) * 6;
    for (int k = 0; k < 6; ++k) o[k] = tacc[k];
  }
#endif

  // ---- epilogue: background, normalisation (ldi.py:122-125, 157-182) ---------
  const size_t P = (size_t)Ht * Wt;
  const size_t obase = c.all_layers ? (size_t)b * P
                                    : ((size_t)l_begin * d.B + b) * P;
  if (npass > 1) {  // composed totals of the layer passes, from registers
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cell = tid + k * SWEEP_T;
      if (cell >= (th_eff << TWL)) continue;
      const int cy = cell >> TWL, cx = cell & (TW - 1);
      const int gy = ty0 + cy, gx = tx0 + cx;
      if (gx >= Wt) continue;
      const size_t o = obase + (size_t)gy * Wt + gx;
      const float wd = safe_den(Tsum[k].w);
      a.out_img[3 * o + 0] = div_rn(Tsum[k].x, wd);
      a.out_img[3 * o + 1] = div_rn(Tsum[k].y, wd);
      a.out_img[3 * o + 2] = div_rn(Tsum[k].z, wd);
      a.out_wts[o] = Tsum[k].w;
      a.out_disp[o] = Tdmax[k];
    }
    return;
  }
  const float bgs = (float)NLW * d.bg_wt;
  if (!WANT_DISP && c.stream_out && tw_eff == TW) {
    // Whole 128-byte lines per store instruction, non-temporal (the rendered
    // view is written once; see store_stream_f4 in lsi_common.h and the compact
    // stream kernel's epilogue): lane q of a tile row writes floats 4q .. 4q+3
    // of the row's 3 * TW colour floats -- they belong to cells 4q/3 and 4q/3+1.
    constexpr int QR = 3 * TW / 4;  // 16-byte stores per tile row (colours)
    auto cell_at = [&](int cy, int cx) {
      return tile[((cy + ROW0) << TWL) + (cx >> 1) + ((cx & 1) << (TWL - 1))];
    };
    for (int q = tid; q < th_eff * QR; q += SWEEP_T) {
      const int cy = q / QR, j = q - cy * QR;
      const unsigned f = 4u * (unsigned)j;
      const unsigned c0 = __umulhi(f, 0xAAAAAAABu) >> 1;  // f / 3
      const unsigned o3 = f - 3u * c0;
      const float4 A = cell_at(cy, (int)c0);
      const float4 Bc = cell_at(cy, min((int)c0 + 1, TW - 1));
      const float wa = safe_den(A.w + bgs), wb = safe_den(Bc.w + bgs);
      const float ax = div_rn(A.x + bgs, wa), ay = div_rn(A.y + bgs, wa),
                  az = div_rn(A.z + bgs, wa);
      const float bx = div_rn(Bc.x + bgs, wb), by = div_rn(Bc.y + bgs, wb),
                  bz = div_rn(Bc.z + bgs, wb);
      const float v0 = o3 == 0 ? ax : (o3 == 1 ? ay : az);
      const float v1 = o3 == 0 ? ay : (o3 == 1 ? az : bx);
      const float v2 = o3 == 0 ? az : (o3 == 1 ? bx : by);
      const float v3 = o3 == 0 ? bx : (o3 == 1 ? by : bz);
      const size_t o = obase + (size_t)(ty0 + cy) * Wt + tx0;
      store_stream_f4(a.out_img + 3 * o + f, v0, v1, v2, v3);
    }
    for (int q = tid; q < th_eff * (TW / 4); q += SWEEP_T) {
      const int cy = q / (TW / 4), cx = 4 * (q - cy * (TW / 4));
      const size_t o = obase + (size_t)(ty0 + cy) * Wt + tx0 + cx;
      store_stream_f4(a.out_wts + o, cell_at(cy, cx).w + bgs, cell_at(cy, cx + 1).w + bgs,
                      cell_at(cy, cx + 2).w + bgs, cell_at(cy, cx + 3).w + bgs);
    }
    return;
  }
  for (int cell = tid; cell < (th_eff << TWL); cell += SWEEP_T) {
    const int cy = cell >> TWL, cx = cell & (TW - 1);
    const int gy = ty0 + cy, gx = tx0 + cx;
    if (gx >= Wt) continue;
    const int slot = ((cy + ROW0) << TWL) + (cx >> 1) + ((cx & 1) << (TWL - 1));
    const float4 t = tile[slot];
    const size_t o = obase + (size_t)gy * Wt + gx;
    const float w = t.w + bgs;
    const float wd = safe_den(w);
    a.out_img[3 * o + 0] = div_rn(t.x + bgs, wd);
    a.out_img[3 * o + 1] = div_rn(t.y + bgs, wd);
    a.out_img[3 * o + 2] = div_rn(t.z + bgs, wd);
    a.out_wts[o] = w;
    // (one layer per workgroup here: compose with one layer, or per-layer
    // outputs -- the layer's disparity normalised by its own weight)
    if (WANT_DISP) a.out_disp[o] = div_rn(dsum[slot], wd);
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 16-byte loads of 4 consecutive pixels: channels-last texture, unit pixel
// strides, every row / layer / batch offset a multiple of 4 floats.
bool sweep_vec4(const SplatArgs& a) {
  const LsiSplatDesc& d = a.d;
  bool ok = d.W % 4 == 0 && d.tex_sc == 1 && d.tex_sx == 3 && d.disp_sx == 1 &&
            d.tex_sy % 4 == 0 && d.tex_sb % 4 == 0 && d.tex_sl % 4 == 0 &&
            d.disp_sy % 4 == 0 && d.disp_sb % 4 == 0 && d.disp_sl % 4 == 0 &&
            aligned16(a.tex) && aligned16(a.disp);
  if (d.flags & LSI_HAS_MASK)
    ok = ok && d.mask_sx == 1 && d.mask_sy % 4 == 0 && d.mask_sb % 4 == 0 &&
         d.mask_sl % 4 == 0 && aligned16(a.mask);
  return ok;
}

template <int TWL, bool WD>
const void* sweep_fn2(bool vec4, bool has_mask) {
  return vec4 ? (has_mask ? (const void*)splat_sweep_kernel<TWL, true, true, WD>
                          : (const void*)splat_sweep_kernel<TWL, true, false, WD>)
              : (has_mask ? (const void*)splat_sweep_kernel<TWL, false, true, WD>
                          : (const void*)splat_sweep_kernel<TWL, false, false, WD>);
}
template <int TWL>
const void* sweep_fn(bool vec4, bool has_mask, bool want_disp) {
  return want_disp ? sweep_fn2<TWL, true>(vec4, has_mask)
                   : sweep_fn2<TWL, false>(vec4, has_mask);
}

}  // namespace

int lsi_sweep_launch(const SplatArgs& a, const float2* range, hipStream_t stream) {
  const LsiSplatDesc* d = &a.d;
  if (d->H > 65535 || d->W > 65535) return LSI_EINVAL;  // item packing
  SweepCfg c;
  // a tile of (up to) 4096 cells: 64 KB of sums (+ two dummy rows)
  const int twl = d->Wt <= 32 ? 5 : (d->Wt <= 64 ? 6 : 7);
  const int TW = 1 << twl;
  const bool compose = (d->flags & LSI_COMPOSE) != 0;
  const int nz = compose ? 1 : d->L;
#ifndef LSI_SWEEP_CELLS
#define LSI_SWEEP_CELLS (LSI_SWEEP_T < 1024 ? 2048 : 4096)
#endif
  c.th = LSI_SWEEP_CELLS / TW;
  if (d->tune_rows > 0 && d->tune_rows < c.th) c.th = d->tune_rows;
  // shorter tiles when the tall ones would leave CUs idle
  while (c.th > 8 && (long)((d->Ht + c.th - 1) / c.th) *
                             ((d->Wt + TW - 1) / TW) * d->B * nz < 256)
    c.th >>= 1;
  if (c.th > d->Ht) c.th = (d->Ht + 7) & ~7;
  c.tiles_x = (d->Wt + TW - 1) / TW;
  c.nq4 = (d->W + 4 * SEGW - 1) / (4 * SEGW);
  c.all_layers = compose ? 1 : 0;
  {
    static const char* so_env = getenv("LSI_SWEEP_STREAM_OUT");  // experiments: 0 = scalar stores
    c.stream_out = ((so_env ? atoi(so_env) : 1) && d->Wt % 4 == 0 &&
                    aligned16(a.out_img) && aligned16(a.out_wts)) ? 1 : 0;
  }
  c.inv_nq4 = 1.0f / (float)c.nq4;
  c.inv_nlw = 1.0f / (float)(compose ? d->L : 1);
  const int tiles_y = (d->Ht + c.th - 1) / c.th;
  c.nty = tiles_y;
  // adaptive tile rows: up to 1.5x the nominal height (48 rows of 128 cells:
  // 100 KB of sums)
  const bool want_disp = (d->flags & LSI_WANT_DISP) != 0;
  if (want_disp && !a.out_disp) return LSI_ENULL;
  // (with the disparity output: 4 more bytes per cell, and the composed
  // totals live in registers, 4 cells per thread: nominal tiles only)
  c.adaptive = (tiles_y >= 2 && tiles_y <= 63 && d->Ht <= SWEEP_CAP &&
                !want_disp && !(d->reserved & 4096)) ? 1 : 0;
  c.thmax = c.adaptive ? c.th + c.th / 2 : c.th;
  const size_t lds = (size_t)(c.thmax + ROW0 + 1) * TW * (want_disp ? 20 : 16) + (size_t)SWEEP_CAP * 4 + 64 * 4 + 16 +
                     LSI_SWEEP_MAXL * 8;
  const bool vec4 = sweep_vec4(a), has_mask = (d->flags & LSI_HAS_MASK) != 0;
  const void* fn = twl == 5 ? sweep_fn<5>(vec4, has_mask, want_disp)
                            : (twl == 6 ? sweep_fn<6>(vec4, has_mask, want_disp)
                                        : sweep_fn<7>(vec4, has_mask, want_disp));
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return LSI_ELAUNCH;
  void* kargs[3] = {const_cast<SplatArgs*>(&a), &c, &range};
  if (hipLaunchKernel(fn, dim3(c.tiles_x * tiles_y, d->B, nz), dim3(SWEEP_T), kargs,
                      lds, stream) != hipSuccess)
    return LSI_ELAUNCH;
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
