// LSI_PATH_STREAM: forward splat for row-uniform projections (rectified stereo)
// without floating-point atomics on the hot path.
//
// Why this exists: on gfx950 ds_add_f32 retires ~0.33 lanes/clk/CU (measured,
// tools/microbench2.hip: 194 clk per wave instruction) while a plain LDS
// read-modify-write of a float4 costs ~20 clk per wave instruction and integer
// LDS atomics ~9 clk.  A splat needs 16 accumulations per source pixel, so the
// accumulation has to be plain RMW on memory only one wave touches.
//
// Precondition (lsi_stream_ok): for every b, M[1][0] = M[1][3] = M[2][0] =
// M[2][3] = 0 and the normaliser is positive.  Then the target ROW of a source
// pixel, its two row weights (wy0, wy1) and the normaliser depend on the source
// row only, and the corner weights factor as (wx * wy) -- up to the reference's
// 1e-3 clamp on the product (sampling.py:218-222), which is honoured exactly by
// routing the rare affected corners through an exact slow path.
//
// Structure.  Workgroup = (band of target rows, batch element b), NW <= 12 waves.
//   task  = (source row y, 256-pixel segment j, group of layers).
//   prologue  wave 0 finds the band's source rows (analytic inverse of Y(y),
//             verified with the exact fp32 Y) and fills the task table (row
//             geometry, clamp thresholds, window origin and size per task)
//             while the other waves clear the band's tile.  One barrier.
//   task loop NO barrier inside.  Waves draw tasks from a ticket counter; the
//             task order puts consecutive tickets on source rows a fifth of
//             the band apart, so the tasks in flight merge into different
//             tile rows.  Per task and layer an ITEM = 4 consecutive pixels
//             per lane (dwordx4 loads).  Two register sets take turns: each is
//             refilled right after the projection has consumed it, so two
//             items of loads are in flight per wave (the loaded HBM latency is
//             about two item periods).  The pixels are projected as packed
//             pairs and V * wx (V = (r,g,b,1) * pixel weight) is added into
//             the wave's PRIVATE window of float4 cells in LDS by plain
//             read-modify-write -- route A: the lane's 4 pixels are summed
//             into the 4 cells they can reach in registers first (4 RMWs per
//             lane instead of 8; needs floor(X) strictly increasing across
//             the wave: one DPP compare + ballot); route B: per pixel; route
//             B': per pixel, cells shared between lanes (folded disparity
//             fields), the lanes of a cell elected one at a time through a
//             byte table; route C: the general one, also for pixels outside
//             the window.  Cells live even/odd interleaved so that
//             lanes two cells apart hit consecutive 16-byte slots.
//   merge     after a task's last layer the wave adds window * (wy0, wy1) into
//             the task's two tile rows and clears the window: under the two
//             ROW locks (long bands), under per-CELL try-locks taken with one
//             integer LDS exchange per lane (short bands, where every merge
//             would queue at the same two row locks: kernel modes 3 / 4), or,
//             with LSI_DETERMINISTIC, strictly in ticket order.
//   Corners the factorisation cannot represent exactly (a side whose product
//   with only the LARGER row weight survives the clamp; cells outside the
//   window because the disparity leaves [0, max_disp]) go to a small per-wave
//   queue of (tile cell, value) and are added to the tile at the merge --
//   exact for any input, slow only when such corners are common.  Windows
//   extend beyond the image: cells left of column 0 / right of column Wt-1 are
//   dummies the merge drops, which is the reference's border rule.
//   epilogue  one barrier, then (tile + background) normalised, each output
//   written once.  With source-row bands (cfg.exchange) the first and last
//   tile rows are shared with the neighbouring bands and combined through the
//   workspace by whichever band finishes second (no spinning).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
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

constexpr int SEG = 256;   // source pixels per task (64 lanes x 4)
// Threads per workgroup at most: 768 = 12 waves of up to 168 VGPRs.  With 1024
// (16 waves, 128 VGPRs) the pixel loop spills 8-15 registers (33-53 in the
// general mode) and every workload measured slower: cfg3 93 vs 88 us, a 4-view
// shard of cfg3 37 vs 33 us, fwd_both 280 vs 224 us.
#ifndef LSI_STREAM_MAXT
#define LSI_STREAM_MAXT 768
#endif
constexpr int MAXNW = LSI_STREAM_MAXT / 64;

// Per-task table entries, filled for a whole chunk of tasks at once (one task
// per lane: by wave 0 in the prologue, by all threads for later chunks) so that
// no wave recomputes row geometry when it draws a task.
struct __attribute__((aligned(16))) TaskA {  // what the merge needs
  int row0;        // target row of the task's top contribution, band-relative
  float wy0, wy1;  // row weights incl. border masks (sampling.py:210-211);
                   // both 0: the task has nothing to do
  int win;         // window: absolute first cell | number of cells << 16
};
struct __attribute__((aligned(16))) TaskB {
  float nden, rn;  // normaliser n'(y) and its reciprocal
  int y, xs;       // source row, first source pixel of the segment
};

struct __attribute__((aligned(16))) TaskC {
  float tmin, tmax;  // clamp thresholds on a side weight (clamp_threshold)
  int l0, l1;        // layers [l0, l1) of this task
};

struct alignas(16) StreamCfg {  // (kernarg offset: see epilogue_args)
  int R;      // target rows per workgroup
  int wmax;   // window cells per wave
  int cap;    // task-table entries per chunk
  int nb;     // 64-cell units per target row
  int ngrp;   // compose mode: layer groups per (row, segment) ...
  int lpg;    // ... of this many layers each
  int qcap;   // per-wave queue entries for corners outside the window scheme
  int both;   // lsi_splat_fwd_both: a second tile sums the layers' tiles
  float inv_nb, inv_gx;  // reciprocals for division-free indexing
  float inv_per_row, inv_ngrp, inv_nseg;
  // boundary-row exchange area in the workspace (see stream_exchange_layout)
  int* xcount;     // [npass][B][nbands] arrival counters, zero between calls
  float4* xpart;   // [npass][B][nbands][2][Wt] partial rows
  long long* tstamps;  // instrumented build, flag 4 (tools/phase_probe.py)
  int exchange;  // 1: bands own source rows and exchange boundary target rows
                 // 0: bands own target rows and re-read one halo row pair
};

#define LSI_COMPILER_FENCE() asm volatile("" ::: "memory")

typedef float f2 __attribute__((ext_vector_type(2)));

// exp(a) for a packed pair; same compensated scheme as lsi::exp_accurate
__device__ __forceinline__ f2 exp_accurate2(f2 a) {
  const float L2E_HI = 1.44269502e+00f, L2E_LO = 1.92596299e-08f;
  const float LN2 = 6.93147182e-01f;
  const f2 t = a * L2E_HI;
  f2 r = {__fmaf_rn(a.x, L2E_HI, -t.x), __fmaf_rn(a.y, L2E_HI, -t.y)};
  r.x = __fmaf_rn(a.x, L2E_LO, r.x);
  r.y = __fmaf_rn(a.y, L2E_LO, r.y);
  const f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  const f2 rl = r * LN2;
  f2 o = {__fmaf_rn(e.x, rl.x, e.x), __fmaf_rn(e.y, rl.y, e.y)};
  return o;
}

// n / d for 0 <= n < 2^22, d > 0, rcp = fl(1/d): no integer-division sequence
__device__ __forceinline__ int div_small(int n, int d, float rcp) {
  int q = (int)((float)n * rcp);
  const int r = n - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// Memory order of the arrival counter of the boundary-row exchange (see the
// hand-off in the epilogue; DESIGN.md 4.1 for the measurement)
#ifndef LSI_XCHG_ORDER
#define LSI_XCHG_ORDER __ATOMIC_RELAXED
#endif

// Device-coherent accesses for the boundary-row exchange: agent-scope relaxed
// atomics are write-through / cache-bypassing (sc1), so no L2 write-back or
// invalidate (a full agent-scope fence costs tens of microseconds here).
__device__ __forceinline__ void store_coherent(float4* p, float4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  const unsigned long long lo =
      (unsigned long long)__float_as_uint(v.x) |
      ((unsigned long long)__float_as_uint(v.y) << 32);
  const unsigned long long hi =
      (unsigned long long)__float_as_uint(v.z) |
      ((unsigned long long)__float_as_uint(v.w) << 32);
  __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 load_coherent(const float4* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo =
      __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi =
      __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo),
                     __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi),
                     __uint_as_float((unsigned)(hi >> 32)));
}

// t + v*w on all four channels as two packed FMAs (v_pk_fma_f32)
__device__ __forceinline__ float4 f4_pkfma(float4 t, float4 v, float w) {
  f2 lo = {t.x, t.y}, hi = {t.z, t.w};
  const f2 vlo = {v.x, v.y}, vhi = {v.z, v.w}, ww = {w, w};
  lo = __builtin_elementwise_fma(vlo, ww, lo);
  hi = __builtin_elementwise_fma(vhi, ww, hi);
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// accumulations are not index-critical: fused multiply-add
__device__ __forceinline__ float4 f4_fma(float4 t, float4 v, float w) {
  t.x = __fmaf_rn(v.x, w, t.x); t.y = __fmaf_rn(v.y, w, t.y);
  t.z = __fmaf_rn(v.z, w, t.z); t.w = __fmaf_rn(v.w, w, t.w);
  return t;
}

// Exact slow path for one source pixel (rare): the reference's x-axis
// footprint with clipped cells (sampling.py:193-211) from floor(X) and the two
// un-masked side weights; the up-to-four corners are added with their exact
// weights clamp(wx*wy) to the extras tile by fp32 LDS atomics.  Selects, not
// multiplies, apply the masks: non-finite inputs add nothing.
__device__ __forceinline__ void slow_corners(float* extras, float4 V, float x0,
                                             float gx, float fx, float xmax,
                                             float wy0, float wy1, int row0,
                                             int rows, int Wt) {
  const float x1 = x0 + 1.0f;
  const float x0s = fminf(fmaxf(x0, 0.0f), xmax);
  const float x1s = fminf(fmaxf(x1, 0.0f), xmax);
  const float wx[2] = {(x0 == x0s) ? gx : 0.0f, (x1 == x1s) ? fx : 0.0f};
  const int cx[2] = {(int)x0s, (int)x1s};
  const float wy[2] = {wy0, wy1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = row0 + (k >> 1);
    const float c = wx[k & 1] * wy[k >> 1];
    if (!(c > 1.0e-3f) || r < 0 || r >= rows) continue;
    float* e = extras + ((size_t)r * Wt + cx[k & 1]) * 4;
    atomic_add_f32(e + 0, V.x * c);
    atomic_add_f32(e + 1, V.y * c);
    atomic_add_f32(e + 2, V.z * c);
    atomic_add_f32(e + 3, V.w * c);
  }
}

// Kernel arguments that only the epilogue needs, fetched from the kernarg
// segment where they are used (scalar loads, ~200 cycles once per pass) instead
// of living in scalar registers through the whole kernel: the kernel needs more
// uniform values than there are SGPRs, and every spilled one costs v_readlane /
// v_writelane instructions in code all waves execute.
typedef const __attribute__((address_space(4))) char* KernargPtr;
struct EpilogueArgs {
  float* out_img; float* out_wts; int* xcount; float4* xpart; float bg; int B;
  float* out_img_c; float* out_wts_c;
};
__device__ __forceinline__ EpilogueArgs epilogue_args() {
  KernargPtr ka = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));  // opaque here: the loads below are not hoisted
  typedef const __attribute__((address_space(4))) SplatArgs* AP;
  typedef const __attribute__((address_space(4))) StreamCfg* CP;
  AP ap = (AP)ka;
  CP cp = (CP)(ka + ((sizeof(SplatArgs) + alignof(StreamCfg) - 1) &
                     ~(alignof(StreamCfg) - 1)));
  EpilogueArgs e;
  e.out_img = ap->out_img; e.out_wts = ap->out_wts;
  e.out_img_c = ap->out_img_c; e.out_wts_c = ap->out_wts_c;
  e.bg = ap->d.bg_wt; e.B = ap->d.B;
  e.xcount = cp->xcount; e.xpart = cp->xpart;
  return e;
}

// Same for what only the merge and the epilogue need of the launch plan.
struct UnitArgs { int nb; float inv_nb; };
__device__ __forceinline__ UnitArgs unit_args() {
  KernargPtr ka = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  typedef const __attribute__((address_space(4))) StreamCfg* CP;
  CP cp = (CP)(ka + ((sizeof(SplatArgs) + alignof(StreamCfg) - 1) &
                     ~(alignof(StreamCfg) - 1)));
  UnitArgs u;
  u.nb = cp->nb; u.inv_nb = cp->inv_nb;
  return u;
}

// n-th float above / below a positive finite float
__device__ __forceinline__ float next_up(float t) {
  return __int_as_float(__float_as_int(t) + 1);
}
__device__ __forceinline__ float next_down(float t) {
  return __int_as_float(__float_as_int(t) - 1);
}

// Smallest side weight w with fl(w * wy) > 1e-3f (sampling.py:218-222 keeps a
// corner iff its rounded weight product exceeds 1e-3).  fp32 rounding is
// monotone, so "w >= threshold" is EXACTLY "fl(w*wy) > 1e-3f" for every w:
// one compare instead of a multiply and a compare in the hot loop.  +Inf when
// no weight <= 1 qualifies.
__device__ __forceinline__ float clamp_threshold(float wy) {
  if (!(wy > 0.0f)) return __builtin_inff();
  float t = div_rn(1.0e-3f, wy);
  if (!(t < 4.0f)) return __builtin_inff();
  for (int k = 0; k < 8; ++k) {
    const float p = next_down(t);
    if (p * wy > 1.0e-3f) t = p; else break;
  }
  for (int k = 0; k < 8; ++k) {
    if (!(t * wy > 1.0e-3f)) t = next_up(t); else break;
  }
  return t;
}

// lane l-1's value by DPP wave_shr:1 (VALU, no LDS round trip); lane 0 gets 0
__device__ __forceinline__ float lane_below(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf,
                                                 0xf, true));
}

#define LSI_RFL(x) __builtin_amdgcn_readfirstlane(x)

// streamed inputs: every byte is read once (halo rows: twice)
typedef float lsi_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
#ifdef LSI_STREAM_NT
  const lsi_f4 v = __builtin_nontemporal_load(reinterpret_cast<const lsi_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
#define LSI_LD4(p) ld4_stream(p)

// SIMPLE: the normaliser is exactly 1 and row 3 of M is (0,0,0,1) for every
// batch element (rectified stereo): u = q0 and D = d with no division.
// MODE 1 / 2: compose mode without a mask input (the training / benchmark
// configuration) with halo / exchange bands: the code and scalar registers of
// the other modes are gone.  MODE 0: everything, decided at run time.
// FULL: W is a multiple of the 256-pixel segment: every lane always has pixels.
// Try-lock / unlock of one LDS lock word (byte address): writes 1, returns what
// was there (0 = acquired).  Integer LDS exchanges are cheap; `ds_add_f32` is not.
__device__ __forceinline__ void cell_try2(unsigned a0, unsigned a1, int& o0, int& o1) {
  const int one = 1;
  asm volatile(
      "ds_wrxchg_rtn_b32 %0, %2, %4\n\tds_wrxchg_rtn_b32 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o0), "=&v"(o1)
      : "v"(a0), "v"(a1), "v"(one)
      : "memory");
}
__device__ __forceinline__ int cell_try1(unsigned a0) {
  int o;
  const int one = 1;
  asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(o) : "v"(a0), "v"(one) : "memory");
  return o;
}
__device__ __forceinline__ void cell_unlock(unsigned a0) {
  const int zero = 0;
  asm volatile("ds_write_b32 %0, %1" : : "v"(a0), "v"(zero) : "memory");
}

// Folded fields: the lanes of a wave that name the same window cell add one
// after the other.  `cnt`: the wave's table of one byte per window cell (four
// cells per word, zero between uses).  One returning integer LDS add gives a
// lane its rank among the lanes of its cell (64 lanes: a byte never carries);
// round k is rank k's turn, then the words touched are cleared.  (As in
// lsi_splat_stream2.hip; round 4.)
__device__ __forceinline__ int cell_rank(unsigned char* cnt, int cell, bool act) {
  if (!act) return -1;
  const unsigned sh = 8u * ((unsigned)cell & 3u);
  const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(cnt + (cell & ~3)), 1u << sh);
  return (int)((old >> sh) & 0xffu);
}
__device__ __forceinline__ void cell_rank_reset(unsigned char* cnt, int cell, bool act) {
  if (act) *reinterpret_cast<unsigned*>(cnt + (cell & ~3)) = 0u;
}

template <int LAYOUT, bool SIMPLE, int MODE, bool FULL, int MAXT = LSI_STREAM_MAXT>  // LAYOUT 0: channels-last, 1: planar
__global__ __launch_bounds__(MAXT) void splat_stream_kernel(SplatArgs a,
                                                           StreamCfg cfg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const LsiSplatDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = blockDim.x, NW = T >> 6;
  const int R = cfg.R, WMAX = cfg.wmax;
  const int Wt = d.Wt, Ht = d.Ht, W = d.W;
  // XCD-aware placement (speed only): workgroup i runs on XCD i % 8, each with
  // its own L2.  Neighbouring bands re-read each other's halo rows, so give
  // every XCD a contiguous run of bands (bijective remap, any grid size).
  int b, band;
  {
    const unsigned nwg = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7u, q = nwg >> 3, r8 = nwg & 7u;
    const unsigned base =
        xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const unsigned id = base + (lin >> 3);
    if (nwg < (1u << 22)) {
      b = div_small((int)id, (int)gridDim.x, cfg.inv_gx);
      band = (int)id - b * (int)gridDim.x;
    } else {
      band = id % gridDim.x;
      b = id / gridDim.x;
    }
  }
  // Two decompositions of the image into row bands (cfg.exchange):
  //  1: band = the source rows whose top target row k = floor(Y) lies in
  //     [row0, row0 + R) (band 0 also takes k = -1).  They add into target
  //     rows row0 .. row0 + R: the tile.  Its first row is shared with the
  //     band above and its last with the band below; the two partial rows of a
  //     shared target row are combined through the workspace by whichever
  //     band finishes second.  No source row is read twice.
  //  0: band = target rows [row0, row0 + R); it reads every source row that
  //     touches them (k in [row0 - 1, row0 + R - 1]) and drops the
  //     contributions that fall outside: one extra k per band, no exchange.
  const int nbands = gridDim.x;
  const int row0 = band * R;
  constexpr bool LEAN = MODE != 0;
  // MODE 3 / 4 = MODE 1 / 2 with one lock per tile CELL instead of one per tile
  // row for the merge (short bands: few tile rows, every merge would queue at
  // the same two row locks)
  constexpr bool CELL = MODE >= 3;
  const int xchg = (MODE == 1 || MODE == 3) ? 0 : ((MODE == 2 || MODE == 4) ? 1 : cfg.exchange);
  const int rows = min(xchg ? R + 1 : R, Ht - row0);  // tile rows
  const int k_lo = xchg ? (band == 0 ? -1 : row0) : row0 - 1;
  const int k_hi = xchg ? min(row0 + R, Ht) - 1 : row0 + rows - 1;
  const int top_shared = (xchg && band > 0) ? 1 : 0;
  const int bot_shared = (xchg && row0 + R < Ht) ? 1 : 0;  // tile row R exists

  // ---- LDS carve (every offset a multiple of 16) --------------------------
  float4* rb_all = reinterpret_cast<float4*>(smem_raw);      // [NW][WMAX + 16]
  unsigned char* sc_all =
      reinterpret_cast<unsigned char*>(
          rb_all + NW * (2 * (((WMAX / 2 + 15) & ~15) + 8)));  // [NW][WMAX]
  float* extras = reinterpret_cast<float*>(sc_all + NW * WMAX);  // tile [R+x][Wt][4]
  const int CAP = cfg.cap;
  // (both outputs: the composed tile follows the layer's tile)
  float4* const ctile4 =
      reinterpret_cast<float4*>(extras + (R + cfg.exchange) * Wt * 4);
  TaskA* taskA = reinterpret_cast<TaskA*>(
      extras + (R + cfg.exchange) * Wt * 4 * (cfg.both ? 2 : 1));  // [CAP]
  TaskB* taskB = reinterpret_cast<TaskB*>(taskA + CAP);           // [CAP]
  TaskC* taskC = reinterpret_cast<TaskC*>(taskB + CAP);           // [CAP]
  // [0..1] source rows, [2] task tickets, [3] range found analytically,
  // [4..5] exchange arrival order, [6] merge turn (deterministic mode),
  // [8 ..] one lock per tile row
  int* ctl = reinterpret_cast<int*>(taskC + CAP);
  int* locks = ctl + 8;
  // per-wave queue of corners the window cannot represent (see push below)
  const int Q = cfg.qcap;
  float4* const qv = reinterpret_cast<float4*>(
                         ctl + ((8 + R + 2 + 3) & ~3)) + wave * Q;   // [NW][Q]
  int* const qc = reinterpret_cast<int*>(
                      reinterpret_cast<float4*>(ctl + ((8 + R + 2 + 3) & ~3)) +
                      NW * Q) + wave * Q;                              // [NW][Q]
  // CELL: one lock word per tile cell, behind the queues
  int* const clk = reinterpret_cast<int*>(
                       reinterpret_cast<float4*>(ctl + ((8 + R + 2 + 3) & ~3)) +
                       NW * Q) + NW * Q;              // [(R + exchange) * Wt]
  const unsigned clk_addr = (unsigned)(uintptr_t)clk;
  // Window cell c lives at slot (c >> 1) + (c & 1) * WHS: lanes are two cells
  // apart, so their 16-byte cells are adjacent slots (no bank conflicts), and
  // the odd half starts 8 slots off a 256-byte boundary so that a run of
  // consecutive cells (the merge) is conflict-free too.
  const int WHS = ((WMAX / 2 + 15) & ~15) + 8;
  const int WCELLS = 2 * WHS;  // slots per window
  float4* const rb = rb_all + wave * WCELLS;  // this wave's private window
  unsigned char* const sc = sc_all + wave * WMAX;
  float4* const tile4 = reinterpret_cast<float4*>(extras);

  // Everything read from global / kernarg memory inside the loops is copied to
  // registers first: the LDS ordering fences below are compiler memory
  // barriers and would otherwise force re-loads in the hot loop.
  float m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = a.M[16 * b + k];
  const float s = d.trg_downsampling;
  const float max_disp = d.max_disp, zscale = d.zbuf_scale;
  // element strides fit in 32 bits (checked by lsi_stream_ok): half the scalar
  // registers of the 64-bit descriptor fields
  const int tex_sl = (int)d.tex_sl, tex_sb = (int)d.tex_sb,
            tex_sy = (int)d.tex_sy, tex_sc = (int)d.tex_sc;
  const int disp_sl = (int)d.disp_sl, disp_sb = (int)d.disp_sb,
            disp_sy = (int)d.disp_sy;
  const int mask_sl = (int)d.mask_sl, mask_sb = (int)d.mask_sb,
            mask_sy = (int)d.mask_sy;
  const float* __restrict__ g_tex = a.tex;
  const float* __restrict__ g_disp = a.disp;
  const float* __restrict__ g_mask = a.mask;
  // Timing stamps (tools/phase_probe.py) are compiled only into the
  // instrumented build (-DLSI_STREAM_HOOKS=1): they cost scalar registers the
  // production kernel cannot spare.
#if LSI_STREAM_HOOKS
  const int dbg = d.reserved;
#else
  constexpr int dbg = 0;
#endif
  const int nlayers = d.L;
  const float xmax = (float)Wt - 1.0f, ymax = (float)Ht - 1.0f;
  const bool has_mask = LEAN ? false : (d.flags & LSI_HAS_MASK) != 0;
  const bool compose = LEAN ? true : (d.flags & LSI_COMPOSE) != 0;
  const bool ordered = (d.flags & LSI_DETERMINISTIC) != 0;
  const float inv_md = div_rn(1.0f, max_disp);

  long long* tdbg = (dbg & 4)
                        ? cfg.tstamps + ((size_t)b * gridDim.x + band) * 160
                        : nullptr;
#if LSI_STREAM_HOOKS
  // per-wave cycle totals of the task loop's sections (tools/phase_probe.py)
  long long prof[6] = {0, 0, 0, 0, 0, 0};
  long long prof_t = 0;
#define LSI_PROF_START() prof_t = (long long)__builtin_readcyclecounter()
#define LSI_PROF(k)                                                   \
  do {                                                                \
    const long long now_ = (long long)__builtin_readcyclecounter();   \
    prof[k] += now_ - prof_t;                                         \
    prof_t = now_;                                                    \
  } while (0)
#else
#define LSI_PROF_START()
#define LSI_PROF(k)
#endif
  int tslot = 0;
#define LSI_TSTAMP()                                               \
  do {                                                             \
    if (tdbg && tid == 0 && tslot < 12)                            \
      tdbg[tslot++] = (long long)__builtin_readcyclecounter();     \
  } while (0)
  LSI_TSTAMP();

  // Row-uniform target coordinate Y(y) = (q1/n')*s - 0.5 (exact op order).
  auto row_Y = [&](int y, float& nden) {
    const float py = (float)y + 0.5f;
    const float q1 = mrow(m, 1, 0.5f, py, 0.0f);
    if (SIMPLE) {
      nden = 1.0f;
      return q1 * s - 0.5f;
    }
    nden = safe_den(mrow(m, 2, 0.5f, py, 0.0f));
    return div_rn(q1, nden) * s - 0.5f;
  };

  // ---- one-time init ------------------------------------------------------
  for (int i = tid; i < NW * WCELLS; i += T)
    rb_all[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // windows start (and are left) zero
  for (int i = tid; i < (NW * WMAX + 3) / 4; i += T)  // the waves' cell-rank tables
    reinterpret_cast<unsigned*>(sc_all)[i] = 0u;
  for (int i = tid; i < rows * Wt; i += T)
    tile4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cfg.both)
    for (int i = tid; i < rows * Wt; i += T)
      ctile4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 8 + R + 2) ctl[tid] = 0;  // tickets, turn, locks
  if (CELL)
    for (int i = tid; i < rows * Wt; i += T) clk[i] = 0;
  LSI_TSTAMP();
  const int nseg = (W + SEG - 1) / SEG;
  const int NGRP = compose ? cfg.ngrp : 1;  // layer groups per (row, segment)
  const int LPG = compose ? cfg.lpg : 1;    // layers per group
  const int per_row = nseg * NGRP;
  const float inv_per_row = compose ? cfg.inv_per_row : cfg.inv_nseg;
  const float inv_ngrp = compose ? cfg.inv_ngrp : 1.0f;
  // Source rows of the band: floor(Y(y)) in [k_lo, k_hi].  Y is a Moebius
  // function of y, monotone where the normaliser is positive, so the rows form
  // a range whose ends are found by inverting Y and then checking the result
  // with the exact fp32 Y.  Wave 0 does this and fills the first chunk of the
  // task table while the other waves clear the tile.  Maps that are
  // decreasing, too flat for fp32 to keep monotone, or not finite at the ends
  // take the scan further below instead.
  // task = (source row, 256-pixel segment, group of layers); table for tasks
  // [tg0, tg0 + CAP) of ntask, rows from ylo on, layers from lbase on: one
  // task per participating thread (threads first, first + stride, ...)
  auto pad5 = [](int n) { return (n + 4) / 5 * 5; };  // rows of the task order
  auto fill_tasks = [&](int tg0, int first, int stride, int ylo, int ntask,
                        int lbase, int lend, int nreal) {
    // Consecutive tickets go to source rows a fifth of the band apart: row
    // index yi of the (padded) task order is source row (yi % 5) * q + yi / 5,
    // q = rows / 5 rounded up.  The tasks in flight at any time then merge
    // into different tile rows (less waiting at the row locks); the at most 4
    // padding rows are empty tasks.
    const int nrows_pad = (int)(((float)ntask + 0.5f) * inv_per_row);
    const int q5 = (int)(((float)nrows_pad + 0.5f) * 0.2f);
    for (int t = first; t < CAP; t += stride) {
      const int tg = tg0 + t;
      TaskA ta; ta.row0 = -1000000; ta.wy0 = 0.f; ta.wy1 = 0.f; ta.win = 0;
      TaskB tb; tb.nden = 1.0f; tb.rn = 1.0f; tb.y = 0; tb.xs = 0;
      TaskC tc; tc.tmin = __builtin_inff(); tc.tmax = __builtin_inff();
      tc.l0 = 0; tc.l1 = 0;
      if (tg < ntask) {
        // tg / per_row etc. without integer divisions (tg < 2^20: exact)
        const int yi = (int)(((float)tg + 0.5f) * inv_per_row);
        const int rem = tg - yi * per_row;
#ifdef LSI_NO_INTERLEAVE
        const int yr = yi;
#else
        const int y5 = (int)(((float)yi + 0.5f) * 0.2f);
        const int yr = (yi - 5 * y5) * q5 + y5;
#endif
        const int sg = (int)(((float)rem + 0.5f) * inv_ngrp);
        const int grp = rem - sg * NGRP;
        const int y = ylo + yr;
        const int xs = sg * SEG;
        tc.l0 = min(lbase + grp * LPG, lend);
        tc.l1 = min(tc.l0 + LPG, lend);
        const float py = (float)y + 0.5f;
        float nden;
        const float Y = row_Y(y, nden);
        if (finite_f(Y) && fabsf(Y) < 1.0e7f && tc.l1 > tc.l0 && yr < nreal) {
          const Axis ay = splat_axis(Y, ymax);
          ta.row0 = (int)floorf(Y) - row0;
          ta.wy0 = ay.w0;
          ta.wy1 = ay.w1;
          tb.nden = nden;
          tb.rn = SIMPLE ? 1.0f : div_rn(1.0f, nden);
          tb.y = y;
          tb.xs = xs;
          // clamp thresholds against the smaller / larger non-zero row weight
          const float wymin =
              (ay.w0 == 0.f) ? ay.w1
                             : ((ay.w1 == 0.f) ? ay.w0 : fminf(ay.w0, ay.w1));
          tc.tmin = clamp_threshold(wymin);
          tc.tmax = clamp_threshold(fmaxf(ay.w0, ay.w1));
          // window hint: cells reachable for d in [0, max_disp] on the segment
          const int xe = min(xs + SEG, W);
          float lo = __builtin_inff(), hi = -__builtin_inff();
          auto x_of = [&](int xx, float dd) {
            const float q0 = mrow(m, 0, (float)xx + 0.5f, py, dd);
            return (SIMPLE ? q0 : div_rn(q0, nden)) * s - 0.5f;
          };
          if (m[0] > 0.0f) {  // X increases with x; with d by the sign of m03
            const bool neg = m[3] < 0.0f;
            lo = x_of(xs, neg ? max_disp : 0.0f);
            hi = x_of(xe - 1, neg ? 0.0f : max_disp);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float X = x_of((c & 1) ? (xe - 1) : xs,
                                   (c & 2) ? max_disp : 0.0f);
              lo = fminf(lo, X); hi = fmaxf(hi, X);
            }
          }
          if (finite_f(lo) && finite_f(hi) && fabsf(lo) < 1.0e7f &&
              fabsf(hi) < 1.0e7f) {
            // cells [wlo, wlo+wwin) -- NOT confined to the image: cells left
            // of column 0 or right of column Wt-1 are dummies that the merge
            // drops, which is exactly the reference's border rule (a corner
            // outside the image is masked, the others keep their weights:
            // sampling.py:204-211).  Origin stored with a 32768 bias.
            const int c_lo = max((int)floorf(lo) - 1, -32000);
            const int c_hi = min((int)floorf(hi) + 4, 32000);
            const int wwin = max(0, min(WMAX, c_hi - c_lo + 1));
            ta.win = (c_lo + 32768) | (wwin << 16);
          }
        }
      }
      taskA[t] = ta;
      taskB[t] = tb;
      taskC[t] = tc;
    }
  };
  const int npass = compose ? 1 : nlayers;
  const int Lp = compose ? nlayers : 1;
  if (wave == 0) {
    int y_lo = d.H, y_hi = -1;
    bool ranged = false;
    {
      const int H = d.H;
      float nA, nB;
      const float YA = row_Y(0, nA), YB = row_Y(H - 1, nB);
      const bool ok = finite_f(YA) && finite_f(YB) && fabsf(YA) < 65536.0f &&
                      fabsf(YB) < 65536.0f && nA > 0.0f && nB > 0.0f &&
                      (H == 1 || (YB - YA) >= 0.0625f * (float)(H - 1));
      if (ok) {
        // first row y in [0, H] with Y(y) >= k  (H: none).  The 64 lanes test
        // 64 consecutive rows around the estimate with the exact fp32 Y: the
        // answer is the first lane that passes, provided the lanes before it
        // (and, by monotonicity, every earlier row) fail.
        const float inv_s = __builtin_amdgcn_rcpf(s);  // estimates only
        auto lower = [&](float k, bool& good) {
          // (k + 0.5)/s = (py*m11 + m12)/(py*m21 + m22)  =>  py
          const float t = (k + 0.5f) * inv_s;
          const float py = SIMPLE ? (t - m[6]) * __builtin_amdgcn_rcpf(m[5])
                                  : (m[10] * t - m[6]) *
                                        __builtin_amdgcn_rcpf(m[5] - m[9] * t);
          const int c = finite_f(py) ? (int)fminf(fmaxf(ceilf(py - 0.5f), 0.0f),
                                                  (float)H)
                                     : 0;
          const int base = max(0, min(c - 32, H - 64));
          const int y = base + lane;
          float nd;
          // rows past the image count as passing: the result is then H
          const bool pass_ = y >= H || row_Y(y, nd) >= k;
          const unsigned long long mask = __ballot(pass_);
          const int p = mask ? __builtin_ctzll(mask) : 64;
          good = good && mask == (p < 64 ? (~0ull << p) : 0ull) &&
                 (p > 0 || base == 0) && (p < 64 || base + 64 >= H);
          return min(base + p, H);
        };
        bool good = true;
        const int lo = lower((float)k_lo, good);
        const int hi = lower((float)(k_hi + 1), good) - 1;
        if (good) { y_lo = lo; y_hi = hi; ranged = true; }
      }
    }
    if (lane == 0) { ctl[0] = y_lo; ctl[1] = y_hi; ctl[3] = ranged; }
    if (ranged)
      fill_tasks(0, lane, 64, y_lo,
                 pad5((y_hi >= y_lo) ? (y_hi - y_lo + 1) : 0) * per_row, 0, Lp,
                 y_hi - y_lo + 1);
  }
  LSI_TSTAMP();
  __syncthreads();
  if (!ctl[3]) {  // general case: every thread tests its rows
    __syncthreads();  // (everyone has read the flag)
    if (tid == 0) { ctl[0] = d.H; ctl[1] = -1; }
    __syncthreads();
    int lo = d.H, hi = -1;
    for (int y = tid; y < d.H; y += T) {
      float nd;
      const float Y = row_Y(y, nd);
      if (!finite_f(Y)) continue;
      const float y0 = floorf(Y);
      if (y0 >= (float)k_lo && y0 <= (float)k_hi) {
        lo = min(lo, y); hi = max(hi, y);
      }
    }
    if (hi >= 0) { atomicMin(&ctl[0], lo); atomicMax(&ctl[1], hi); }
    __syncthreads();
    const int ylo = ctl[0], yhi = ctl[1];
    fill_tasks(0, tid, T, ylo, pad5((yhi >= ylo) ? (yhi - ylo + 1) : 0) * per_row,
               0, Lp, yhi - ylo + 1);
    __syncthreads();
  }
  const int y_lo = ctl[0], y_hi = ctl[1];
  LSI_TSTAMP();
  const int nsrc = (y_hi >= y_lo) ? (y_hi - y_lo + 1) : 0;
  const size_t P = (size_t)Ht * Wt;
  const int ntask = pad5(nsrc) * per_row;

  struct PxData { float4 d4, t0, t1, t2, m4; };

  for (int pass = 0; pass < npass; ++pass) {
    const int l_begin = compose ? 0 : pass;
    for (int chunk0 = 0; chunk0 < ntask; chunk0 += CAP) {
      // chunk 0 of pass 0 is in the table from the prologue
      if (chunk0 > 0 || pass > 0) {
        // (the previous chunk's closing barrier protects the table)
        fill_tasks(chunk0, tid, T, y_lo, ntask, l_begin, l_begin + Lp, nsrc);
        if (tid == 0) { ctl[2] = 0; ctl[6] = 0; }
        __syncthreads();
      }
      const int nchunk = min(CAP, ntask - chunk0);
      // ================= task loop: no barrier inside ========================
      // Waves draw tasks from a ticket counter; a wave splats its task's
      // pixels along x into its private window (plain LDS read-modify-write),
      // then adds the window into the task's two tile rows under row locks.
      // The loads of the next layer -- or of the next task's first layer --
      // are in flight during the window phase.
      auto draw = [&]() {
        int sl = 0;
        if (lane == 0) sl = atomicAdd(&ctl[2], 1);
        return LSI_RFL(sl);
      };
      const float* p_disp = g_disp;
      const float* p_tex = g_tex;
      const float* p_mask = g_mask;
      int ld_left = 0;  // layers of the loader's task still to be issued
      auto aim = [&](int sl) {  // source pointers at task sl's first layer
        const TaskB tb = taskB[sl];
        const TaskC tc = taskC[sl];
        const int y = LSI_RFL(tb.y), xs = LSI_RFL(tb.xs), l0 = LSI_RFL(tc.l0);
        ld_left = LSI_RFL(tc.l1) - l0;
        // lanes past the end of the row load the row's last pixels again (the
        // compute side masks them): no predicate on the loads
        const int x = FULL ? xs + 4 * lane : min(xs + 4 * lane, W - 4);
        p_disp = g_disp + (long)l0 * disp_sl + (long)b * disp_sb +
                 (long)y * disp_sy + x;
        p_tex = g_tex + (long)l0 * tex_sl + (long)b * tex_sb +
                (long)y * tex_sy + (LAYOUT == 0 ? 3 * x : x);
        if (has_mask)
          p_mask = g_mask + (long)l0 * mask_sl + (long)b * mask_sb +
                   (long)y * mask_sy + x;
      };
      // Always exactly one load per input: every path through the loop then
      // has the same number of loads in flight, and the compiler can wait for
      // one register set while the other one's loads stay outstanding.
      auto load_layer = [&](PxData& o) {
        o.d4 = LSI_LD4(p_disp);
        if (LAYOUT == 0) {
          o.t0 = LSI_LD4(p_tex); o.t1 = LSI_LD4(p_tex + 4);
          o.t2 = LSI_LD4(p_tex + 8);
        } else {
          o.t0 = LSI_LD4(p_tex);
          o.t1 = LSI_LD4(p_tex + tex_sc);
          o.t2 = LSI_LD4(p_tex + 2 * tex_sc);
        }
        if (has_mask) o.m4 = LSI_LD4(p_mask);
        p_disp += disp_sl;
        p_tex += tex_sl;
        if (has_mask) p_mask += mask_sl;
      };
      // merge order of the deterministic mode: strictly by ticket
      auto wait_turn = [&](int sl) {
        if (lane == 0) {
          while (__hip_atomic_load(&ctl[6], __ATOMIC_ACQUIRE,
                                   __HIP_MEMORY_SCOPE_WORKGROUP) != sl)
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      };
      auto first_started_from = [&](int sl) {  // (masked rows are never started)
        while (sl < nchunk) {
          const TaskA ta = taskA[sl];
          if (LSI_RFL((ta.wy0 != 0.0f || ta.wy1 != 0.0f) ? 1 : 0)) break;
          ++sl;
        }
        return sl;
      };
      auto pass_turn = [&](int sl) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const int nx = first_started_from(sl + 1);
        if (lane == 0)
          __hip_atomic_store(&ctl[6], nx, __ATOMIC_RELEASE,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      auto lock_row = [&](int r) {
        if (lane == 0) {
          for (;;) {
            int expect = 0;
            if (__hip_atomic_compare_exchange_strong(
                    &locks[r], &expect, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_WORKGROUP))
              break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      };
      auto unlock_row = [&](int r) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
          __hip_atomic_store(&locks[r], 0, __ATOMIC_RELEASE,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
      };

      // Corners that the factorised window cannot represent exactly (a side
      // whose product with only the LARGER row weight survives the clamp;
      // pixels whose cells fall outside the window) are queued per wave as
      // (tile cell, value) and added to the tile under the row locks: at the
      // task's merge, or earlier when the queue is full.  No fp32 atomics and
      // no tile access in the pixel loop.
      int qn = 0;  // wave-uniform fill
      int q_row0 = 0;  // tile row of the current task (locks for a flush)
      bool q_use_a = false, q_use_b = false;
      // one value into one tile cell under that cell's lock (CELL; lanes may
      // name the same cell: the losers of a round try again)
      auto cell_add = [&](bool pred, int tcell, float4 v) {
        const unsigned la = clk_addr + (unsigned)tcell * 4u;
        while (__ballot(pred) != 0ull) {
          if (pred && cell_try1(la) == 0) {
            float4* e = tile4 + tcell;
            float4 t = *e;
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            *e = t;
            LSI_COMPILER_FENCE();
            cell_unlock(la);
            pred = false;
          }
        }
      };
      auto apply_queue = [&]() {  // caller holds the rows (or: CELL)
        if (CELL && !ordered) {
          for (int i0 = 0; i0 < qn; i0 += 64) {
            const int i = i0 + lane;
            const bool p = i < qn;
            cell_add(p, p ? qc[i] : 0, p ? qv[i] : make_float4(0.f, 0.f, 0.f, 0.f));
          }
          qn = 0;
          return;
        }
        for (int i = lane; i < qn; i += 64) {
          const float4 v = qv[i];
          float* e = extras + (long)qc[i] * 4;
          atomic_add_f32(e + 0, v.x);  // two records may name the same cell
          atomic_add_f32(e + 1, v.y);
          atomic_add_f32(e + 2, v.z);
          atomic_add_f32(e + 3, v.w);
        }
        qn = 0;
      };
      int cur_slot = 0;
      // queue full (rare): these lanes' corners go to the tile at once
      auto direct = [&](bool pred, int tcell, float4 val) {
        if (CELL && !ordered) {
          cell_add(pred, pred ? tcell : 0, val);
          return;
        }
        if (ordered) {
          wait_turn(cur_slot);  // (the turn is passed on after the merge)
        } else {
          if (q_use_a) lock_row(q_row0);
          if (q_use_b) lock_row(q_row0 + 1);
        }
        if (pred) {
          float* e = extras + (long)tcell * 4;
          atomic_add_f32(e + 0, val.x);
          atomic_add_f32(e + 1, val.y);
          atomic_add_f32(e + 2, val.z);
          atomic_add_f32(e + 3, val.w);
        }
        if (!ordered) {
          if (q_use_b) unlock_row(q_row0 + 1);
          if (q_use_a) unlock_row(q_row0);
        }
      };
      auto push = [&](bool pred, int tcell, float4 val) {
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) return;
        const int n = __builtin_popcountll(mask);
        if (qn + n <= Q) {
          const int idx =
              qn + (int)__builtin_amdgcn_mbcnt_hi(
                       (unsigned)(mask >> 32),
                       __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (pred) {
            qv[idx] = val;
            qc[idx] = tcell;
          }
          qn += n;
        } else {
          direct(pred, tcell, val);
        }
      };
      // exact 4-corner footprint of one pixel (sampling.py:193-222) for lanes
      // whose cells are not inside the window: clipped cells, border masks,
      // clamp on the full product; non-finite inputs add nothing
      auto push_corners = [&](bool pred, float4 V, float x0, float gx, float fx,
                              float wy0_, float wy1_, int trow0) {
        const float x1 = x0 + 1.0f;
        const float x0s = fminf(fmaxf(x0, 0.0f), xmax);
        const float x1s = fminf(fmaxf(x1, 0.0f), xmax);
        const float wx[2] = {(x0 == x0s) ? gx : 0.0f, (x1 == x1s) ? fx : 0.0f};
        const int cx[2] = {(int)x0s, (int)x1s};
        const float wy[2] = {wy0_, wy1_};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = trow0 + (k >> 1);
          const float c = wx[k & 1] * wy[k >> 1];
          const bool ok = pred && (c > 1.0e-3f) && r >= 0 && r < rows;
          push(ok, r * Wt + cx[k & 1],
               make_float4(V.x * c, V.y * c, V.z * c, V.w * c));
        }
      };

      if (ordered) {  // the first turn belongs to the first task that starts
        const int f0 = first_started_from(0);
        if (tid == 0) ctl[6] = f0;
        __syncthreads();
      }
      // ---- the wave's item stream -------------------------------------------
      // An item = one layer of a task: 4 pixels per lane.  The loader side
      // (tickets, addresses, global loads) runs two items ahead of the compute
      // side: two register sets take turns, each refilled right after the
      // projection has consumed it, so a wave keeps two items' loads in flight
      // (the loaded HBM latency is about two item periods).
      // Every register set carries the tag of the item it holds: -1 (none), or
      // the task's slot | bit 20 (first layer of its task) | bit 21 (last).
      // (Wave-uniform state is passed through readfirstlane where it steers a
      // branch: the compiler then emits scalar branches, not exec masking.)
      int ld_done = 0;   // tickets exhausted
      int ld_slot = 0;   // the loader's task
      int ld_first = 0;  // its next item is its first
      auto issue = [&](PxData& dst) -> int {
        if (LSI_RFL(ld_left) == 0 && LSI_RFL(ld_done) == 0) {
          const int sl = draw();
          if (sl >= nchunk) {
            ld_done = 1;
          } else {
            const TaskA ta = taskA[sl];
            // rows masked at the border add nothing: not even started (this
            // register set then idles for one turn)
            if (LSI_RFL((ta.wy0 != 0.0f || ta.wy1 != 0.0f) ? 1 : 0)) {
              aim(sl);
              ld_slot = sl;
              ld_first = 1 << 20;
            }
          }
        }
        int tag = -1;
        const bool have = LSI_RFL(ld_left) != 0;
        if (!have) {  // nothing to load: a harmless re-read of the first bytes
          p_disp = g_disp; p_tex = g_tex;
          if (has_mask) p_mask = g_mask;
        }
        load_layer(dst);
        if (have) {
          ld_left = LSI_RFL(ld_left) - 1;
          tag = ld_slot | ld_first | (ld_left == 0 ? (1 << 21) : 0);
          ld_first = 0;
        }
        return LSI_RFL(tag);
      };

      // compute side: state of the task in progress (wave-uniform unless noted)
      int slot = 0;
      int t_row0 = 0, t_wlo = 0, t_wwin = 0, cmax = 0, has_max = 0, rmax_row = 0;
      int win_ok = 0, fast_ok = 0, fastA_ok = 0;
      unsigned wspan = 0u, wspanA = 0u;
      float nden = 1.f, rn = 1.f, tmin = 0.f, wy0 = 0.f, wy1 = 0.f;
      float wymin = 0.f, wymax = 0.f, wlo_f = 0.f;
      bool inrange = true;  // per lane
      unsigned long long inr_mask = ~0ull, edge_mask = 1ull, win_mask = 0ull;
      f2 qb[2], q3b[2];     // per lane: layer-independent part of q = M p
      qb[0] = qb[1] = q3b[0] = q3b[1] = f2{0.f, 0.f};

      auto item = [&](PxData& cur, const int tag) -> int {
        // (a set without an item only asks the loader again: the loads of a
        // set are always issued at this one place, so the compiler never has
        // to reconcile register sets that are still in flight)
        const bool live = tag >= 0;
        if (live && (tag & (1 << 20))) {  // ---- the item starts a task ------
          slot = tag & 0xfffff;
          const TaskA ta = taskA[slot];  // LDS broadcast reads
          const TaskB tb = taskB[slot];
          const TaskC tc = taskC[slot];
          // wave-uniform by construction; tell the compiler (scalar registers)
          t_row0 = LSI_RFL(ta.row0);
          const int t_win = LSI_RFL(ta.win);
          t_wlo = (t_win & 0xffff) - 32768; t_wwin = t_win >> 16;
          const int y = LSI_RFL(tb.y);
          const int xs = LSI_RFL(tb.xs);
          nden = tb.nden;
          tmin = tc.tmin;
          const int x = xs + 4 * lane;
          inrange = FULL ? true : x < W;
          const float py = (float)y + 0.5f;
          // layer-independent part of q = M p, in the contract's rounding
          // order: ((px*m00 + py*m01) + m02), then + d*m03 per layer
          const float pym01 = py * m[1];
          const float pym31 = py * m[13];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f2 px = {(float)(x + 2 * h) + 0.5f, (float)(x + 2 * h) + 1.5f};
            qb[h] = px * m[0] + pym01;
            qb[h] = qb[h] + m[2];
            if (!SIMPLE) {
              q3b[h] = px * m[12] + pym31;
              q3b[h] = q3b[h] + m[14];
            }
          }
          rn = tb.rn;
          wy0 = ta.wy0; wy1 = ta.wy1;
          // smallest non-zero row weight: a side is exactly factorisable iff
          // its product with this one survives the 1e-3 clamp (rounding is
          // monotone)
          wymin = (wy0 == 0.f) ? wy1 : ((wy1 == 0.f) ? wy0 : fminf(wy0, wy1));
          wlo_f = (float)t_wlo;
          // Clamp handling (sampling.py:218-222): corner weight w_x*w_y
          // survives iff fl(w_x*w_y) > 1e-3.  A side whose product with the
          // SMALLER row weight survives factorises exactly and goes to the
          // window; otherwise it is taken out of the window (weight 0) and only
          // its product with the LARGER row weight can survive: that one corner
          // is queued for the tile.
          wymax = fmaxf(wy0, wy1);
          const int rmax = t_row0 + ((wy1 > wy0) ? 1 : 0);
          has_max =
              LSI_RFL(((wymax != wymin) && rmax >= 0 && rmax < rows) ? 1 : 0);
          cmax = rmax * Wt + t_wlo;  // tile cell of window cell 0, row rmax
          rmax_row = rmax;
          // last admissible left-cell offset; a window of fewer than two cells
          // (the whole segment maps outside the image) admits no lane at all
          wspan = (unsigned)max(t_wwin - 2, 0);
          win_ok = t_wwin >= 2 ? 1 : 0;
          // the fast routes assume at most one clamped side per pixel
          fast_ok = LSI_RFL((win_ok && tmin <= 0.5f) ? 1 : 0);
          // route A touches cells cl0 .. cl0 + 3
          wspanA = (unsigned)max(t_wwin - 4, 0);
          fastA_ok = LSI_RFL((t_wwin >= 4 && tmin <= 0.5f) ? 1 : 0);
          win_mask = win_ok ? ~0ull : 0ull;
          // lanes exempt from the "strictly increasing" test: lane 0, and the
          // tail lanes beyond the image (no in-window lane follows them)
          inr_mask = FULL ? ~0ull : __ballot(inrange);
          edge_mask = ~inr_mask | 1ull;
          cur_slot = slot;
          q_row0 = t_row0;
          q_use_a = wy0 != 0.f && t_row0 >= 0 && t_row0 < rows;
          q_use_b = wy1 != 0.f && t_row0 + 1 >= 0 && t_row0 + 1 < rows;
          LSI_PROF(5);  // task setup
        }
#if LSI_STREAM_HOOKS
        if (LEAN) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        LSI_PROF(0);  // waiting for this item's loads
#endif
        int next_tag = -1;
        float x0v[4], w0v[4], w1v[4];
        float4 Vv[4];  // route A: the lane's 4 cell sums; else V of its 4 pixels
        int cl0 = 0, routeA = 0;
        if (live) {
          // ---- projection of the 4 pixels as two packed pairs (v_pk_*_f32) --
          {
            const float dv[4] = {cur.d4.x, cur.d4.y, cur.d4.z, cur.d4.w};
            float pwv[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f2 dvp = {dv[2 * h], dv[2 * h + 1]};
              // q0 = ((px*m00 + py*m01) + m02) + d*m03, each op rounded
              const f2 q0 = qb[h] + dvp * m[3];
              f2 q3, u;
              if (SIMPLE) {
                q3 = dvp;
                u = q0;  // index-critical u = q0 / n' with n' == 1 exactly
              } else {
                q3 = q3b[h] + dvp * m[15];
                u.x = div_rn(q0.x, nden);  // index-critical: IEEE division
                u.y = div_rn(q0.y, nden);
              }
              const f2 X = u * s - 0.5f;
              // sampling.py:193-211 on the x axis: floor, x1 - x, x - x0
              const f2 x0 = {floorf(X.x), floorf(X.y)};
              const f2 gx = (x0 + 1.0f) - X;
              const f2 fx = X - x0;
              // weights are not index-critical: reciprocal multiplies
              const f2 dd = SIMPLE ? q3 : q3 * rn;
              const f2 xn = dd * inv_md;
              // helpers.py:180-193: exp((clip(x,0,1) - 0.5)*scale) * [x > 0]
              f2 c = {__builtin_amdgcn_fmed3f(xn.x, 0.0f, 1.0f),
                      __builtin_amdgcn_fmed3f(xn.y, 0.0f, 1.0f)};
              c = (c - 0.5f) * zscale;
              f2 e = exp_accurate2(c);
              e.x = xn.x > 0.0f ? e.x : 0.0f;  // NaN disparity -> weight 0
              e.y = xn.y > 0.0f ? e.y : 0.0f;
              if (has_mask) {
                const f2 mkp = {h == 0 ? cur.m4.x : cur.m4.z,
                                h == 0 ? cur.m4.y : cur.m4.w};
                e = e * mkp;
              }
              x0v[2 * h] = x0.x; x0v[2 * h + 1] = x0.y;
              w0v[2 * h] = gx.x; w0v[2 * h + 1] = gx.y;
              w1v[2 * h] = fx.x; w1v[2 * h + 1] = fx.y;
              pwv[2 * h] = e.x; pwv[2 * h + 1] = e.y;
            }
            // pixel i's colour channels (register selection at compile time)
            auto tex_of = [&](int i, float& r, float& g, float& b_) {
              if (LAYOUT == 0) {
                const float t[12] = {cur.t0.x, cur.t0.y, cur.t0.z, cur.t0.w,
                                     cur.t1.x, cur.t1.y, cur.t1.z, cur.t1.w,
                                     cur.t2.x, cur.t2.y, cur.t2.z, cur.t2.w};
                r = t[3 * i]; g = t[3 * i + 1]; b_ = t[3 * i + 2];
              } else {
                const float t0[4] = {cur.t0.x, cur.t0.y, cur.t0.z, cur.t0.w};
                const float t1[4] = {cur.t1.x, cur.t1.y, cur.t1.z, cur.t1.w};
                const float t2[4] = {cur.t2.x, cur.t2.y, cur.t2.z, cur.t2.w};
                r = t0[i]; g = t1[i]; b_ = t2[i];
              }
            };
            // ---- route A: a lane's 4 pixels land in cells cl0 .. cl0+3 ------
            // Their 8 side contributions are summed per cell in registers and
            // the window gets 4 read-modify-writes per lane instead of 8 (the
            // kernel is bound by LDS cycles).  Lanes' cells of one RMW are
            // distinct when floor(X) of the lanes' FIRST pixels increases
            // strictly across the wave.
            cl0 = (int)(x0v[0] - wlo_f);
            int dl[4];
            dl[0] = 0;
            unsigned long long regA =
                __ballot((unsigned)cl0 <= wspanA) &
                (__ballot(x0v[0] > lane_below(x0v[0])) | edge_mask);
#pragma unroll
            for (int i = 1; i < 4; ++i) {
              dl[i] = (int)(x0v[i] - x0v[0]);
              regA &= __ballot((unsigned)dl[i] <= 2u);
            }
            routeA = LSI_RFL((((~regA & inr_mask) == 0ull) && fastA_ok) ? 1 : 0);
            if (routeA) {
#pragma unroll
              for (int k = 0; k < 4; ++k) Vv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
              for (int i = 0; i < ((dbg & 512) ? 0 : 4); ++i) {  // 512: counters only
                float w0 = w0v[i], w1 = w1v[i];
                float r, g, b_;
                tex_of(i, r, g, b_);
                const float4 V =
                    make_float4(r * pwv[i], g * pwv[i], b_ * pwv[i], pwv[i]);
                // clamp (at most the smaller side: tmin <= 0.5 on this route)
                const unsigned long long clamped =
                    __ballot(!(fminf(w0, w1) >= tmin)) & inr_mask;
                if (clamped != 0ull) {
                  const bool c0 = !(w0 >= tmin), c1 = !(w1 >= tmin);
                  if (has_max) {
                    const float kq = (c0 ? w0 : w1) * wymax;
                    const int cellq = t_wlo + cl0 + dl[i] + (c0 ? 0 : 1);
                    push(inrange && (c0 || c1) && kq > 1.0e-3f &&
                             (unsigned)cellq < (unsigned)Wt,
                         rmax_row * Wt + cellq,
                         make_float4(V.x * kq, V.y * kq, V.z * kq, V.w * kq));
                  }
                  if (c0) w0 = 0.0f;
                  if (c1) w1 = 0.0f;
                }
                if (i == 0) {
                  Vv[0] = make_float4(V.x * w0, V.y * w0, V.z * w0, V.w * w0);
                  Vv[1] = make_float4(V.x * w1, V.y * w1, V.z * w1, V.w * w1);
                } else {
                  const bool e0 = dl[i] == 0, e1 = dl[i] == 1, e2 = dl[i] == 2;
                  const float a0 = e0 ? w0 : 0.0f;
                  const float a1 = e1 ? w0 : (e0 ? w1 : 0.0f);
                  const float a2 = e2 ? w0 : (e1 ? w1 : 0.0f);
                  const float a3 = e2 ? w1 : 0.0f;
                  Vv[0] = f4_pkfma(Vv[0], V, a0);
                  Vv[1] = f4_pkfma(Vv[1], V, a1);
                  Vv[2] = f4_pkfma(Vv[2], V, a2);
                  Vv[3] = f4_pkfma(Vv[3], V, a3);
                }
              }
            } else {
              // V = (r, g, b, 1) * pixel weight, for all 4 pixels
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float r, g, b_;
                tex_of(i, r, g, b_);
                Vv[i] = make_float4(r * pwv[i], g * pwv[i], b_ * pwv[i], pwv[i]);
              }
            }
            // (after this the layer's input registers are dead and can take
            // the next loads)
          }
          // The derived values are pinned here so that the projection is not
          // sunk below the loads: the loads then overwrite dead registers and
          // need no copies.
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            asm volatile("" : "+v"(Vv[i].x), "+v"(Vv[i].y), "+v"(Vv[i].z),
                              "+v"(Vv[i].w), "+v"(x0v[i]), "+v"(w0v[i]),
                              "+v"(w1v[i]));
          }
          LSI_PROF(1);  // projection
        }
        next_tag = issue(cur);  // the set takes the item two ahead
        LSI_PROF(2);  // ticket, addresses, load issue
        if (live) {
          if (!(dbg & 128)) {  // 128: no window phase (timing only)
          // ---- window phase ---------------------------------------------------
          // One test per layer decides between the branch-free fast route (all
          // 4 pixels of every lane land inside the window, and floor(X) is
          // strictly increasing across the wave, so the lanes' left cells are
          // distinct) and the exact general route.
          if (routeA) {
            // cells cl0, cl0+2 share a half of the window, cl0+1, cl0+3 the other
            const int par = cl0 & 1, hlf = cl0 >> 1;
            float4* ce = rb + hlf + par * WHS;              // cell cl0
            float4* co = rb + hlf + par + (1 - par) * WHS;  // cell cl0 + 1
            if (FULL || inrange) {
              float4 t;
              t = ce[0]; t.x += Vv[0].x; t.y += Vv[0].y; t.z += Vv[0].z; t.w += Vv[0].w; ce[0] = t;
              LSI_COMPILER_FENCE();
              t = co[0]; t.x += Vv[1].x; t.y += Vv[1].y; t.z += Vv[1].z; t.w += Vv[1].w; co[0] = t;
              LSI_COMPILER_FENCE();
              t = ce[1]; t.x += Vv[2].x; t.y += Vv[2].y; t.z += Vv[2].z; t.w += Vv[2].w; ce[1] = t;
              LSI_COMPILER_FENCE();
              t = co[1]; t.x += Vv[3].x; t.y += Vv[3].y; t.z += Vv[3].z; t.w += Vv[3].w; co[1] = t;
            }
            LSI_COMPILER_FENCE();
          } else {
          int clv[4];
          unsigned long long regular = ~0ull, inwin = ~0ull;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // left-cell offset in the window (+-Inf saturates; NaN gives
            // offset 0 but both its side weights are then clamped to 0)
            clv[i] = (int)(x0v[i] - wlo_f);
            inwin &= __ballot((unsigned)clv[i] <= wspan);
            regular &= __ballot(x0v[i] > lane_below(x0v[i])) | edge_mask;
          }
          regular &= inwin;
          // route B': every pixel inside the window but floor(X) not
          // increasing across the wave (folded disparity fields): the per-pixel
          // code of route B with the lanes of a cell elected one at a time
          const bool fold = ((~inwin & inr_mask) == 0ull) && fast_ok &&
                            ((~regular & inr_mask) != 0ull);
          if ((((~regular & inr_mask) == 0ull) && fast_ok) || fold) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float w0 = w0v[i], w1 = w1v[i];
              const float4 V = Vv[i];
              // w0 + w1 = 1: at most the smaller side can be clamped
              const unsigned long long clamped =
                  __ballot(!(fminf(w0, w1) >= tmin)) & inr_mask;
              if (clamped != 0ull) {
                // (tmin <= 0.5 on this route: only the smaller side can be)
                const bool c0 = !(w0 >= tmin), c1 = !(w1 >= tmin);
                if (has_max) {
                  const float k = (c0 ? w0 : w1) * wymax;
                  const int cellq = t_wlo + clv[i] + (c0 ? 0 : 1);
                  push(inrange && (c0 || c1) && k > 1.0e-3f &&
                           (unsigned)cellq < (unsigned)Wt,
                       rmax_row * Wt + cellq,
                       make_float4(V.x * k, V.y * k, V.z * k, V.w * k));
                }
                if (c0) w0 = 0.0f;
                if (c1) w1 = 0.0f;
              }
              float4* cell = rb + (clv[i] >> 1) + (clv[i] & 1) * WHS;
              float4* cell1 = rb + ((clv[i] + 1) >> 1) + ((clv[i] + 1) & 1) * WHS;
              if (!fold) {
                if (FULL || inrange) {
                  *cell = f4_fma(*cell, V, w0);
                  LSI_COMPILER_FENCE();
                  *cell1 = f4_fma(*cell1, V, w1);
                }
                LSI_COMPILER_FENCE();
              } else {
                // the lanes of a cell take turns by rank (cell_rank below)
                const bool act = FULL || inrange;
                const int rank = cell_rank(sc, clv[i], act);
                for (int k = 0; __ballot(rank >= k) != 0ull; ++k) {
                  if (rank == k) {
                    *cell = f4_fma(*cell, V, w0);
                    LSI_COMPILER_FENCE();
                    *cell1 = f4_fma(*cell1, V, w1);
                  }
                  LSI_COMPILER_FENCE();
                }
                cell_rank_reset(sc, clv[i], act);
              }
            }
          } else {
            // ---- general route, pixel by pixel ------------------------------
            // (one copy of the code: the loop is not unrolled, the pixel's
            // values are selected -- this route is rare)
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
              auto pick = [&](float a0, float a1, float a2, float a3) {
                return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
              };
              const float x0 = pick(x0v[0], x0v[1], x0v[2], x0v[3]);
              float w0 = pick(w0v[0], w0v[1], w0v[2], w0v[3]);
              float w1 = pick(w1v[0], w1v[1], w1v[2], w1v[3]);
              const float4 V = make_float4(
                  pick(Vv[0].x, Vv[1].x, Vv[2].x, Vv[3].x),
                  pick(Vv[0].y, Vv[1].y, Vv[2].y, Vv[3].y),
                  pick(Vv[0].z, Vv[1].z, Vv[2].z, Vv[3].z),
                  pick(Vv[0].w, Vv[1].w, Vv[2].w, Vv[3].w));
              const int cl = (int)(x0 - wlo_f);
              const bool in_b = (unsigned)cl <= wspan;
              const bool inw = in_b && inrange && win_ok;
              const unsigned long long inw_mask =
                  __ballot(in_b) & inr_mask & win_mask;
              const unsigned long long mono_ok =
                  __ballot(x0 > lane_below(x0)) | edge_mask;
              // clamped sides: !(p > 1e-3) is also true for NaN weights
              const bool c0 = !(w0 * wymin > 1.0e-3f);
              const bool c1 = !(w1 * wymin > 1.0e-3f);
              if (has_max) {
                const float k0 = w0 * wymax, k1 = w1 * wymax;
                push(inw && c0 && k0 > 1.0e-3f &&
                         (unsigned)(t_wlo + cl) < (unsigned)Wt, cmax + cl,
                     make_float4(V.x * k0, V.y * k0, V.z * k0, V.w * k0));
                push(inw && c1 && k1 > 1.0e-3f &&
                         (unsigned)(t_wlo + cl + 1) < (unsigned)Wt, cmax + cl + 1,
                     make_float4(V.x * k1, V.y * k1, V.z * k1, V.w * k1));
              }
              // lanes outside the window: exact 4-corner path (cells outside
              // the image and non-finite X fail its range tests, add nothing)
              push_corners(inrange && !inw && V.w != 0.0f, V, x0, w0, w1, wy0,
                           wy1, t_row0);
              if (c0) w0 = 0.0f;
              if (c1) w1 = 0.0f;
              // (dereferenced by in-window lanes only)
              float4* cell = rb + (cl >> 1) + (cl & 1) * WHS;
              float4* cell1 = rb + ((cl + 1) >> 1) + ((cl + 1) & 1) * WHS;
              if (mono_ok == ~0ull) {
                if (inw) {
                  *cell = f4_fma(*cell, V, w0);
                  LSI_COMPILER_FENCE();
                  *cell1 = f4_fma(*cell1, V, w1);
                }
                LSI_COMPILER_FENCE();
              } else if (inw_mask != 0ull) {
                // some lanes may share a left cell: they take turns by rank;
                // the lanes of one round have distinct cells
                const int rank = cell_rank(sc, cl, inw);
                for (int k = 0; __ballot(rank >= k) != 0ull; ++k) {
                  if (rank == k) {
                    *cell = f4_fma(*cell, V, w0);
                    LSI_COMPILER_FENCE();
                    *cell1 = f4_fma(*cell1, V, w1);
                  }
                  LSI_COMPILER_FENCE();
                }
                cell_rank_reset(sc, cl, inw);
              }
            }
          }
          }
          }
        }
        LSI_PROF(3);  // window phase
        if (live && (tag & (1 << 21))) {
          // ---- last layer done: window -> the task's two tile rows ----------
          const bool use_a = q_use_a, use_b = q_use_b;
          if (ordered) {
            wait_turn(slot);
          } else if (!CELL) {  // ascending order: no deadlock
            if (use_a) lock_row(t_row0);
            if (use_b) lock_row(t_row0 + 1);
          }
          if (qn != 0) apply_queue();
          float4* trow = tile4 + (long)t_row0 * Wt + t_wlo;
          if (CELL && !ordered) {
            // every lane locks its own pair of cells (one exchange each, both
            // in flight), adds, releases; a lane that lost one to another
            // wave's merge gives nothing up it still needs and tries again
            const unsigned lrow = clk_addr + (unsigned)(t_row0 * Wt + t_wlo) * 4u;
            for (int c = lane; c < t_wwin; c += 64) {
              float4* wc = rb + (c >> 1) + (c & 1) * WHS;
              const float4 v = *wc;
              *wc = make_float4(0.f, 0.f, 0.f, 0.f);  // ready for the next task
              const bool inside = (unsigned)(t_wlo + c) < (unsigned)Wt;
              bool na = use_a && inside, nb = use_b && inside;
              const unsigned la = lrow + (unsigned)c * 4u, lb = la + (unsigned)Wt * 4u;
              while (__ballot(na || nb) != 0ull) {
                if (na && nb) {
                  int oa, ob;
                  cell_try2(la, lb, oa, ob);
                  if (oa == 0) { trow[c] = f4_fma(trow[c], v, wy0); }
                  if (ob == 0) { trow[Wt + c] = f4_fma(trow[Wt + c], v, wy1); }
                  LSI_COMPILER_FENCE();
                  if (oa == 0) { cell_unlock(la); na = false; }
                  if (ob == 0) { cell_unlock(lb); nb = false; }
                } else if (na) {
                  if (cell_try1(la) == 0) {
                    trow[c] = f4_fma(trow[c], v, wy0);
                    LSI_COMPILER_FENCE();
                    cell_unlock(la);
                    na = false;
                  }
                } else if (nb) {
                  if (cell_try1(lb) == 0) {
                    trow[Wt + c] = f4_fma(trow[Wt + c], v, wy1);
                    LSI_COMPILER_FENCE();
                    cell_unlock(lb);
                    nb = false;
                  }
                }
              }
            }
          } else {
#ifndef LSI_MERGE_BATCHED
          for (int c = lane; c < t_wwin; c += 64) {
            float4* wc = rb + (c >> 1) + (c & 1) * WHS;
            const float4 v = *wc;
            *wc = make_float4(0.f, 0.f, 0.f, 0.f);  // ready for the next task
            const bool inside = (unsigned)(t_wlo + c) < (unsigned)Wt;
            if (use_a && inside) trow[c] = f4_fma(trow[c], v, wy0);
            if (use_b && inside) trow[Wt + c] = f4_fma(trow[Wt + c], v, wy1);
          }
#else
          // two cells per lane and round: all reads first, then all writes
          // (one LDS round trip for six accesses)
          for (int c0 = 0; c0 < t_wwin; c0 += 128) {
            const int ca = c0 + lane, cb = ca + 64;
            const bool va = ca < t_wwin, vb = cb < t_wwin;
            float4* wa = rb + (ca >> 1) + (ca & 1) * WHS;
            float4* wb = rb + (cb >> 1) + (cb & 1) * WHS;
            const bool ia = va && (unsigned)(t_wlo + ca) < (unsigned)Wt;
            const bool ib = vb && (unsigned)(t_wlo + cb) < (unsigned)Wt;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 xa = z4, xb = z4, a0 = z4, a1 = z4, b0 = z4, b1 = z4;
            if (va) xa = *wa;
            if (vb) xb = *wb;
            if (use_a && ia) a0 = trow[ca];
            if (use_b && ia) a1 = trow[Wt + ca];
            if (use_a && ib) b0 = trow[cb];
            if (use_b && ib) b1 = trow[Wt + cb];
            LSI_COMPILER_FENCE();
            if (va) *wa = z4;  // ready for the next task
            if (vb) *wb = z4;
            if (use_a && ia) trow[ca] = f4_fma(a0, xa, wy0);
            if (use_b && ia) trow[Wt + ca] = f4_fma(a1, xa, wy1);
            if (use_a && ib) trow[cb] = f4_fma(b0, xb, wy0);
            if (use_b && ib) trow[Wt + cb] = f4_fma(b1, xb, wy1);
          }
#endif
          }
          if (ordered) {
            pass_turn(slot);
          } else if (!CELL) {
            if (use_b) unlock_row(t_row0 + 1);
            if (use_a) unlock_row(t_row0);
          }
        }
        LSI_PROF(4);  // merge
        return next_tag;
      };

      PxData setA, setB;
      setA.d4 = setA.t0 = setA.t1 = setA.t2 = make_float4(0.f, 0.f, 0.f, 0.f);
      setA.m4 = make_float4(1.f, 1.f, 1.f, 1.f);
      setB = setA;
      int tagA = issue(setA);
      int tagB = issue(setB);
      LSI_PROF_START();
      // (a set without an item -- a masked row was drawn, or the tickets ran
      // out -- tries again; the stream ends when both sets are empty and no
      // ticket is left)
      while (LSI_RFL((tagA >= 0 || tagB >= 0 || ld_done == 0) ? 1 : 0)) {
        tagA = item(setA, tagA);
        tagB = item(setB, tagB);
      }
      if (tdbg && lane == 0 && chunk0 == 0 && pass == 0) {
        tdbg[12 + wave] = (long long)__builtin_readcyclecounter();
#if LSI_STREAM_HOOKS
        for (int k = 0; k < 6; ++k) tdbg[32 + wave * 8 + k] = prof[k];
#endif
      }
      LSI_TSTAMP();
      __syncthreads();  // every window is merged: the tile is complete
      LSI_TSTAMP();
    }

    // ================= epilogue for this pass ===============================
    const EpilogueArgs ea = epilogue_args();
    const UnitArgs ua = unit_args();
    const int NB = ua.nb, nunits = rows * NB;
    const float lbg = compose ? (float)nlayers * ea.bg : ea.bg;
    const int lo_ = compose ? 0 : pass;
    auto finish = [&](int r, int cell, float4 A) {  // normalise and store
      const float A0 = A.x + lbg, A1 = A.y + lbg, A2 = A.z + lbg;
      const float Wsum = A.w + lbg;
      const float wd = safe_den(Wsum);
      const size_t o =
          ((size_t)lo_ * ea.B + b) * P + (size_t)(row0 + r) * Wt + cell;
      ea.out_img[3 * o + 0] = div_rn(A0, wd);
      ea.out_img[3 * o + 1] = div_rn(A1, wd);
      ea.out_img[3 * o + 2] = div_rn(A2, wd);
      ea.out_wts[o] = Wsum;
    };
    // boundary j (between bands j-1 and j): counter and two partial rows
    const size_t xb = ((size_t)pass * ea.B + b) * nbands;
    auto xrow = [&](int j, int side) {
      return ea.xpart + ((xb + j) * 2 + side) * (size_t)Wt;
    };
    for (int unit = wave; unit < nunits; unit += NW) {
      const int r = div_small(unit, NB, ua.inv_nb);
      const int cell = (unit - r * NB) * 64 + lane;
      if (cell >= Wt) continue;
      float4* tcell = tile4 + r * Wt + cell;
      const float4 A = *tcell;
      if (r == 0 && top_shared) {
        store_coherent(xrow(band, 1) + cell, A);      // lower band's share
      } else if (r == R && bot_shared) {
        store_coherent(xrow(band + 1, 0) + cell, A);  // upper band's share
      } else {
        finish(r, cell, A);
        if (cfg.both) {  // (halo bands only: every tile row is final here)
          float4 c4 = ctile4[r * Wt + cell];
          c4.x += A.x; c4.y += A.y; c4.z += A.z; c4.w += A.w;
          if (pass + 1 < npass) {
            ctile4[r * Wt + cell] = c4;
          } else {  // ldi.py:167-174: sum of the layers' canvases, normalised
            const float lbg_c = (float)nlayers * ea.bg;
            const float Wsum = c4.w + lbg_c;
            const float wd = safe_den(Wsum);
            const size_t o = (size_t)b * P + (size_t)(row0 + r) * Wt + cell;
            ea.out_img_c[3 * o + 0] = div_rn(c4.x + lbg_c, wd);
            ea.out_img_c[3 * o + 1] = div_rn(c4.y + lbg_c, wd);
            ea.out_img_c[3 * o + 2] = div_rn(c4.z + lbg_c, wd);
            ea.out_wts_c[o] = Wsum;
          }
        }
        if (pass + 1 < npass) *tcell = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    LSI_TSTAMP();
    if (top_shared || bot_shared) {
      // the partial rows are performed (write-through stores, waited for by
      // every thread) before one thread announces this band on each boundary
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) {
        ctl[4] = top_shared
                     ? __hip_atomic_fetch_add(&ea.xcount[xb + band], 1,
                                              LSI_XCHG_ORDER,
                                              __HIP_MEMORY_SCOPE_AGENT)
                     : 0;
        ctl[5] = bot_shared
                     ? __hip_atomic_fetch_add(&ea.xcount[xb + band + 1], 1,
                                              LSI_XCHG_ORDER,
                                              __HIP_MEMORY_SCOPE_AGENT)
                     : 0;
      }
      __syncthreads();
      const bool fin_top = top_shared && ctl[4] == 1;
      const bool fin_bot = bot_shared && ctl[5] == 1;
      if (fin_top || fin_bot) {
        for (int unit = wave; unit < nunits; unit += NW) {
          const int r = div_small(unit, NB, ua.inv_nb);
          const int cell = (unit - r * NB) * 64 + lane;
          if (cell >= Wt) continue;
          const bool top = (r == 0 && fin_top), bot = (r == R && fin_bot);
          if (!top && !bot) continue;
          const float4 mine_v = *(tile4 + r * Wt + cell);
          const float4* other = top ? xrow(band, 0) : xrow(band + 1, 1);
          const float4 o4 = load_coherent(other + cell);
          finish(r, cell, make_float4(mine_v.x + o4.x, mine_v.y + o4.y,
                                      mine_v.z + o4.z, mine_v.w + o4.w));
        }
        if (tid == 0) {  // leave the counters zero for the next call
          if (fin_top)
            __hip_atomic_store(&ea.xcount[xb + band], 0, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          if (fin_bot)
            __hip_atomic_store(&ea.xcount[xb + band + 1], 0, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (pass + 1 < npass) {  // shared tile rows start the next pass empty
        __syncthreads();
        for (int i = tid; i < Wt; i += T) {
          if (top_shared) tile4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bot_shared) tile4[R * Wt + i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    LSI_TSTAMP();
    if (pass + 1 < npass) __syncthreads();
  }
}

// task-table entries per chunk
int stream_cap(int ntask) {
  int cap = (ntask + 15) / 16 * 16;
  if (cap < 16) cap = 16;
  if (cap > 512) cap = 512;
  return cap;
}

size_t stream_lds_bytes(const LsiSplatDesc* d, int tile_rows, int nw, int wmax,
                        int cap, int qcap, int cell = 0) {  // (tile_rows: both tiles counted)
  return (cell ? (size_t)tile_rows * d->Wt * 4 : 0) + (size_t)nw * (2 * (((wmax / 2 + 15) & ~15) + 8)) * 16 +
         (size_t)nw * wmax +
         (size_t)tile_rows * d->Wt * 16 +
         (size_t)cap * (sizeof(TaskA) + sizeof(TaskB) + sizeof(TaskC)) +
         (size_t)((8 + tile_rows + 2 + 3) & ~3) * 4 + (size_t)nw * qcap * 20 +
         16;
}

// layout class of the texture strides: 0 channels-last, 1 planar, 2 RGBD
// pixels (LSI_PACKED_RGBD: colour and disparity interleaved), -1 none of them
int tex_layout(const LsiSplatDesc* d) {
  const bool al = (d->tex_sl % 4 == 0) && (d->tex_sb % 4 == 0) &&
                  (d->tex_sy % 4 == 0);
  if (!al) return -1;
  if ((d->flags & LSI_PACKED_RGBD) && d->tex_sc == 1 && d->tex_sx == 4 &&
      d->disp_sx == 4 && d->tex_sl == d->disp_sl && d->tex_sb == d->disp_sb &&
      d->tex_sy == d->disp_sy)
    return 2;
  if (d->tex_sc == 1 && d->tex_sx == 3) return 0;
  if (d->tex_sx == 1 && d->tex_sc % 4 == 0) return 1;
  return -1;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int lsi_stream_ok(const LsiSplatDesc* d, const float* M) {
  if (!d || !M) return 0;
  if (!lsi_rowband_ok(d, M)) return 0;
  // the disparity output: composed, by the compact instance only (checked with
  // the layout below)
  const bool want_disp = (d->flags & LSI_WANT_DISP) != 0;
  if (want_disp && !(d->flags & LSI_COMPOSE)) return 0;
  if (d->W % 4 != 0) return 0;
  if (d->Wt > 32767) return 0;  // window origin is kept in 16 bits
  {  // the kernel keeps element strides in 32 bits
    const int64_t st[] = {d->tex_sl, d->tex_sb, d->tex_sy, d->tex_sc, d->disp_sl,
                          d->disp_sb, d->disp_sy, d->mask_sl, d->mask_sb,
                          d->mask_sy};
    for (int64_t v : st)
      if (v < 0 || v > 0x7fffffffLL) return 0;
  }
  const int layout = tex_layout(d);
  if (layout < 0) return 0;
  if (layout != 2 &&
      (d->disp_sx != 1 || d->disp_sy % 4 || d->disp_sb % 4 || d->disp_sl % 4))
    return 0;
  // RGBD pixels: only the compact instance reads them (no mask, unit
  // normaliser: checked below) -- else the any-stride TILE path
  if ((layout == 2 || want_disp) &&
      (layout == 1 || d->L > 15 ||
       (d->flags & (LSI_HAS_MASK | LSI_DETERMINISTIC)) ||
       (layout == 0 && (d->tex_sx != 3 || d->tex_sc != 1))))
    return 0;
  if ((d->flags & LSI_HAS_MASK) &&
      (d->mask_sx != 1 || d->mask_sy % 4 || d->mask_sb % 4 || d->mask_sl % 4))
    return 0;
  const float s = d->trg_downsampling;
  float need = 0.0f;
  bool simple = true, rows_ok = true;
  for (int b = 0; b < d->B; ++b) {
    const float* m = M + 16 * b;
    if (m[4] != 0.0f || m[8] != 0.0f) return 0;  // M[1][0], M[2][0]
    if (!(m[5] >= 1.0f)) rows_ok = false;
    if (!(m[9] == 0.0f && m[10] == 1.0f && m[12] == 0.0f && m[13] == 0.0f &&
          m[14] == 0.0f && m[15] == 1.0f))
      simple = false;
    // normaliser over the rows (independent of x here)
    const float n0 = m[9] * 0.5f + m[10];
    const float n1 = m[9] * ((float)d->H - 0.5f) + m[10];
    const float nmin = fminf(n0, n1);
    const float span = (fabsf(m[0]) * (float)SEG + fabsf(m[3]) * d->max_disp) /
                       nmin * s;
    if (!(span == span)) return 0;
    need = fmaxf(need, span);
  }
  if ((layout == 2 || want_disp) && !simple) return 0;
  int win = (int)ceilf(need) + 8;  // the window's margin: 1 cell left, 4 right
  win = (win + 15) / 16 * 16;
  if (win < 64) win = 64;
  if (win > 512) win = 512;  // beyond this the excess takes the exact slow path
  return win | (simple ? LSI_STREAM_SIMPLE_BIT : 0) |
         ((simple && rows_ok) ? LSI_STREAM_ROWS_BIT : 0);
}

// Workspace layout of the boundary-row exchange for bands of R rows: arrival
// counters first, then the partial rows.
struct XLayout { size_t count_bytes, part_bytes; int nbands, npass; };
static XLayout stream_exchange_layout(const LsiSplatDesc* d, int R) {
  XLayout x;
  x.npass = (d->flags & LSI_COMPOSE) ? 1 : d->L;
  x.nbands = (d->Ht + R - 1) / R;
  x.count_bytes =
      (((size_t)x.npass * d->B * x.nbands * sizeof(int)) + 255) / 256 * 256;
  x.part_bytes = (size_t)x.npass * d->B * x.nbands * 2 * d->Wt * 16;
  return x;
}

size_t lsi_stream_workspace_bytes(const LsiSplatDesc* d) {
  const XLayout x = stream_exchange_layout(d, 1);  // worst case: 1-row bands
  return x.count_bytes + x.part_bytes;
}

// Band height, waves per workgroup and layer groups per task.  Units: shader
// cycles.  The cost model: a workgroup spends a fixed prologue (clear tile and
// windows, row range, task table), then its waves stream through the tasks
// with no barrier -- bounded either by the waves' own latency chains or by the
// CU's instruction issue shared with the co-resident workgroups -- then the
// last waves finish alone (tail), then the epilogue.
struct StreamPlan { int R, nw, xch, ngrp, lpg, cap, qcap, cell; size_t lds; double est; };

static int stream_plan(const LsiSplatDesc* d, int wmax, bool both,
                       StreamPlan* out) {
  const int nseg = (d->W + SEG - 1) / SEG;
  const int grp_override = (d->reserved >> 12) & 0xf;  // experiments only
  static const char* cap_env = getenv("LSI_STREAM_LDS_CAP");  // experiments
  const size_t lds_cap = cap_env ? (size_t)atol(cap_env) : 160 * 1024;
  static const char* tl_env = getenv("LSI_STREAM_TLAT");
  static const char* ti_env = getenv("LSI_STREAM_TISSUE");
  // per (task, layer) item = 4 pixels per lane: a wave alone needs ~t_lat for
  // it; a CU retires one item per ~t_issue when enough waves share its SIMDs
  const double t_lat = tl_env ? atof(tl_env) : 3000.0;
  const double t_issue = ti_env ? atof(ti_env) : 400.0;
  const bool compose = (d->flags & LSI_COMPOSE) != 0;
  const int layers = compose ? d->L : 1;
  const int npass = compose ? 1 : d->L;
  const bool lean = compose && !(d->flags & LSI_HAS_MASK);
  const int force_mode = (d->reserved >> 16) & 3;  // experiments: 1 halo, 2 exchange
  StreamPlan best; best.est = -1.0; best.nw = 0;
  for (int xch = 0; xch <= (both ? 0 : 1); ++xch) {
    if (force_mode && !both && xch != force_mode - 1) continue;
    for (int R = 1; R <= 64; R *= 2) {
      if (d->tune_rows > 0 && R != d->tune_rows) continue;
      if (d->tune_rows <= 0 && R > 1 && R / 2 >= d->Ht) break;
      const long nwg = (long)((d->Ht + R - 1) / R) * d->B;
      // source rows per band ~ R / s (one more target row's worth when the
      // band re-reads its halo)
      const int srows =
          (int)ceilf((float)(R + 1 - xch) / d->trg_downsampling);
      for (int ngrp = 1; ngrp <= layers; ++ngrp) {
        if (grp_override && ngrp != (grp_override < layers ? grp_override : layers))
          continue;
        const int lpg = (layers + ngrp - 1) / ngrp;
        if ((layers + lpg - 1) / lpg != ngrp) continue;  // same split, fewer groups
        // (the kernel pads the rows of its task order to a multiple of 5)
        const int ntask = (srows + 4) / 5 * 5 * nseg * ngrp;
        const int cap_override = (d->reserved >> 20) & 0xff;  // experiments
        const int cap = cap_override ? cap_override * 16 : stream_cap(ntask);
        // merge exclusion: one lock per tile row, or (lean variants, short
        // bands) one per tile cell -- experiments: reserved bits 18-19 force
        // 1 = row locks, 2 = cell locks
        const int force_cell = (d->reserved >> 18) & 3;
        for (int cell = 0; cell <= 1; ++cell) {
        // (measured: cell locks win while a band has few tasks -- cfg2 25.9 ->
        // 19.8 us, a 4-view shard of cfg3 29.6 -> 27.0 us -- and lose once the
        // row interleave alone keeps the row locks free: cfg5 105 -> 121 us)
        const bool cell_ok = lean && !both && R + xch <= 8 && ntask <= 40;
        if (cell != (cell_ok ? 1 : 0) && !(force_cell == 1 && cell == 0)) continue;
        if (force_cell == 1 && cell == 1) continue;
        StreamPlan local; local.nw = 0; local.est = -1.0;
        double local_pref = 0.0;
        for (int c = MAXNW; c >= 4; --c) {
          if (d->tune_threads > 0 && c != (d->tune_threads + 63) / 64) continue;
          int q = 64;
          const int trows = (R + xch) * (both ? 2 : 1);
          while (q >= 16 &&
                 stream_lds_bytes(d, trows, c, wmax, cap, q, cell) > lds_cap)
            q /= 2;
          if (q < 16) continue;
          const size_t lds = stream_lds_bytes(d, trows, c, wmax, cap, q, cell);
          long k = (long)(160 * 1024 / lds);   // co-resident workgroups per CU
          if (k > MAXNW / c) k = MAXNW / c;    // (168 VGPRs: 12 waves per CU)
          if (k < 1) k = 1;
          const long kk = (nwg + 255) / 256 < k ? (nwg + 255) / 256 : k;
          const long rounds = (nwg + 256 * kk - 1) / (256 * kk);
          const double cells = (double)(R + xch) * d->Wt + (double)c * wmax;
          const double fixed = 5000.0 + cells / (c * 64.0) * 10.0 +
                               (double)cap / c * 3.0;
          // a wave's time per item: its own latency chain, or its share of
          // the CU's issue capacity when every wave is busy
          const double per_item = fmax(t_lat, (double)(c * kk) * t_issue);
          // waves take tasks in rounds: the last round is rarely full
          const double task_rounds = ceil((double)ntask / c);
          // a merge holds its two row locks for ~900 cycles: with few tile
          // rows the merges of a band queue up behind each other
          const double merge_serial =
              cell ? 0.0 : (double)ntask * 900.0 / fmax(1.0, (R + xch) / 2.0);
          const double loop =
              fmax(task_rounds * (lpg * per_item + 1600.0), merge_serial);
          const double epi = (double)(R + xch) * d->Wt / (c * 64.0) * 70.0 +
                             2500.0 + 16000.0 * xch;
          const double est = (double)rounds * (fixed + npass * (loop + epi));
          // Among the wave counts of ONE configuration, prefer waves spread
          // evenly over the four SIMDs (and, with them, the larger per-wave
          // queue): measured ~3 % at cfg3 (12 waves x 64 entries 95.7 us, 15 x
          // 16 98.8 us; long bands only: short ones are bound by their fixed
          // phases).  Configurations compete on the plain estimate.
          const double pref = est * ((c % 4 != 0 && !cell) ? 1.05 : 1.0);
          if (local.nw == 0 || pref < local_pref) {
            local_pref = pref;
            local.est = est; local.R = R; local.nw = c; local.xch = xch;
            local.ngrp = ngrp; local.lpg = lpg; local.cap = cap; local.qcap = q;
            local.lds = lds; local.cell = cell;
          }
        }
        // (ties go to the later candidate: longer bands, less halo)
        if (local.nw != 0 && (best.est < 0.0 || local.est <= best.est)) best = local;
        }
      }
    }
  }
  if (best.nw == 0) return LSI_EINVAL;
  *out = best;
  static const bool verbose = getenv("LSI_STREAM_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "lsi stream plan: R=%d waves=%d %s layer-groups=%d x %d "
            "table=%d queue=%d %s-locks est=%.0f cycles lds=%zu\n", best.R, best.nw,
            best.xch ? "exchange" : "halo", best.ngrp, best.lpg, best.cap,
            best.qcap, best.cell ? "cell" : "row", best.est, best.lds);
  return LSI_OK;
}

int lsi_stream_launch(const SplatArgs& a, hipStream_t stream) {
  const LsiSplatDesc* d = &a.d;
  const int layout = tex_layout(d);
  if (layout < 0 || d->W % 4 != 0) return LSI_EINVAL;
  if (!aligned16(a.tex) || (layout != 2 && !aligned16(a.disp)) ||
      ((d->flags & LSI_HAS_MASK) && !aligned16(a.mask)))
    return LSI_EINVAL;
  if ((d->tune_window & ~LSI_STREAM_FLAG_BITS) <= 0)
    return LSI_EINVAL;  // from lsi_stream_ok
  if (d->flags & LSI_WANT_DISP) {
    // composed view by the compact instance, then its per-layer-tile variant as
    // the disparity pass (ldi.py:147-180: each layer's splatted disparity over
    // its own weight, maximum over the layers)
    if (!a.out_disp) return LSI_ENULL;
    SplatArgs a1 = a;
    a1.d.flags &= ~LSI_WANT_DISP;
    a1.out_disp = nullptr;
    // (lsi_stream_ok admits only what the compact instance takes; with the
    // instance switched off -- LSI_STREAM2=0, test bits -- the any-pose path
    // renders the request)
    if (!lsi_stream2_applies(a1, (d->tune_window & LSI_STREAM_SIMPLE_BIT) != 0, layout))
      return lsi_tile_launch(a, stream);
    const int wmax = d->tune_window & ~LSI_STREAM_FLAG_BITS;
    const int rc = lsi_stream2_launch(a1, wmax, stream);
    if (rc != LSI_OK) return rc;
    return lsi_stream2_launch(a, wmax, stream, true);
  }
  if (lsi_stream2_applies(a, (d->tune_window & LSI_STREAM_SIMPLE_BIT) != 0, layout))
    return lsi_stream2_launch(a, d->tune_window & ~LSI_STREAM_FLAG_BITS, stream);
  // (RGBD pixels outside the compact instance's cases, e.g. per-layer outputs
  // alone: the any-stride path)
  if (layout == 2) return lsi_tile_launch(a, stream);
  const int NB = (d->Wt + 63) / 64;
  StreamCfg cfg;
  cfg.wmax = d->tune_window & ~LSI_STREAM_FLAG_BITS;
  StreamPlan plan;
  const bool both = a.out_img_c != nullptr;
  if (both && (d->flags & LSI_COMPOSE)) return LSI_EINVAL;
  if (stream_plan(d, cfg.wmax, both, &plan) != LSI_OK) return LSI_EINVAL;
  const int R = plan.R, nw = plan.nw;
  const int threads = nw * 64;
  const size_t lds = plan.lds;
  if (lds > 160 * 1024) return LSI_EINVAL;
  cfg.R = R;
  cfg.exchange = plan.xch;
  cfg.cap = plan.cap;
  cfg.ngrp = plan.ngrp;
  cfg.lpg = plan.lpg;
  cfg.qcap = plan.qcap;
  cfg.both = both ? 1 : 0;
  cfg.nb = NB;
  cfg.inv_nb = 1.0f / (float)NB;
  cfg.inv_gx = 1.0f / (float)((d->Ht + R - 1) / R);
  {
    const int nseg = (d->W + SEG - 1) / SEG;
    cfg.inv_per_row = 1.0f / (float)(nseg * plan.ngrp);
    cfg.inv_ngrp = 1.0f / (float)plan.ngrp;
    cfg.inv_nseg = 1.0f / (float)nseg;
  }
  // boundary-row exchange area
  const XLayout x = stream_exchange_layout(d, R);
  if (!a.canvas && (cfg.exchange || (d->reserved & 4))) return LSI_ENULL;
  // the counters keep the place they have with 1-row bands, whatever R is: a
  // kept workspace (LSI_WS_KEEP) then never sees partial rows where a later
  // call with the same dimensions looks for zeroed counters
  const size_t part_off = stream_exchange_layout(d, 1).count_bytes;
  cfg.xcount = nullptr;
  cfg.xpart = nullptr;
  if (cfg.exchange) {
    if (a.ws_bytes < part_off + x.part_bytes) return LSI_EWORKSPACE;
    if (!aligned16(a.canvas)) return LSI_EINVAL;
    cfg.xcount = reinterpret_cast<int*>(a.canvas);
    cfg.xpart = reinterpret_cast<float4*>(reinterpret_cast<char*>(a.canvas) +
                                          part_off);
  }
  cfg.tstamps = nullptr;
  if (d->reserved & 4) {  // phase probe: stamps after the regular workspace
    const size_t off = (lsi_splat_workspace_bytes(d) + 255) / 256 * 256;
    if (a.ws_bytes < off + (size_t)x.nbands * d->B * 160 * 8) return LSI_EWORKSPACE;
    cfg.tstamps = reinterpret_cast<long long*>(
        reinterpret_cast<char*>(a.canvas) + off);
  }
  if (cfg.exchange && !(d->flags & LSI_WS_KEEP)) {
    if (hipMemsetAsync(a.canvas, 0, x.count_bytes, stream) != hipSuccess)
      return LSI_ELAUNCH;
  }
  dim3 grid((d->Ht + R - 1) / R, d->B);
  const bool simple = (d->tune_window & LSI_STREAM_SIMPLE_BIT) != 0;
  const bool lean = (d->flags & LSI_COMPOSE) && !(d->flags & LSI_HAS_MASK);
  const int mode = lean ? (cfg.exchange ? 2 : 1) + (plan.cell ? 2 : 0) : 0;
  const void* fn;
#define LSI_PICK3(L_, S_, M_, F_) ((const void*)splat_stream_kernel<L_, S_, M_, F_>)
#define LSI_PICK2(L_, S_, M_) \
  (full ? LSI_PICK3(L_, S_, M_, true) : LSI_PICK3(L_, S_, M_, false))
#define LSI_PICK(L_, S_) \
  (mode == 1 ? LSI_PICK2(L_, S_, 1) : mode == 2 ? LSI_PICK2(L_, S_, 2) :   \
   mode == 3 ? LSI_PICK2(L_, S_, 3) : mode == 4 ? LSI_PICK2(L_, S_, 4) :   \
   LSI_PICK2(L_, S_, 0))
  const bool full = d->W % SEG == 0;
  if (layout == 0)
    fn = simple ? LSI_PICK(0, true) : LSI_PICK(0, false);
  else
    fn = simple ? LSI_PICK(1, true) : LSI_PICK(1, false);
#undef LSI_PICK3
#undef LSI_PICK2
#undef LSI_PICK
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return LSI_ELAUNCH;
  void* kargs[2] = {const_cast<SplatArgs*>(&a), &cfg};
  if (hipLaunchKernel(fn, grid, dim3(threads), kargs, lds, stream) != hipSuccess)
    return LSI_ELAUNCH;
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
