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
// Structure.  Workgroup = (band of target rows, batch element b), NW <= 16 waves.
//   task  = (source row y, 256-pixel segment j), all layers of the pass.
//     prologue wave 0 finds the band's source rows (analytic inverse of Y(y),
//             verified with the exact fp32 Y) and fills the task table while
//             the other waves clear the band's tile.
//     x-pass  waves draw tasks from a ticket counter.  Lanes load 4 consecutive
//             pixels (dwordx4), project them, and add V*wx0 / V*wx1
//             (V = (r,g,b,1)*pixel weight) into the task's PRIVATE window of
//             float4 cells in LDS by plain RMW.  Lanes are 4 pixels apart, so
//             their cells are distinct whenever floor(X) is strictly
//             increasing across the wave (checked); otherwise the lanes are
//             ranked per cell with an integer LDS atomic and the RMW is issued
//             rank by rank.
//     barrier
//     merge   every 64-cell unit of the band's LDS tile is owned by one wave,
//             which adds window[cell] * wy of each task that touches its row
//             (deterministic summation order).
//     barrier (only if another step follows)
//   Corners that the factorisation cannot represent exactly (clamped products,
//   cells outside the window because the disparity leaves [0, max_disp]) are
//   added to the tile with fp32 LDS atomics -- exact for any input, slow only
//   when such corners are common.
//   epilogue: (tile + background) normalised, each output written once.  With
//   source-row bands (cfg.exchange) the first and last tile rows are shared
//   with the neighbouring bands and combined through the workspace.
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
constexpr int MAXNW = 16;
// lsi_stream_ok's return value: window cells, plus this bit when every batch
// element has normaliser == 1 and M row 3 == (0,0,0,1) (division-free kernel)
constexpr int LSI_STREAM_SIMPLE_BIT = 1 << 20;

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

struct alignas(16) StreamCfg {  // (kernarg offset: see epilogue_args)
  int R;      // target rows per workgroup
  int wmax;   // window cells per task
  int tpw;    // windows per wave: nw * tpw tasks are in flight per step
  int cap;    // task-table entries, a multiple of nw * tpw
  int nb;     // 64-cell units per target row
  int steps_per_chunk;  // cap / (nw * tpw)
  float inv_nb, inv_nwin, inv_gx;  // reciprocals for division-free indexing
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

// SIMPLE: the normaliser is exactly 1 and row 3 of M is (0,0,0,1) for every
// batch element (rectified stereo): u = q0 and D = d with no division.
// MODE 1 / 2: compose mode without a mask input (the training / benchmark
// configuration) with halo / exchange bands: the code and scalar registers of
// the other modes are gone.  MODE 0: everything, decided at run time.
// FULL: W is a multiple of the 256-pixel segment: every lane always has pixels.
template <int LAYOUT, bool SIMPLE, int MODE, bool FULL>  // LAYOUT 0: channels-last, 1: planar
__global__ __launch_bounds__(1024) void splat_stream_kernel(SplatArgs a,
                                                           StreamCfg cfg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const LsiSplatDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = blockDim.x, NW = T >> 6;
  const int R = cfg.R, WMAX = cfg.wmax, TPW = cfg.tpw;
  const int NWIN = NW * TPW;  // windows (= tasks) per step, <= 64
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
  //     contributions that fall outside: one extra k per band, no exchange
  //     (cheaper when a band is only a few microseconds of work).
  const int nbands = gridDim.x;
  const int row0 = band * R;
  constexpr bool LEAN = MODE != 0;
  const int xchg = MODE == 1 ? 0 : (MODE == 2 ? 1 : cfg.exchange);
  const int rows = min(xchg ? R + 1 : R, Ht - row0);  // tile rows
  const int k_lo = xchg ? (band == 0 ? -1 : row0) : row0 - 1;
  const int k_hi = xchg ? min(row0 + R, Ht) - 1 : row0 + rows - 1;
  const int top_shared = (xchg && band > 0) ? 1 : 0;
  const int bot_shared = (xchg && row0 + R < Ht) ? 1 : 0;  // tile row R exists

  float4* rb_all = reinterpret_cast<float4*>(smem_raw);  // [NWIN][WMAX]
  unsigned* cnt_all = reinterpret_cast<unsigned*>(rb_all + NWIN * WMAX);
  float* extras = reinterpret_cast<float*>(cnt_all + NW * WMAX);  // [R+1][Wt][4]
  const int CAP = cfg.cap;
  TaskA* taskA = reinterpret_cast<TaskA*>(
      extras + (R + cfg.exchange) * Wt * 4);  // [CAP]
  TaskB* taskB = reinterpret_cast<TaskB*>(taskA + CAP);           // [CAP]
  int* yrange = reinterpret_cast<int*>(taskB + CAP);  // [0..1] rows, [2] slot ticket, [4..5] exchange order
  unsigned* cnt = cnt_all + wave * WMAX;

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
  // Timing-experiment hooks (tools/phase_probe.py, bench.py --debug-flags) are
  // compiled only into the instrumented build (-DLSI_STREAM_HOOKS=1): they cost
  // a dozen scalar registers the production kernel cannot spare.
#if LSI_STREAM_HOOKS
  const int dbg = d.reserved;
#else
  constexpr int dbg = 0;
#endif
  const int nlayers = d.L;
  const float xmax = (float)Wt - 1.0f, ymax = (float)Ht - 1.0f;
  const bool has_mask = LEAN ? false : (d.flags & LSI_HAS_MASK) != 0;
  const bool compose = LEAN ? true : (d.flags & LSI_COMPOSE) != 0;
  const float inv_md = div_rn(1.0f, max_disp);

  long long* tdbg = (dbg & 4)
                        ? cfg.tstamps + ((size_t)b * gridDim.x + band) * 32
                        : nullptr;
  int tslot = 0;
#define LSI_TSTAMP()                                               \
  do {                                                             \
    if (tdbg && tid == 0 && tslot < 32)                            \
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
  // (task windows are zeroed by their owning wave when the task starts)
  for (int i = tid; i < NW * WMAX; i += T) cnt_all[i] = 0u;
  for (int i = tid; i < rows * Wt; i += T)
    reinterpret_cast<float4*>(extras)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid == 0) yrange[2] = 0;  // slot tickets (first read after a barrier)
  LSI_TSTAMP();
  const int nseg = (W + SEG - 1) / SEG;
  const float inv_nseg = 1.0f / (float)nseg;
  // Source rows of the band: floor(Y(y)) in [k_lo, k_hi].  Y is a Moebius
  // function of y, monotone where the normaliser is positive, so the rows form
  // a range whose ends are found by inverting Y and then checking the result
  // with the exact fp32 Y.  Wave 0 does this and fills the first chunk of the
  // task table while the other waves clear the tile: the uniform arithmetic is
  // not repeated by (and does not contend with) the other waves.  Maps that
  // are decreasing, too flat for fp32 to keep monotone, or not finite at the
  // ends take the scan further below instead.
  // task table for tasks [tg0, tg0 + CAP) of ntask, rows from ylo on: one task
  // per participating thread (threads first, first + stride, ...)
  auto fill_tasks = [&](int tg0, int first, int stride, int ylo, int ntask) {
    for (int t = first; t < CAP; t += stride) {
        const int tg = tg0 + t;
        TaskA ta; ta.row0 = -1000000; ta.wy0 = 0.f; ta.wy1 = 0.f; ta.win = 0;
        TaskB tb; tb.nden = 1.0f; tb.rn = 1.0f; tb.y = 0; tb.xs = 0;
        if (tg < ntask) {
          // tg / nseg without an integer division (tg < 2^20: exact in fp32)
          const int yi = (int)(((float)tg + 0.5f) * inv_nseg);
          const int y = ylo + yi;
          const int xs = (tg - yi * nseg) * SEG;
          const float py = (float)y + 0.5f;
          float nden;
          const float Y = row_Y(y, nden);
          if (finite_f(Y) && fabsf(Y) < 1.0e7f) {
            const Axis ay = splat_axis(Y, ymax);
            ta.row0 = (int)floorf(Y) - row0;
            ta.wy0 = ay.w0;
            ta.wy1 = ay.w1;
            tb.nden = nden;
            tb.rn = SIMPLE ? 1.0f : div_rn(1.0f, nden);
            tb.y = y;
            tb.xs = xs;
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
              // cells [wlo, wlo+wwin) confined to the image: an in-window lane
              // then needs no border mask (both its cells are valid)
              const int c_lo = max((int)floorf(lo) - 1, 0);
              const int c_hi = min((int)floorf(hi) + 2, Wt - 1);
              const int wwin = max(0, min(WMAX, c_hi - c_lo + 1));
              ta.win = c_lo | (wwin << 16);
            }
          }
        }
        taskA[t] = ta;
        taskB[t] = tb;
      }
    };
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
    if (lane == 0) { yrange[0] = y_lo; yrange[1] = y_hi; yrange[3] = ranged; }
    LSI_TSTAMP();
    if (ranged)
      fill_tasks(0, lane, 64, y_lo,
                 ((y_hi >= y_lo) ? (y_hi - y_lo + 1) : 0) * nseg);
  }
  LSI_TSTAMP();
  __syncthreads();
  LSI_TSTAMP();
  if (!yrange[3]) {  // general case: every thread tests its rows
    __syncthreads();  // (everyone has read the flag)
    if (tid == 0) { yrange[0] = d.H; yrange[1] = -1; }
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
    if (hi >= 0) { atomicMin(&yrange[0], lo); atomicMax(&yrange[1], hi); }
    __syncthreads();
    const int ylo = yrange[0], yhi = yrange[1];
    fill_tasks(0, tid, T, ylo, ((yhi >= ylo) ? (yhi - ylo + 1) : 0) * nseg);
    __syncthreads();
  }
  const int y_lo = yrange[0], y_hi = yrange[1];
  LSI_TSTAMP();
  const int nsrc = (y_hi >= y_lo) ? (y_hi - y_lo + 1) : 0;
  const size_t P = (size_t)Ht * Wt;

  const int npass = compose ? 1 : nlayers;
  const int Lp = compose ? nlayers : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int l_begin = compose ? 0 : pass;
    const int ntask = nsrc * nseg;
    const int nstep = div_small(ntask + NWIN - 1, NWIN, cfg.inv_nwin);

    const int steps_per_chunk = cfg.steps_per_chunk;

    for (int step = 0, sidx = 0; step < nstep; ++step, ++sidx) {
      if (sidx == steps_per_chunk) sidx = 0;
      // chunk 0 is in the table from the prologue, and still is at the start
      // of a later pass unless this pass needed more than one chunk
      if (sidx == 0 && (step > 0 || (pass > 0 && nstep > steps_per_chunk))) {
        // (the previous step's closing barrier protects the table)
        fill_tasks(step * NWIN, tid, T, y_lo, ntask);
        __syncthreads();
      }
      LSI_TSTAMP();
      // ================= x-pass ==============================================
      // task = (source row y, 256-pixel segment j), all layers of the pass.
      // Waves draw task slots from a ticket counter (tasks differ in cost);
      // the wave that draws slot t owns LDS window [t] until the merge.
      for (;;) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&yrange[2], 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= NWIN) break;
        const TaskA ta = taskA[sidx * NWIN + slot];  // LDS broadcast reads
        const TaskB tb = taskB[sidx * NWIN + slot];
        // wave-uniform by construction; tell the compiler (scalar registers)
        const int t_valid = __builtin_amdgcn_readfirstlane(
            (ta.wy0 != 0.0f || ta.wy1 != 0.0f) ? 1 : 0);
        if (!t_valid || (dbg & 64)) continue;  // dbg 64: overhead-only timing
        const int t_row0 = __builtin_amdgcn_readfirstlane(ta.row0);
        const int t_win = __builtin_amdgcn_readfirstlane(ta.win);
        const int t_wlo = t_win & 0xffff, t_wwin = t_win >> 16;
        const int y = __builtin_amdgcn_readfirstlane(tb.y);
        const int xs = __builtin_amdgcn_readfirstlane(tb.xs);
        const float nden = tb.nden;

        float4* rb = rb_all + slot * WMAX;
        const int x = xs + 4 * lane;
        const bool inrange = FULL ? true : x < W;
        const float py = (float)y + 0.5f;
        // row-uniform pieces of q = M p, in the contract's rounding order
        const float pym01 = py * m[1];
        const float pym31 = py * m[13];
        const float rn = tb.rn;
        const float wy0 = ta.wy0, wy1 = ta.wy1;
        // smallest non-zero row weight: a side is exactly factorisable iff its
        // product with this one survives the 1e-3 clamp (rounding is monotone)
        const float wymin =
            (wy0 == 0.f) ? wy1 : ((wy1 == 0.f) ? wy0 : fminf(wy0, wy1));
        const float wlo_f = (float)t_wlo;

        struct PxData { float4 d4, t0, t1, t2, m4; };
        // per-lane source pointers, advanced by one layer stride per load
        const float* p_disp = g_disp + (long)l_begin * disp_sl +
                              (long)b * disp_sb + (long)y * disp_sy + x;
        const float* p_tex = g_tex + (long)l_begin * tex_sl + (long)b * tex_sb +
                             (long)y * tex_sy + (LAYOUT == 0 ? 3 * x : x);
        const float* p_mask = has_mask ? g_mask + (long)l_begin * mask_sl +
                                             (long)b * mask_sb +
                                             (long)y * mask_sy + x
                                       : nullptr;
        auto load_layer = [&](PxData& o) {
          if (inrange) {
            o.d4 = *reinterpret_cast<const float4*>(p_disp);
            if (LAYOUT == 0) {
              const float4* t4 = reinterpret_cast<const float4*>(p_tex);
              o.t0 = t4[0]; o.t1 = t4[1]; o.t2 = t4[2];
            } else {
              o.t0 = *reinterpret_cast<const float4*>(p_tex);
              o.t1 = *reinterpret_cast<const float4*>(p_tex + tex_sc);
              o.t2 = *reinterpret_cast<const float4*>(p_tex + 2 * tex_sc);
            }
            if (has_mask) o.m4 = *reinterpret_cast<const float4*>(p_mask);
          }
          p_disp += disp_sl;
          p_tex += tex_sl;
          if (has_mask) p_mask += mask_sl;
        };

        // Clamp handling (sampling.py:218-222): corner weight w_x*w_y survives
        // iff fl(w_x*w_y) > 1e-3.  A side whose product with the SMALLER row
        // weight survives factorises exactly (rounding is monotone) and goes
        // to the window; otherwise it is taken out of the window (weight 0)
        // and only its product with the LARGER row weight can survive: that
        // one corner is added to the extras tile.
        const float wymax = fmaxf(wy0, wy1);
        const int rmax = t_row0 + ((wy1 > wy0) ? 1 : 0);
        const int has_max = __builtin_amdgcn_readfirstlane(
            ((wymax != wymin) && rmax >= 0 && rmax < rows && !(dbg & 1)) ? 1
                                                                         : 0);
        float* const emax = extras + ((long)rmax * Wt + t_wlo) * 4;
        // last admissible left-cell offset; a window of fewer than two cells
        // (the whole segment maps outside the image) admits no lane at all
        const unsigned wspan = (unsigned)max(t_wwin - 2, 0);
        const int win_ok = t_wwin >= 2 ? 1 : 0;
        const unsigned long long win_mask = win_ok ? ~0ull : 0ull;
        // lanes exempt from the "strictly increasing" test: lane 0, and the
        // tail lanes beyond the image (no in-window lane follows them)
        const unsigned long long inr_mask = FULL ? ~0ull : __ballot(inrange);
        const unsigned long long edge_mask = ~inr_mask | 1ull;
        const int l_end = l_begin + Lp;

        PxData cur;
        cur.d4 = cur.t0 = cur.t1 = cur.t2 = make_float4(0.f, 0.f, 0.f, 0.f);
        cur.m4 = make_float4(1.f, 1.f, 1.f, 1.f);
        load_layer(cur);
        // zero this task's window while the first loads are in flight (same
        // wave, in-order LDS: no barrier needed before its own RMWs)
        for (int c = lane; c < t_wwin; c += 64)
          rb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        LSI_COMPILER_FENCE();

        for (int l = l_begin; l < l_end; ++l) {
          // ---- projection of the 4 pixels as two packed pairs (v_pk_*_f32) --
          float x0v[4], w0v[4], w1v[4];
          float4 Vv[4];
          {
            const float dv[4] = {cur.d4.x, cur.d4.y, cur.d4.z, cur.d4.w};
            float pwv[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f2 px = {(float)(x + 2 * h) + 0.5f,
                             (float)(x + 2 * h) + 1.5f};
              const f2 dvp = {dv[2 * h], dv[2 * h + 1]};
              // q0 = ((px*m00 + py*m01) + m02) + d*m03, each op rounded
              f2 q0 = px * m[0] + pym01;
              q0 = q0 + m[2];
              q0 = q0 + dvp * m[3];
              f2 q3, u;
              if (SIMPLE) {
                q3 = dvp;
                u = q0;  // index-critical u = q0 / n' with n' == 1 exactly
              } else {
                q3 = px * m[12] + pym31;
                q3 = q3 + m[14];
                q3 = q3 + dvp * m[15];
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
            // V = (r, g, b, 1) * pixel weight, for all 4 pixels: after this the
            // layer's input registers are dead and can take the next loads
            if (LAYOUT == 0) {
              Vv[0] = make_float4(cur.t0.x * pwv[0], cur.t0.y * pwv[0],
                                  cur.t0.z * pwv[0], pwv[0]);
              Vv[1] = make_float4(cur.t0.w * pwv[1], cur.t1.x * pwv[1],
                                  cur.t1.y * pwv[1], pwv[1]);
              Vv[2] = make_float4(cur.t1.z * pwv[2], cur.t1.w * pwv[2],
                                  cur.t2.x * pwv[2], pwv[2]);
              Vv[3] = make_float4(cur.t2.y * pwv[3], cur.t2.z * pwv[3],
                                  cur.t2.w * pwv[3], pwv[3]);
            } else {
              Vv[0] = make_float4(cur.t0.x * pwv[0], cur.t1.x * pwv[0],
                                  cur.t2.x * pwv[0], pwv[0]);
              Vv[1] = make_float4(cur.t0.y * pwv[1], cur.t1.y * pwv[1],
                                  cur.t2.y * pwv[1], pwv[1]);
              Vv[2] = make_float4(cur.t0.z * pwv[2], cur.t1.z * pwv[2],
                                  cur.t2.z * pwv[2], pwv[2]);
              Vv[3] = make_float4(cur.t0.w * pwv[3], cur.t1.w * pwv[3],
                                  cur.t2.w * pwv[3], pwv[3]);
            }
          }
          // next layer's loads in flight during the LDS phase, no register copy
          // The derived values are pinned here so that the projection is not
          // sunk below the loads: the loads then overwrite dead registers and
          // need no copies.
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            asm volatile("" : "+v"(Vv[i].x), "+v"(Vv[i].y), "+v"(Vv[i].z),
                              "+v"(Vv[i].w), "+v"(x0v[i]), "+v"(w0v[i]),
                              "+v"(w1v[i]));
          }
          if (l + 1 < l_end && !(dbg & 256)) load_layer(cur);  // 256: timing

          // ---- LDS phase, pixel by pixel (cells of one lane's pixels overlap)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x0 = x0v[i];
            float w0 = w0v[i], w1 = w1v[i];
            const float4 V = Vv[i];
            // left-cell offset in the window (+-Inf saturates; NaN gives
            // offset 0 but both its side weights are then clamped to 0 below)
            const int cl = (int)(x0 - wlo_f);
#ifdef LSI_EXPERIMENT_NOCHECK  // timing experiment only: results are wrong
            {
              float4* cellx = rb + (cl & 127);
              if (inrange) {
                cellx[0] = f4_fma(cellx[0], V, w0);
                LSI_COMPILER_FENCE();
                cellx[1] = f4_fma(cellx[1], V, w1);
              }
              LSI_COMPILER_FENCE();
              continue;
            }
#endif
            // in-window lanes: both cells inside the (in-image) window.  Each
            // ballot is taken straight from one compare; the masks are
            // combined with scalar ops.
            const bool in_b = (unsigned)cl <= wspan;
            const bool inw = in_b && inrange && win_ok;
            const unsigned long long inw_mask = __ballot(in_b) & inr_mask & win_mask;
            // lane l-1's floor(X) by DPP wave_shr:1 (VALU, no LDS round trip)
            const float prev = __int_as_float(__builtin_amdgcn_mov_dpp(
                __float_as_int(x0), 0x138, 0xf, 0xf, true));
            const unsigned long long mono_ok = __ballot(x0 > prev) | edge_mask;
            // clamped sides: !(p > 1e-3) is also true for NaN weights
            const bool c0 = !(w0 * wymin > 1.0e-3f);
            const bool c1 = !(w1 * wymin > 1.0e-3f);
            const unsigned long long clamp_mask =
                (__ballot(c0) | __ballot(c1)) & inw_mask;
            if (has_max) {
              if (clamp_mask != 0ull) {
                const float k0 = w0 * wymax, k1 = w1 * wymax;
                if (inw && c0 && k0 > 1.0e-3f) {
                  float* e = emax + cl * 4;
                  atomic_add_f32(e + 0, V.x * k0);
                  atomic_add_f32(e + 1, V.y * k0);
                  atomic_add_f32(e + 2, V.z * k0);
                  atomic_add_f32(e + 3, V.w * k0);
                }
                if (inw && c1 && k1 > 1.0e-3f) {
                  float* e = emax + cl * 4 + 4;
                  atomic_add_f32(e + 0, V.x * k1);
                  atomic_add_f32(e + 1, V.y * k1);
                  atomic_add_f32(e + 2, V.z * k1);
                  atomic_add_f32(e + 3, V.w * k1);
                }
              }
            }
            // lanes outside the window: exact slow path (cells outside the
            // image and non-finite X fail its range tests and add nothing)
            if ((inr_mask & ~inw_mask) != 0ull && !(dbg & 1)) {
              if (inrange && !inw && V.w != 0.0f)
                slow_corners(extras, V, x0, w0, w1, xmax, wy0, wy1, t_row0,
                             rows, Wt);
            }
            if (c0) w0 = 0.0f;
            if (c1) w1 = 0.0f;
            float4* cell = rb + cl;  // dereferenced by in-window lanes only
            if (mono_ok == ~0ull || (dbg & 2)) {
              if (inw && !(dbg & 128)) {  // dbg 128: no window traffic (timing)
                cell[0] = f4_fma(cell[0], V, w0);
                LSI_COMPILER_FENCE();
                cell[1] = f4_fma(cell[1], V, w1);
              }
              LSI_COMPILER_FENCE();
            } else if (inw_mask != 0ull) {
              // some lanes may share cells: serialise by arrival rank
              unsigned* cn = cnt + cl;
              unsigned rank = 0u;
              if (inw) rank = atomicAdd(cn, 1u);
              for (unsigned r = 0;; ++r) {
                if (__ballot(inw && rank >= r) == 0ull) break;
                if (inw && rank == r) {
                  cell[0] = f4_fma(cell[0], V, w0);
                  LSI_COMPILER_FENCE();
                  cell[1] = f4_fma(cell[1], V, w1);
                }
                LSI_COMPILER_FENCE();
              }
              if (inw) *cn = 0u;
              LSI_COMPILER_FENCE();
            }
          }
        }
      }
      if (tdbg && lane == 0 && step == 0) tdbg[16 + wave] = (long long)__builtin_readcyclecounter();
      LSI_TSTAMP();
      __syncthreads();
      LSI_TSTAMP();

      // ================= merge: cell owners gather the windows =============
      if (tid == 0) yrange[2] = 0;  // next step's tickets (no draws until then)
      {
        const UnitArgs ua = unit_args();
        const int NB = ua.nb, nunits = rows * NB;
        // lane t holds task slot t's table entry; the slots that touch a unit
        // are found with one ballot and their entries broadcast by readlane
        TaskA mine;
        mine.row0 = -1000000; mine.wy0 = 0.f; mine.wy1 = 0.f; mine.win = 0;
        if (lane < NWIN) mine = taskA[sidx * NWIN + lane];
        const int mine_wlo = mine.win & 0xffff, mine_wwin = mine.win >> 16;
        for (int unit = wave; unit < nunits; unit += NW) {
          const int r = div_small(unit, NB, ua.inv_nb);
          const int c0 = (unit - r * NB) * 64;
          const int cell = c0 + lane;
          const bool hit =
              ((mine.row0 == r && mine.wy0 != 0.f) ||
               (mine.row0 + 1 == r && mine.wy1 != 0.f)) &&
              (mine_wlo <= c0 + 63) && (mine_wlo + mine_wwin > c0);
          unsigned long long todo = __ballot(hit);
          // entry t of the table, broadcast; value of window t at this lane's
          // cell (0 outside the window) and the row weight that applies
          auto fetch = [&](int t, float4& v, float& wy) {
            const int q_row0 = __builtin_amdgcn_readlane(mine.row0, t);
            const int q_win = __builtin_amdgcn_readlane(mine.win, t);
            const int q_wlo = q_win & 0xffff, q_wwin = q_win >> 16;
            const float q_wy0 = __int_as_float(
                __builtin_amdgcn_readlane(__float_as_int(mine.wy0), t));
            const float q_wy1 = __int_as_float(
                __builtin_amdgcn_readlane(__float_as_int(mine.wy1), t));
            wy = (q_row0 == r) ? q_wy0 : q_wy1;
            const int rel = cell - q_wlo;
            const bool in = rel >= 0 && rel < q_wwin && cell < Wt;
            v = rb_all[t * WMAX + (in ? rel : 0)];
            if (!in) wy = 0.0f;
          };
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          const bool any = todo != 0ull;
          while (todo) {  // two independent LDS reads in flight per iteration
            const int t0 = __builtin_ctzll(todo);
            todo &= todo - 1;
            float4 va, vb = make_float4(0.f, 0.f, 0.f, 0.f);
            float wa, wb = 0.0f;
            fetch(t0, va, wa);
            if (todo) {
              const int t1 = __builtin_ctzll(todo);
              todo &= todo - 1;
              fetch(t1, vb, wb);
            }
            acc = f4_fma(acc, va, wa);
            acc = f4_fma(acc, vb, wb);
          }
          if (any && cell < Wt) {  // the unit's owner adds into the tile
            float4* tcell = reinterpret_cast<float4*>(extras) + r * Wt + cell;
            float4 tv = *tcell;
            tv.x += acc.x; tv.y += acc.y; tv.z += acc.z; tv.w += acc.w;
            *tcell = tv;
          }
        }
      }
      LSI_TSTAMP();
      // windows and the task table are reused by the next step; after the
      // last one every wave goes on to its own cells' epilogue
      if (step + 1 < nstep) __syncthreads();
      LSI_TSTAMP();
    }

    // ================= epilogue for this pass ===============================
    // (each wave finishes the units it merged: no barrier needed before)
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
      float4* tcell = reinterpret_cast<float4*>(extras) + r * Wt + cell;
      const float4 A = *tcell;
      if (r == 0 && top_shared) {
        store_coherent(xrow(band, 1) + cell, A);      // lower band's share
      } else if (r == R && bot_shared) {
        store_coherent(xrow(band + 1, 0) + cell, A);  // upper band's share
      } else {
        finish(r, cell, A);
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
        yrange[4] = top_shared
                        ? __hip_atomic_fetch_add(&ea.xcount[xb + band], 1,
                                                 __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)
                        : 0;
        yrange[5] = bot_shared
                        ? __hip_atomic_fetch_add(&ea.xcount[xb + band + 1], 1,
                                                 __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)
                        : 0;
      }
      __syncthreads();
      const bool fin_top = top_shared && yrange[4] == 1;
      const bool fin_bot = bot_shared && yrange[5] == 1;
      if (fin_top || fin_bot) {
        for (int unit = wave; unit < nunits; unit += NW) {
          const int r = div_small(unit, NB, ua.inv_nb);
          const int cell = (unit - r * NB) * 64 + lane;
          if (cell >= Wt) continue;
          const bool top = (r == 0 && fin_top), bot = (r == R && fin_bot);
          if (!top && !bot) continue;
          const float4 mine_v =
              *(reinterpret_cast<float4*>(extras) + r * Wt + cell);
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
          if (top_shared)
            reinterpret_cast<float4*>(extras)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bot_shared)
            reinterpret_cast<float4*>(extras)[R * Wt + i] =
                make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    LSI_TSTAMP();
    if (pass + 1 < npass) __syncthreads();
  }
}

// task-table entries: whole steps, about 256 tasks per refill
int stream_cap(int nw, int tpw) {
  const int nwin = nw * tpw;
  return nwin * (256 / nwin > 0 ? 256 / nwin : 1);
}

size_t stream_lds_bytes(const LsiSplatDesc* d, int tile_rows, int nw, int wmax,
                        int tpw) {
  return (size_t)nw * tpw * wmax * 16 + (size_t)nw * wmax * 4 +
         (size_t)tile_rows * d->Wt * 16 +
         (size_t)stream_cap(nw, tpw) * (sizeof(TaskA) + sizeof(TaskB)) + 32;
}

// layout class of the texture strides: 0 channels-last, 1 planar, -1 neither
int tex_layout(const LsiSplatDesc* d) {
  const bool al = (d->tex_sl % 4 == 0) && (d->tex_sb % 4 == 0) &&
                  (d->tex_sy % 4 == 0);
  if (!al) return -1;
  if (d->tex_sc == 1 && d->tex_sx == 3) return 0;
  if (d->tex_sx == 1 && d->tex_sc % 4 == 0) return 1;
  return -1;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int lsi_stream_ok(const LsiSplatDesc* d, const float* M) {
  if (!d || !M) return 0;
  if (!lsi_rowband_ok(d, M)) return 0;
  if (d->flags & LSI_WANT_DISP) return 0;
  if (d->W % 4 != 0) return 0;
  if (d->Wt > 32767) return 0;  // window origin is kept in 16 bits
  {  // the kernel keeps element strides in 32 bits
    const int64_t st[] = {d->tex_sl, d->tex_sb, d->tex_sy, d->tex_sc, d->disp_sl,
                          d->disp_sb, d->disp_sy, d->mask_sl, d->mask_sb,
                          d->mask_sy};
    for (int64_t v : st)
      if (v < 0 || v > 0x7fffffffLL) return 0;
  }
  if (tex_layout(d) < 0) return 0;
  if (d->disp_sx != 1 || d->disp_sy % 4 || d->disp_sb % 4 || d->disp_sl % 4)
    return 0;
  if ((d->flags & LSI_HAS_MASK) &&
      (d->mask_sx != 1 || d->mask_sy % 4 || d->mask_sb % 4 || d->mask_sl % 4))
    return 0;
  const float s = d->trg_downsampling;
  float need = 0.0f;
  bool simple = true;
  for (int b = 0; b < d->B; ++b) {
    const float* m = M + 16 * b;
    if (m[4] != 0.0f || m[8] != 0.0f) return 0;  // M[1][0], M[2][0]
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
  int win = (int)ceilf(need) + 8;
  win = (win + 63) / 64 * 64;
  if (win < 64) win = 64;
  if (win > 512) win = 512;  // beyond this the excess takes the exact slow path
  return win | (simple ? LSI_STREAM_SIMPLE_BIT : 0);
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

// Band height, waves and windows per workgroup.  One workgroup runs per CU at
// a time (registers), so the cost model is: rounds of workgroups x (fixed
// prologue/epilogue + per-step barriers and merge + the longest chain of tasks
// on a SIMD).  Units: shader cycles, fitted to tools/phase_probe.py timelines.
static int stream_plan(const LsiSplatDesc* d, int wmax, StreamCfg* cfg,
                       int* nw_out) {
  const int nseg = (d->W + SEG - 1) / SEG;
  const int tpw_override = (d->reserved >> 12) & 0xf;  // experiments only
  static const char* cap_env = getenv("LSI_STREAM_LDS_CAP");  // experiments
  const size_t lds_cap = cap_env ? (size_t)atol(cap_env) : 156 * 1024;
  double best = -1.0;
  int bR = 0, bnw = 0, btpw = 0, bx = 0;
  const int layers = (d->flags & LSI_COMPOSE) ? d->L : 1;
  const int npass = (d->flags & LSI_COMPOSE) ? 1 : d->L;
  const int force_mode = (d->reserved >> 16) & 3;  // experiments: 1 halo, 2 exchange
  for (int xch = 0; xch <= 1; ++xch) {
    if (force_mode && xch != force_mode - 1) continue;
    for (int R = 1; R <= 64; R *= 2) {
      if (d->tune_rows > 0 && R != d->tune_rows) continue;
      if (d->tune_rows <= 0 && R > 1 && R / 2 >= d->Ht) break;
      const long nwg = (long)((d->Ht + R - 1) / R) * d->B;
      const long rounds = (nwg + 255) / 256;
      // source rows per band ~ R / s (one more target row's worth when the
      // band re-reads its halo); one task per (row, 256-pixel segment)
      const int ntask =
          (int)ceilf((float)(R + 1 - xch) / d->trg_downsampling) * nseg;
      // a task alone on a SIMD is latency-bound (~500 cycles per pixel
      // iteration: two dependent LDS round trips); the CU's four SIMDs issue
      // one pixel iteration per ~42 cycles when enough waves share them
      const double t_lat = layers * 4 * 500.0, t_issue = layers * 4 * 42.5;
      for (int c = MAXNW; c >= 4; --c) {
        if (d->tune_threads > 0 && c != (d->tune_threads + 63) / 64) continue;
        for (int t = 1; t <= 8 && c * t <= 64; ++t) {
          if (tpw_override && t != tpw_override) continue;
          if (stream_lds_bytes(d, R + xch, c, wmax, t) > lds_cap) break;
          const int steps = (ntask + c * t - 1) / (c * t);
          const double per_step = (double)ntask / steps;  // tasks in a step
          // tickets balance the waves; the step still ends ~a third of a
          // wave's share after the average wave (measured), then merges
          const double step_cost = 6000.0 + 0.35 * (per_step / c) * t_lat;
          const double xpass =
              fmax((double)ntask / c * t_lat, ntask * t_issue) + 0.5 * t_lat;
          const double est =
              (double)rounds *
              (8000.0 +
               npass * (3000.0 + 6500.0 * xch + steps * step_cost + xpass));
          if (best < 0.0 || est < best) {
            best = est; bR = R; bnw = c; btpw = t; bx = xch;
          }
        }
      }
    }
  }
  if (bnw == 0) return LSI_EINVAL;
  cfg->R = bR;
  cfg->tpw = btpw;
  cfg->exchange = bx;
  *nw_out = bnw;
  static const bool verbose = getenv("LSI_STREAM_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "lsi stream plan: R=%d waves=%d windows/wave=%d %s est=%.0f "
            "cycles lds=%zu\n", bR, bnw, btpw, bx ? "exchange" : "halo", best,
            stream_lds_bytes(d, bR + bx, bnw, wmax, btpw));
  return LSI_OK;
}

int lsi_stream_launch(const SplatArgs& a, hipStream_t stream) {
  const LsiSplatDesc* d = &a.d;
  const int layout = tex_layout(d);
  if (layout < 0 || (d->flags & LSI_WANT_DISP) || d->W % 4 != 0)
    return LSI_EINVAL;
  if (!aligned16(a.tex) || !aligned16(a.disp) ||
      ((d->flags & LSI_HAS_MASK) && !aligned16(a.mask)))
    return LSI_EINVAL;
  if ((d->tune_window & ~LSI_STREAM_SIMPLE_BIT) <= 0)
    return LSI_EINVAL;  // from lsi_stream_ok
  const int NB = (d->Wt + 63) / 64;
  StreamCfg cfg;
  cfg.wmax = d->tune_window & ~LSI_STREAM_SIMPLE_BIT;
  int nw = 0;
  if (stream_plan(d, cfg.wmax, &cfg, &nw) != LSI_OK) return LSI_EINVAL;
  const int R = cfg.R, tpw = cfg.tpw;
  const int threads = nw * 64;
  const size_t lds = stream_lds_bytes(d, R + cfg.exchange, nw, cfg.wmax, tpw);
  if (lds > 160 * 1024) return LSI_EINVAL;
  cfg.cap = stream_cap(nw, tpw);
  cfg.nb = NB;
  cfg.steps_per_chunk = cfg.cap / (nw * tpw);
  cfg.inv_nb = 1.0f / (float)NB;
  cfg.inv_nwin = 1.0f / (float)(nw * tpw);
  cfg.inv_gx = 1.0f / (float)((d->Ht + R - 1) / R);
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
    if (a.ws_bytes < off + (size_t)x.nbands * d->B * 32 * 8) return LSI_EWORKSPACE;
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
  const int mode = lean ? (cfg.exchange ? 2 : 1) : 0;
  const void* fn;
#define LSI_PICK2(L_, S_, M_)                                              \
  (full ? (const void*)splat_stream_kernel<L_, S_, M_, true>                \
        : (const void*)splat_stream_kernel<L_, S_, M_, false>)
#define LSI_PICK(L_, S_) \
  (mode == 1 ? LSI_PICK2(L_, S_, 1) : mode == 2 ? LSI_PICK2(L_, S_, 2) : LSI_PICK2(L_, S_, 0))
  const bool full = d->W % SEG == 0;
  if (layout == 0)
    fn = simple ? LSI_PICK(0, true) : LSI_PICK(0, false);
  else
    fn = simple ? LSI_PICK(1, true) : LSI_PICK(1, false);
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
