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
// Structure.  Workgroup = (band of R target rows, batch element b), NW waves.
//   task  = (source row y, 256-pixel segment j), all layers of the pass; every
//           step each wave runs one task:
//     x-pass  lanes load 4 consecutive pixels (dwordx4), project them, and add
//             V*wx0 / V*wx1 (V = (r,g,b,1)*pixel weight) into the wave's
//             PRIVATE window of float4 cells in LDS by plain RMW.  Lanes are 4
//             pixels apart, so their cells are distinct whenever floor(X) is
//             strictly increasing across the wave (checked); otherwise the
//             lanes are ranked per cell with an integer LDS atomic and the RMW
//             is issued rank by rank.
//     barrier
//     merge   every target cell of the band is owned by one lane, which keeps
//             its accumulator in REGISTERS and adds window[cell] * wy of each
//             task that touches its row.  No shared accumulation tile, no
//             locks, deterministic summation order.
//     barrier
//   Corners that the factorisation cannot represent exactly (clamped products,
//   cells outside the window because the disparity leaves [0, max_disp]) are
//   added with fp32 LDS atomics into a small `extras` tile -- exact for any
//   input, slow only when such corners are common.
//   epilogue: (acc + extras + background) normalised, each output written once.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"
#include "lsi_splat_internal.h"

#pragma clang fp contract(off)

using namespace lsi;

namespace {

constexpr int SEG = 256;   // source pixels per task (64 lanes x 4)
constexpr int MAXU = 3;    // target-cell units (64 cells) owned per wave
constexpr int MAXNW = 12;
// lsi_stream_ok's return value: window cells, plus this bit when every batch
// element has normaliser == 1 and M row 3 == (0,0,0,1) (division-free kernel)
constexpr int LSI_STREAM_SIMPLE_BIT = 1 << 20;

struct TaskInfo {
  int row0;        // target row of the task's top contribution, band-relative
  float wy0, wy1;  // row weights incl. border masks (sampling.py:210-211)
  int wlo, wwin;   // window: absolute first cell, number of cells
};

struct StreamCfg {
  int R;      // target rows per workgroup
  int wmax;   // window cells per task
  int tpw;    // tasks (windows) per wave per step
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

// accumulations are not index-critical: fused multiply-add
__device__ __forceinline__ float4 f4_fma(float4 t, float4 v, float w) {
  t.x = __fmaf_rn(v.x, w, t.x); t.y = __fmaf_rn(v.y, w, t.y);
  t.z = __fmaf_rn(v.z, w, t.z); t.w = __fmaf_rn(v.w, w, t.w);
  return t;
}

// Exact slow path for one source pixel (rare): recomputes the reference's
// x-axis footprint with clipped cells and adds the up-to-four corners with
// their exact weights clamp(wx*wy) into the extras tile by fp32 LDS atomics.
__device__ __forceinline__ void slow_corners(float* extras, float4 V, float X,
                                             float xmax, float wy0, float wy1,
                                             int row0, int rows, int Wt) {
  const Axis ax = splat_axis(X, xmax);
  const float wc[4] = {clamp_small(ax.w0 * wy0), clamp_small(ax.w1 * wy0),
                       clamp_small(ax.w0 * wy1), clamp_small(ax.w1 * wy1)};
  const int cx[2] = {(int)ax.c0s, (int)ax.c1s};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = row0 + (k >> 1);
    if (wc[k] == 0.0f || r < 0 || r >= rows) continue;
    float* e = extras + ((size_t)r * Wt + cx[k & 1]) * 4;
    atomic_add_f32(e + 0, V.x * wc[k]);
    atomic_add_f32(e + 1, V.y * wc[k]);
    atomic_add_f32(e + 2, V.z * wc[k]);
    atomic_add_f32(e + 3, V.w * wc[k]);
  }
}

// SIMPLE: the normaliser is exactly 1 and row 3 of M is (0,0,0,1) for every
// batch element (rectified stereo): u = q0 and D = d with no division.
template <int LAYOUT, bool SIMPLE>  // LAYOUT 0: channels-last RGB, 1: planar
__global__ __launch_bounds__(768) void splat_stream_kernel(SplatArgs a,
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
    band = id % gridDim.x;
    b = id / gridDim.x;
  }
  const int row0 = band * R;
  const int rows = min(R, Ht - row0);
  const int NB = (Wt + 63) >> 6;
  const int nunits = rows * NB;

  float4* rb_all = reinterpret_cast<float4*>(smem_raw);  // [NWIN][WMAX]
  unsigned* cnt_all = reinterpret_cast<unsigned*>(rb_all + NWIN * WMAX);
  float* extras = reinterpret_cast<float*>(cnt_all + NW * WMAX);  // [R][Wt][4]
  TaskInfo* tinfo = reinterpret_cast<TaskInfo*>(extras + R * Wt * 4);  // [64]
  int* yrange = reinterpret_cast<int*>(tinfo + 64);
  unsigned* cnt = cnt_all + wave * WMAX;

  // Everything read from global / kernarg memory inside the loops is copied to
  // registers first: the LDS ordering fences below are compiler memory
  // barriers and would otherwise force re-loads in the hot loop.
  float m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = a.M[16 * b + k];
  const float s = d.trg_downsampling;
  const float max_disp = d.max_disp, zscale = d.zbuf_scale;
  const long tex_sl = d.tex_sl, tex_sb = d.tex_sb, tex_sy = d.tex_sy,
             tex_sc = d.tex_sc;
  const long disp_sl = d.disp_sl, disp_sb = d.disp_sb, disp_sy = d.disp_sy;
  const long mask_sl = d.mask_sl, mask_sb = d.mask_sb, mask_sy = d.mask_sy;
  const float* __restrict__ g_tex = a.tex;
  const float* __restrict__ g_disp = a.disp;
  const float* __restrict__ g_mask = a.mask;
  const int dbg = d.reserved;
  const int nlayers = d.L;
  const float xmax = (float)Wt - 1.0f, ymax = (float)Ht - 1.0f;
  const bool has_mask = d.flags & LSI_HAS_MASK;
  const bool compose = d.flags & LSI_COMPOSE;
  const float inv_md = div_rn(1.0f, max_disp);

  long long* tdbg = (a.d.reserved & 4)
                        ? reinterpret_cast<long long*>(a.canvas) +
                              ((size_t)b * gridDim.x + band) * 32
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
  for (int i = tid; i < R * Wt * 4; i += T) extras[i] = 0.0f;
  if (tid == 0) { yrange[0] = d.H; yrange[1] = -1; }
  __syncthreads();
  {  // source rows whose target rows (y0, y0+1) intersect the band
    int lo = d.H, hi = -1;
    for (int y = tid; y < d.H; y += T) {
      float nd;
      const float Y = row_Y(y, nd);
      if (!finite_f(Y)) continue;
      const float y0 = floorf(Y);
      if (y0 >= (float)(row0 - 1) && y0 <= (float)(row0 + rows - 1)) {
        lo = min(lo, y); hi = max(hi, y);
      }
    }
    if (hi >= 0) { atomicMin(&yrange[0], lo); atomicMax(&yrange[1], hi); }
  }
  __syncthreads();
  LSI_TSTAMP();
  const int y_lo = yrange[0], y_hi = yrange[1];
  const int nsrc = (y_hi >= y_lo) ? (y_hi - y_lo + 1) : 0;
  const int nseg = (W + SEG - 1) / SEG;
  const float inv_nseg = 1.0f / (float)nseg;
  const float bg = d.bg_wt;
  const size_t P = (size_t)Ht * Wt;

  const int npass = compose ? 1 : nlayers;
  const int Lp = compose ? nlayers : 1;
  for (int pass = 0; pass < npass; ++pass) {
    float4 acc[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int l_begin = compose ? 0 : pass;
    const int ntask = nsrc * nseg;
    const int nstep = (ntask + NWIN - 1) / NWIN;

    for (int step = 0; step < nstep; ++step) {
      // ================= x-pass: TPW tasks per wave =========================
      // task = (source row y, 256-pixel segment j), all layers of the pass;
      // task slot k*NW + wave owns LDS window [slot]
      for (int k = 0; k < TPW; ++k) {
        const int slot = k * NW + wave;
        const int tg = step * NWIN + slot;
        TaskInfo ti;
        ti.row0 = -1000000; ti.wy0 = 0.f; ti.wy1 = 0.f; ti.wlo = 0; ti.wwin = 0;
        int y = 0, xs = 0;
        float nden = 1.0f;
        bool tvalid = tg < ntask;
        if (tvalid) {
          // tg / nseg without an integer division (tg < 2^20: exact in fp32)
          const int yi = (int)(((float)tg + 0.5f) * inv_nseg);
          const int j = tg - yi * nseg;
          y = y_lo + yi;
          xs = j * SEG;
          const float py = (float)y + 0.5f;
          const float Y = row_Y(y, nden);
          tvalid = false;
          if (finite_f(Y) && fabsf(Y) < 1.0e7f) {
            const Axis ay = splat_axis(Y, ymax);
            ti.row0 = (int)floorf(Y) - row0;
            ti.wy0 = ay.w0;
            ti.wy1 = ay.w1;
            tvalid = (ay.w0 != 0.0f) || (ay.w1 != 0.0f);
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
              // cells [wlo, wlo+wwin) confined to the image: a fast lane then
              // needs no border mask (both its cells are valid)
              const int c_lo = max((int)floorf(lo) - 1, 0);
              const int c_hi = min((int)floorf(hi) + 2, Wt - 1);
              ti.wlo = c_lo;
              ti.wwin = max(0, min(WMAX, c_hi - c_lo + 1));
            }
          }
        }
        if (lane == 0) tinfo[slot] = ti;
        if (!tvalid) continue;
        LSI_TSTAMP();

        float4* rb = rb_all + slot * WMAX;
        const int x = xs + 4 * lane;
        const bool inrange = x < W;
        const float py = (float)y + 0.5f;
        // row-uniform pieces of q = M p, in the contract's rounding order
        const float pym01 = py * m[1];
        const float pym31 = py * m[13];
        const float rn = SIMPLE ? 1.0f : div_rn(1.0f, nden);
        const float wy0 = ti.wy0, wy1 = ti.wy1;
        // smallest non-zero row weight: a side is exactly factorisable iff its
        // product with this one survives the 1e-3 clamp (rounding is monotone)
        const float wymin =
            (wy0 == 0.f) ? wy1 : ((wy1 == 0.f) ? wy0 : fminf(wy0, wy1));
        // the larger x weight is >= ~0.5: its products with the row weights
        // survive the clamp unless a row weight is itself tiny -- then the
        // whole task takes the exact path
        const bool wy_small = wymin <= 2.1e-3f;
        const float wlo_f = (float)ti.wlo;
        const float whi_f = (float)(ti.wlo + ti.wwin - 2);  // last left cell

        struct PxData { float4 d4, t0, t1, t2, m4; };
        auto load_layer = [&](int l, PxData& o) {
          if (!inrange) return;
          o.d4 = *reinterpret_cast<const float4*>(
              g_disp + l * disp_sl + b * disp_sb + y * disp_sy + x);
          const float* tp = g_tex + l * tex_sl + b * tex_sb + y * tex_sy;
          if (LAYOUT == 0) {
            const float4* t4 = reinterpret_cast<const float4*>(tp + 3 * x);
            o.t0 = t4[0]; o.t1 = t4[1]; o.t2 = t4[2];
          } else {
            o.t0 = *reinterpret_cast<const float4*>(tp + x);
            o.t1 = *reinterpret_cast<const float4*>(tp + tex_sc + x);
            o.t2 = *reinterpret_cast<const float4*>(tp + 2 * tex_sc + x);
          }
          if (has_mask)
            o.m4 = *reinterpret_cast<const float4*>(
                g_mask + l * mask_sl + b * mask_sb + y * mask_sy + x);
        };

        // exact threshold test for the clamp: a side's products survive iff
        // fl(w*wymin) > 1e-3; w > thr is a (slightly conservative) sufficient
        // condition evaluated with one compare per side
        const float thr = wy_small ? __builtin_inff()
                                   : div_rn(1e-3f, wymin) * 1.000001f;
        const int wspan = ti.wwin - 2;  // last admissible left-cell offset

        // One layer of the lane's 4 pixels.  Hot path: every lane of the wave
        // is either out of range or "fast" (both cells inside the in-image
        // window, no clamped corner) and floor(X) is strictly increasing
        // across the wave -> plain RMW.  Anything else takes general_px().
        auto do_layer = [&](const PxData& cur) {
          const float dv[4] = {cur.d4.x, cur.d4.y, cur.d4.z, cur.d4.w};
          float cr[4], cg[4], cb[4];
          if (LAYOUT == 0) {
            cr[0] = cur.t0.x; cg[0] = cur.t0.y; cb[0] = cur.t0.z;
            cr[1] = cur.t0.w; cg[1] = cur.t1.x; cb[1] = cur.t1.y;
            cr[2] = cur.t1.z; cg[2] = cur.t1.w; cb[2] = cur.t2.x;
            cr[3] = cur.t2.y; cg[3] = cur.t2.z; cb[3] = cur.t2.w;
          } else {
            cr[0] = cur.t0.x; cr[1] = cur.t0.y; cr[2] = cur.t0.z; cr[3] = cur.t0.w;
            cg[0] = cur.t1.x; cg[1] = cur.t1.y; cg[2] = cur.t1.z; cg[3] = cur.t1.w;
            cb[0] = cur.t2.x; cb[1] = cur.t2.y; cb[2] = cur.t2.z; cb[3] = cur.t2.w;
          }
          // ---- projection of the 4 pixels as two packed pairs (v_pk_*_f32) --
          float x0v[4], w0v[4], w1v[4], pwv[4];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f2 px = {(float)(x + 2 * h) + 0.5f, (float)(x + 2 * h) + 1.5f};
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
            // weights are not index-critical: reciprocal multiplies (<= 2 ulp)
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

          // ---- LDS phase, pixel by pixel (cells of one lane's pixels overlap)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x0 = x0v[i], w0 = w0v[i], w1 = w1v[i], pw = pwv[i];
            // left-cell offset in the window; +-Inf saturates, NaN has pw == 0
            const int cl = (int)(x0 - wlo_f);
            const bool fast = inrange && ((unsigned)cl <= (unsigned)wspan) &&
                              (w0 > thr) && (w1 > thr);
            const float4 V = make_float4(cr[i] * pw, cg[i] * pw, cb[i] * pw, pw);
            // lane l-1's floor(X) by DPP wave_shr:1 (VALU, no LDS round trip)
            const float prev = __int_as_float(__builtin_amdgcn_update_dpp(
                0, __float_as_int(x0), 0x138, 0xf, 0xf, false));
            const bool mono_lane = (lane == 0) || !inrange || (x0 > prev);
            const bool clean = ((__ballot(mono_lane && (fast || !inrange)) == ~0ull)
                                && !(dbg & 16)) || (dbg & 32);
            float4* cell = rb + (fast ? cl : 0);
            if (clean) {
              if (fast) {
                cell[0] = f4_fma(cell[0], V, w0);
                LSI_COMPILER_FENCE();
                cell[1] = f4_fma(cell[1], V, w1);
              }
              LSI_COMPILER_FENCE();
              continue;
            }
            // ---- general px: exact slow corners + (ranked) RMW ---------------
            {
              // lanes that are not fast but can still touch the image: exact
              // slow path.  (non-finite X fails the range tests: dropped)
              const bool slow = inrange && !fast && (pw != 0.0f) &&
                                (x0 >= -1.0f) && (x0 <= xmax);
              if (__ballot(slow) != 0ull && !(dbg & 1)) {
                if (slow) {
                  // X recomputed exactly as in the projection above
                  float q0 = ((float)(x + i) + 0.5f) * m[0] + pym01;
                  q0 = q0 + m[2];
                  q0 = q0 + dv[i] * m[3];
                  const float u = SIMPLE ? q0 : div_rn(q0, nden);
                  slow_corners(extras, V, u * s - 0.5f, xmax, wy0, wy1, ti.row0,
                               rows, Wt);
                }
              }
              const bool mono = (__ballot(mono_lane) == ~0ull) || (dbg & 2);
              if (mono) {
                if (fast) {
                  cell[0] = f4_fma(cell[0], V, w0);
                  LSI_COMPILER_FENCE();
                  cell[1] = f4_fma(cell[1], V, w1);
                }
                LSI_COMPILER_FENCE();
              } else if (__ballot(fast) != 0ull) {
                unsigned* cn = cnt + (fast ? cl : 0);
                unsigned rank = 0u;
                if (fast) rank = atomicAdd(cn, 1u);
                for (unsigned r = 0;; ++r) {
                  if (__ballot(fast && rank >= r) == 0ull) break;
                  if (fast && rank == r) {
                    cell[0] = f4_fma(cell[0], V, w0);
                    LSI_COMPILER_FENCE();
                    cell[1] = f4_fma(cell[1], V, w1);
                  }
                  LSI_COMPILER_FENCE();
                }
                if (fast) *cn = 0u;
                LSI_COMPILER_FENCE();
              }
            }
          }
        };

        // the next layer's loads are in flight while the current one is
        // processed (one copy of the loop body: the kernel must stay small
        // enough for the instruction cache)
        PxData nxt;
        nxt.d4 = nxt.t0 = nxt.t1 = nxt.t2 = make_float4(0.f, 0.f, 0.f, 0.f);
        nxt.m4 = make_float4(1.f, 1.f, 1.f, 1.f);
        const int l_end = l_begin + Lp;
        load_layer(l_begin, nxt);
        // zero this task's window while the first loads are in flight (same
        // wave, in-order LDS: no barrier needed before its own RMWs)
        for (int c = lane; c < ti.wwin; c += 64)
          rb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        LSI_COMPILER_FENCE();
        for (int l = l_begin; l < l_end; ++l) {
          const PxData cur = nxt;
          if (l + 1 < l_end) load_layer(l + 1, nxt);
          do_layer(cur);
        }
      }
      if (tdbg && lane == 0 && step == 0) tdbg[16 + wave] = (long long)__builtin_readcyclecounter();
      LSI_TSTAMP();
      __syncthreads();
      LSI_TSTAMP();

      // ================= merge: cell owners gather the windows =============
      {
        // lane t holds task slot t's table entry; the slots that touch a unit
        // are found with one ballot and their entries broadcast by readlane
        TaskInfo mine;
        mine.row0 = -1000000; mine.wy0 = 0.f; mine.wy1 = 0.f;
        mine.wlo = 0; mine.wwin = 0;
        if (lane < NWIN) mine = tinfo[lane];
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
          const int unit = wave + u * NW;
          if (unit >= nunits) continue;
          const int r = unit / NB;
          const int c0 = (unit - r * NB) * 64;
          const int cell = c0 + lane;
          const bool hit =
              ((mine.row0 == r && mine.wy0 != 0.f) ||
               (mine.row0 + 1 == r && mine.wy1 != 0.f)) &&
              (mine.wlo <= c0 + 63) && (mine.wlo + mine.wwin > c0);
          unsigned long long todo = __ballot(hit);
          // entry t of the table, broadcast; value of window t at this lane's
          // cell (0 outside the window) and the row weight that applies
          auto fetch = [&](int t, float4& v, float& wy) {
            const int q_row0 = __builtin_amdgcn_readlane(mine.row0, t);
            const int q_wlo = __builtin_amdgcn_readlane(mine.wlo, t);
            const int q_wwin = __builtin_amdgcn_readlane(mine.wwin, t);
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
            acc[u] = f4_fma(acc[u], va, wa);
            acc[u] = f4_fma(acc[u], vb, wb);
          }
        }
      }
      LSI_TSTAMP();
      __syncthreads();
      LSI_TSTAMP();
    }

    // ================= epilogue for this pass ===============================
    const float lbg = compose ? (float)nlayers * bg : bg;
    const int lo_ = compose ? 0 : pass;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int unit = wave + u * NW;
      if (unit >= nunits) continue;
      const int r = unit / NB;
      const int cell = (unit - r * NB) * 64 + lane;
      if (cell >= Wt) continue;
      float* e = extras + ((size_t)r * Wt + cell) * 4;
      const float A0 = (acc[u].x + e[0]) + lbg, A1 = (acc[u].y + e[1]) + lbg,
                  A2 = (acc[u].z + e[2]) + lbg, Wsum = (acc[u].w + e[3]) + lbg;
      const float wd = safe_den(Wsum);
      const size_t o =
          ((size_t)lo_ * d.B + b) * P + (size_t)(row0 + r) * Wt + cell;
      a.out_img[3 * o + 0] = div_rn(A0, wd);
      a.out_img[3 * o + 1] = div_rn(A1, wd);
      a.out_img[3 * o + 2] = div_rn(A2, wd);
      a.out_wts[o] = Wsum;
      if (pass + 1 < npass) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
    }
    LSI_TSTAMP();
    __syncthreads();
  }
}

size_t stream_lds_bytes(const LsiSplatDesc* d, int R, int nw, int wmax,
                        int tpw) {
  return (size_t)nw * tpw * wmax * 16 + (size_t)nw * wmax * 4 +
         (size_t)R * d->Wt * 16 + 64 * sizeof(TaskInfo) + 16;
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
  const int nseg = (d->W + SEG - 1) / SEG;
  StreamCfg cfg;
  cfg.wmax = d->tune_window & ~LSI_STREAM_SIMPLE_BIT;
  int R = d->tune_rows;
  if (R <= 0) {  // tallest band that still gives >= 256 workgroups
    R = 1;
    for (int c = 2; c <= 16; c *= 2) {
      if ((long)((d->Ht + c - 1) / c) * d->B < 256) break;
      R = c;
    }
  }
  const int tpw_override = (d->reserved >> 8) & 0xff;  // experiments only
  int nw = 0, tpw = 1;
  for (;;) {
    // source rows per band ~ (R + 1) / s; one task per (row, segment)
    const int ntask =
        (int)ceilf((float)(R + 1) / d->trg_downsampling) * nseg;
    long best = -1;
    nw = 0;
    for (int c = MAXNW; c >= 4; --c) {
      if (d->tune_threads > 0 && c != (d->tune_threads + 63) / 64) continue;
      if (R * NB > MAXU * c) continue;
      for (int t = 1; t <= 6 && c * t <= 64; ++t) {
        if (tpw_override && t != tpw_override) continue;
        if (stream_lds_bytes(d, R, c, cfg.wmax, t) > 150 * 1024) break;
        const int steps = (ntask + c * t - 1) / (c * t);
        // shortest per-wave chain of tasks, then fewest barrier rounds, then
        // more waves (idle waves are free; busy ones hide each other's latency)
        const long score = -(long)steps * t * 1000000 - (long)steps * 1000 + c;
        if (nw == 0 || score > best) { best = score; nw = c; tpw = t; }
      }
    }
    if (nw > 0) break;
    if (R == 1) return LSI_EINVAL;
    R /= 2;
  }
  const int threads = nw * 64;
  const size_t lds = stream_lds_bytes(d, R, nw, cfg.wmax, tpw);
  if (lds > 160 * 1024) return LSI_EINVAL;
  cfg.tpw = tpw;
  cfg.R = R;
  dim3 grid((d->Ht + R - 1) / R, d->B);
  const bool simple = (d->tune_window & LSI_STREAM_SIMPLE_BIT) != 0;
  const void* fn;
  if (layout == 0)
    fn = simple ? (const void*)splat_stream_kernel<0, true>
                : (const void*)splat_stream_kernel<0, false>;
  else
    fn = simple ? (const void*)splat_stream_kernel<1, true>
                : (const void*)splat_stream_kernel<1, false>;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return LSI_ELAUNCH;
  void* kargs[2] = {const_cast<SplatArgs*>(&a), &cfg};
  if (hipLaunchKernel(fn, grid, dim3(threads), kargs, lds, stream) != hipSuccess)
    return LSI_ELAUNCH;
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
