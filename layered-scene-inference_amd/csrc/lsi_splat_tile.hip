// LSI_PATH_TILE: forward splat for ANY projection matrix (general 3-D poses,
// BASELINE config 4) without floating-point atomics.
//
// Why: fp32 atomics -- global or LDS -- retire ~100-300 G/s on gfx950
// (tools/microbench*.hip); a splat issues 16 per source pixel, which makes the
// ATOMIC path ~2 ms at config 4.  Integer LDS atomics are ~20x faster, so the
// scatter is turned into a gather: source pixels are BINNED by their top-left
// target cell with one integer exchange each, and every target cell is then
// summed by the one thread that owns it.
//
// Workgroup = (tile of TH x TW target cells, batch element), one thread per
// cell (1024).  Per layer, the source rows that can reach the tile (a bound
// from the four (x, disparity) corners of each row, with the batch element's
// actual disparity range from a small reduction kernel) are swept in chunks of
// 2048 pixels:
//   A  every thread projects two pixels exactly (same arithmetic as the other
//      paths: indices are bit-exact); a pixel whose top-left cell (x0, y0) lies
//      in the tile or its one-cell upper/left halo loads its colour, writes a
//      32-byte record {c_tl, c_tr, c_bl, c_br, r*w, g*w, b*w, w} to LDS (c = the
//      clamped, border-masked corner weights; sampling.py:193-222) and links
//      it into the list of bin (y0, x0):  next[i] = exchange(head[bin], i).
//   B  the owner of cell (Y, X) walks the lists of bins (Y, X), (Y, X-1),
//      (Y-1, X), (Y-1, X-1) and adds V * c with the weight of the corner that
//      lands on it.
// Epilogue per cell: background, per-layer disparity, compose, normalisation
// (ldi.py:122-125, 157-182), each output written once.  Summation order within
// a cell follows list order (not run-to-run deterministic, like ATOMIC).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"
#include "lsi_splat_internal.h"

#pragma clang fp contract(off)

#ifndef LSI_STREAM_HOOKS
#define LSI_STREAM_HOOKS 0
#endif

using namespace lsi;

namespace {

constexpr int TT = 1024;  // threads per workgroup = cells per tile
constexpr int CH = 2048;  // source pixels per chunk (two per thread)

constexpr int MAXCPT = 4;  // target cells per thread

struct TileCfg {
  int tw_log2;   // tile width  TW = 1 << tw_log2 (32, 64 or 128)
  int th;        // tile height TH = cpt * 1024 / TW
  int tiles_x;   // tiles per row of tiles
  int cpt;       // cells per thread (1, 2 or 4): thread t owns cells t + k*1024
};

// Min and max of the finite-or-infinite disparities of a slice of rows of one
// (layer, batch element) (NaNs are skipped: such pixels are dropped by every
// path).  grid (B * L, LSI_RANGE_SLICES), block 256;
// range[(l * B + b) * LSI_RANGE_SLICES + slice]; consumers fold the slices.
__global__ __launch_bounds__(256) void disp_range_kernel(SplatArgs a,
                                                         float2* range, int vec4) {
  const LsiSplatDesc& d = a.d;
  const int l = blockIdx.x / d.B, b = blockIdx.x - l * d.B, sl = blockIdx.y;
  const int r0 = (int)((long)d.H * sl / LSI_RANGE_SLICES);
  const int r1 = (int)((long)d.H * (sl + 1) / LSI_RANGE_SLICES);
  float lo = __builtin_inff(), hi = -__builtin_inff();
  const float* base = a.disp + (long)l * d.disp_sl + (long)b * d.disp_sb;
  if (vec4) {  // unit pixel stride, 16-byte aligned rows, W % 4 == 0
    const int w4 = d.W >> 2, n4 = (r1 - r0) * w4;
    const float rcp_w4 = 1.0f / (float)w4;
    auto at = [&](int i) {  // i < 2^22 (H * W < 2^24): exact in fp32
      int y = (int)((float)i * rcp_w4);
      int x4 = i - y * w4;
      if (x4 < 0) { --y; x4 += w4; }
      if (x4 >= w4) { ++y; x4 -= w4; }
      return reinterpret_cast<const float4*>(base + (long)(r0 + y) * d.disp_sy)[x4];
    };
    // four independent 16-byte loads in flight per thread
    for (int i = threadIdx.x; i < n4; i += 1024) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = at(min(i + 256 * k, n4 - 1));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lo = fminf(fminf(lo, v[k].x), fminf(fminf(v[k].y, v[k].z), v[k].w));  // skip NaNs
        hi = fmaxf(fmaxf(hi, v[k].x), fmaxf(fmaxf(v[k].y, v[k].z), v[k].w));
      }
    }
  } else {
    for (int y = r0; y < r1; ++y)
      for (int x = threadIdx.x; x < d.W; x += 256) {
        const float v = base[(long)y * d.disp_sy + (long)x * d.disp_sx];
        lo = fminf(lo, v);  // fminf / fmaxf return the non-NaN operand
        hi = fmaxf(hi, v);
      }
  }
  __shared__ float slo[4], shi[4];
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
    range[((size_t)l * d.B + b) * LSI_RANGE_SLICES + sl] = make_float2(lo, hi);
  }
}

template <bool WANT_DISP>
__global__ __launch_bounds__(1024) void splat_tile_kernel(
    SplatArgs a, TileCfg c, const float2* __restrict__ range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const LsiSplatDesc& d = a.d;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int TW = 1 << c.tw_log2, TH = c.th;
  const int tile_y = blockIdx.x / c.tiles_x, tile_x = blockIdx.x - tile_y * c.tiles_x;
  const int ty0 = tile_y * TH, tx0 = tile_x * TW;
  const int Ht = d.Ht, Wt = d.Wt, H = d.H, W = d.W;
  const int nbins = (TH + 1) * (TW + 1);

  float4* recA = reinterpret_cast<float4*>(smem);          // corner weights
  float4* recB = recA + CH;                                // r*w g*w b*w w
  float* recD = reinterpret_cast<float*>(recB + CH);       // dd*w (WANT_DISP)
  int* next = reinterpret_cast<int*>(recD + (WANT_DISP ? CH : 0));
  int* head_all = next + CH;          // two bin tables, used alternately
  int* rowr = head_all + 2 * nbins;   // [0] first candidate row, [1] last

  float m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = a.M[16 * b + k];
  const float s = d.trg_downsampling, max_disp = d.max_disp,
              zscale = d.zbuf_scale;
  const bool has_mask = d.flags & LSI_HAS_MASK;
  const bool compose = d.flags & LSI_COMPOSE;
  const float xmax = (float)Wt - 1.0f, ymax = (float)Ht - 1.0f;
  // acceptance window of the top-left cell (x0, y0), as floats
  const float ay_lo = (float)(ty0 - 1), ay_hi = (float)(ty0 + TH - 1);
  const float ax_lo = (float)(tx0 - 1), ax_hi = (float)(tx0 + TW - 1);

  for (int i = tid; i < 2 * nbins; i += TT) head_all[i] = -1;
  if (tid == 0) { rowr[0] = H; rowr[1] = -1; }
  __syncthreads();

  // ---- source rows that can reach the tile -------------------------------
  {
    float2 dr = make_float2(__builtin_inff(), -__builtin_inff());
    for (int l = 0; l < d.L; ++l)  // this kernel bounds all layers together
      for (int k = 0; k < LSI_RANGE_SLICES; ++k) {
        const float2 r = range[((size_t)l * d.B + b) * LSI_RANGE_SLICES + k];
        dr.x = fminf(dr.x, r.x);
        dr.y = fmaxf(dr.y, r.y);
      }
    int lo = H, hi = -1;
    for (int y = tid; y < H; y += TT) {
      const float py = (float)y + 0.5f;
      float vmin = __builtin_inff(), vmax = -__builtin_inff();
      bool wild = !(dr.x <= dr.y);  // no finite disparity seen: nothing to bound
      float nsign = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float px = (k & 1) ? (float)W - 0.5f : 0.5f;
        const float dv = (k & 2) ? dr.y : dr.x;
        const float q1 = mrow(m, 1, px, py, dv);
        const float n = mrow(m, 2, px, py, dv);
        const float v = div_rn(q1, safe_den(n)) * s - 0.5f;
        // Y is linear-fractional in x and in d: monotone along both as long as
        // the denominator keeps its sign on the row; else no bound holds
        if (!finite_f(v) || !finite_f(n) || n == 0.0f) wild = true;
        if (k == 0) nsign = n;
        else if ((n > 0.0f) != (nsign > 0.0f)) wild = true;
        vmin = fminf(vmin, v);
        vmax = fmaxf(vmax, v);
      }
      // one cell of slack on either side for the rounding of the corner values
      const bool cand = wild || (floorf(vmax) + 1.0f >= ay_lo &&
                                 floorf(vmin) - 1.0f <= ay_hi);
      if (cand) { lo = min(lo, y); hi = max(hi, y); }
    }
    if (hi >= 0) { atomicMin(&rowr[0], lo); atomicMax(&rowr[1], hi); }
  }
  __syncthreads();
  const int ys0 = rowr[0], ys1 = rowr[1];
  // pixel indices fit in 31 bits (H*W < 2^24); row = index / W by reciprocal
  const int px_begin = ys0 * W;
  const int px_end = (ys1 >= ys0) ? (ys1 + 1) * W : px_begin;
  const float rcp_w = 1.0f / (float)W;
  auto row_of = [&](int i) {  // i / W for 0 <= i < 2^24
    int q = (int)((float)i * rcp_w);
    const int r = i - q * W;
    q += (r >= W) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
  };

  const int CPT = c.cpt;  // thread t owns cells t + k * 1024 (row-major in tile)
  const float bg = d.bg_wt;
  const size_t P = (size_t)Ht * Wt;
  const bool shared_canvas = compose && !WANT_DISP;  // as the ATOMIC epilogue
  // compose totals per owned cell
  float T0[MAXCPT], T1[MAXCPT], T2[MAXCPT], TWs[MAXCPT], Tdmax[MAXCPT];
#pragma unroll
  for (int k = 0; k < MAXCPT; ++k) {
    T0[k] = 0.f; T1[k] = 0.f; T2[k] = 0.f; TWs[k] = 0.f; Tdmax[k] = 0.f;
  }
  int parity = 0;  // which bin table the current chunk fills
#if LSI_STREAM_HOOKS
  long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define TILE_STAMP(i) do { const long long tn = __builtin_readcyclecounter(); tacc[i] += tn - tprev; tprev = tn; } while (0)
#else
#define TILE_STAMP(i)
#endif

  for (int l = 0; l < d.L; ++l) {
    float a0[MAXCPT], a1[MAXCPT], a2[MAXCPT], aw[MAXCPT], ad[MAXCPT];  // layer
#pragma unroll
    for (int k = 0; k < MAXCPT; ++k) {
      a0[k] = 0.f; a1[k] = 0.f; a2[k] = 0.f; aw[k] = 0.f; ad[k] = 0.f;
    }
    const float* dbase = a.disp + (long)l * d.disp_sl + (long)b * d.disp_sb;
    const float* tbase = a.tex + (long)l * d.tex_sl + (long)b * d.tex_sb;
    const float* mbase =
        has_mask ? a.mask + (long)l * d.mask_sl + (long)b * d.mask_sb : nullptr;
    // the next chunk's pixels (disparity, colour, mask) are in flight while
    // this one is processed: no dependent global load on the critical path
    struct PxIn { float dv, t0, t1, t2, mk; };
    auto load_px = [&](int base, PxIn (&in)[CH / TT]) {
#pragma unroll
      for (int h = 0; h < CH / TT; ++h) {
        const int i = base + tid + h * TT;
        in[h].dv = 0.0f; in[h].t0 = 0.0f; in[h].t1 = 0.0f; in[h].t2 = 0.0f;
        in[h].mk = 1.0f;
        if (i < px_end) {
          const int y = row_of(i), x = i - y * W;
          in[h].dv = dbase[(long)y * d.disp_sy + (long)x * d.disp_sx];
          const float* tp = tbase + (long)y * d.tex_sy + (long)x * d.tex_sx;
          in[h].t0 = tp[0]; in[h].t1 = tp[d.tex_sc]; in[h].t2 = tp[2 * d.tex_sc];
          if (has_mask)
            in[h].mk = mbase[(long)y * d.mask_sy + (long)x * d.mask_sx];
        }
      }
    };
    PxIn in_next[CH / TT];
    load_px(px_begin, in_next);
    for (int base = px_begin; base < px_end; base += CH) {
      PxIn in_cur[CH / TT];
#pragma unroll
      for (int h = 0; h < CH / TT; ++h) in_cur[h] = in_next[h];
      if (base + CH < px_end) load_px(base + CH, in_next);
      int* head = head_all + parity * nbins;
      // ---- A: project, bin ------------------------------------------------
#pragma unroll
      for (int h = 0; h < CH / TT; ++h) {
        const int ri = tid + h * TT;
        const int i = base + ri;
        if (i >= px_end) continue;
        const int y = row_of(i), x = i - y * W;
        const float dv = in_cur[h].dv;
        const float px = (float)x + 0.5f, py = (float)y + 0.5f;
        // rows first: most rejected pixels miss the tile's rows
        // (non-finite X / Y fail the comparisons: dropped, like every path)
        const float q1 = mrow(m, 1, px, py, dv);
        const float nden = safe_den(mrow(m, 2, px, py, dv));
        const float Y = div_rn(q1, nden) * s - 0.5f;
        const float y0 = floorf(Y);
        if (!(y0 >= ay_lo && y0 <= ay_hi)) continue;
        const float q0 = mrow(m, 0, px, py, dv);
        const float X = div_rn(q0, nden) * s - 0.5f;
        const float x0 = floorf(X);
        if (!(x0 >= ax_lo && x0 <= ax_hi)) continue;
#ifdef LSI_TILE_EXPERIMENT_REJECT_ALL
        if (y0 > -1.0e30f) continue;
#endif
        const float q3 = mrow(m, 3, px, py, dv);
        const float dd = div_rn(q3, nden);
        const float pw =
            zbuffer_weight(div_rn(dd, max_disp), zscale) * in_cur[h].mk;
        if (pw == 0.0f) continue;  // contributes exactly +0 everywhere
        const Axis ax = splat_axis(X, xmax);
        const Axis ay = splat_axis(Y, ymax);
        // the four corner weights, border masks and 1e-3 clamp applied
        // (sampling.py:193-222): tl, tr, bl, br
        recA[ri] = make_float4(clamp_small(ax.w0 * ay.w0), clamp_small(ax.w1 * ay.w0),
                               clamp_small(ax.w0 * ay.w1), clamp_small(ax.w1 * ay.w1));
        recB[ri] = make_float4(in_cur[h].t0 * pw, in_cur[h].t1 * pw,
                               in_cur[h].t2 * pw, pw);
        if (WANT_DISP) recD[ri] = dd * pw;
        const int bin = (int)(y0 - ay_lo) * (TW + 1) + (int)(x0 - ax_lo);
        next[ri] = atomicExch(&head[bin], ri);
      }
      TILE_STAMP(0);
      __syncthreads();
      TILE_STAMP(1);
      // ---- B: every cell gathers its four bins; the other bin table (read in
      // the previous chunk) is cleared for the next one meanwhile
      {
        int* other = head_all + (parity ^ 1) * nbins;
        for (int i = tid; i < nbins; i += TT) other[i] = -1;
      }
      // The up-to-16 lists of a thread's cells are walked together: one
      // pointer-chasing step of each per round, so their LDS latencies overlap.
      {
        int jl[4 * MAXCPT];
#pragma unroll
        for (int q = 0; q < MAXCPT; ++q) {
          const int cell = tid + q * TT;
          const int cy = cell >> c.tw_log2, cx = cell & (TW - 1);
          const bool own = q < CPT && ty0 + cy < Ht && tx0 + cx < Wt;
          const int bin00 = (cy + 1) * (TW + 1) + (cx + 1);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            jl[4 * q + k] =
                own ? head[bin00 - (k >> 1) * (TW + 1) - (k & 1)] : -1;
        }
        // a chunk reaches only a few rows of the tile: most of a wave's cell
        // groups have nothing to gather (wave-uniform skip)
#ifndef LSI_TILE_EXPERIMENT_SKIP_GATHER
#pragma unroll
        for (int q = 0; q < MAXCPT; ++q) {
          while (__ballot((jl[4 * q] & jl[4 * q + 1] & jl[4 * q + 2] &
                           jl[4 * q + 3]) >= 0) != 0ull) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int j = jl[4 * q + k];
              if (j < 0) continue;
              // bin (Y - dy, X - dx) holds pixels whose corner (dy, dx) is
              // this cell: weight component k = 2 dy + dx of the record
              const float4 w4 = recA[j];
              const float4 v4 = recB[j];
              jl[4 * q + k] = next[j];
              const float cw = k == 0 ? w4.x : (k == 1 ? w4.y : (k == 2 ? w4.z : w4.w));
              a0[q] += v4.x * cw; a1[q] += v4.y * cw; a2[q] += v4.z * cw;
              aw[q] += v4.w * cw;
              if (WANT_DISP) ad[q] += recD[j] * cw;
            }
          }
        }
#endif
      }
      TILE_STAMP(2);
      __syncthreads();  // records and `next` are rewritten by the next chunk
      TILE_STAMP(3);
      parity ^= 1;
    }
    // ---- layer done: per-layer outputs / compose -----------------------------
#pragma unroll
    for (int q = 0; q < MAXCPT; ++q) {
      if (q >= CPT) break;
      const int cell = tid + q * TT;
      const int cy = cell >> c.tw_log2, cx = cell & (TW - 1);
      const int gy = ty0 + cy, gx = tx0 + cx;
      if (gy >= Ht || gx >= Wt) continue;
      const size_t op = (size_t)gy * Wt + gx;
      if (shared_canvas) {
        T0[q] += a0[q]; T1[q] += a1[q]; T2[q] += a2[q]; TWs[q] += aw[q];
      } else {
        const float l0 = bg + a0[q], l1 = bg + a1[q], l2 = bg + a2[q],
                    lw = bg + aw[q];
        const float dl = WANT_DISP ? div_rn(ad[q], safe_den(lw)) : 0.0f;
        if (compose) {
          if (l == 0) {
            T0[q] = l0; T1[q] = l1; T2[q] = l2; TWs[q] = lw; Tdmax[q] = dl;
          } else {
            T0[q] += l0; T1[q] += l1; T2[q] += l2; TWs[q] += lw;
            Tdmax[q] = fmaxf(Tdmax[q], dl);
          }
        } else {
          const size_t o = ((size_t)l * d.B + b) * P + op;
          const float wd = safe_den(lw);
          a.out_img[3 * o + 0] = div_rn(l0, wd);
          a.out_img[3 * o + 1] = div_rn(l1, wd);
          a.out_img[3 * o + 2] = div_rn(l2, wd);
          a.out_wts[o] = lw;
          if (WANT_DISP) a.out_disp[o] = dl;
        }
      }
    }
  }
#if LSI_STREAM_HOOKS
  if ((d.reserved & 4) && tid == 0) {
    long long* o = reinterpret_cast<long long*>(
                       const_cast<float2*>(range) + (size_t)d.B * d.L * LSI_RANGE_SLICES) +
                   ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3];
  }
#endif
  if (compose) {
#pragma unroll
    for (int q = 0; q < MAXCPT; ++q) {
      if (q >= CPT) break;
      const int cell = tid + q * TT;
      const int cy = cell >> c.tw_log2, cx = cell & (TW - 1);
      const int gy = ty0 + cy, gx = tx0 + cx;
      if (gy >= Ht || gx >= Wt) continue;
      const size_t o = (size_t)b * P + (size_t)gy * Wt + gx;
      if (shared_canvas) {
        const float lbg = (float)d.L * bg;
        const float w = TWs[q] + lbg;
        const float wd = safe_den(w);
        a.out_img[3 * o + 0] = div_rn(T0[q] + lbg, wd);
        a.out_img[3 * o + 1] = div_rn(T1[q] + lbg, wd);
        a.out_img[3 * o + 2] = div_rn(T2[q] + lbg, wd);
        a.out_wts[o] = w;
      } else {
        const float wd = safe_den(TWs[q]);
        a.out_img[3 * o + 0] = div_rn(T0[q], wd);
        a.out_img[3 * o + 1] = div_rn(T1[q], wd);
        a.out_img[3 * o + 2] = div_rn(T2[q], wd);
        a.out_wts[o] = TWs[q];
        if (WANT_DISP) a.out_disp[o] = Tdmax[q];
      }
    }
  }
}


}  // namespace

size_t lsi_tile_workspace_bytes(const LsiSplatDesc* d) {
  return (size_t)d->B * d->L * LSI_RANGE_SLICES * sizeof(float2);
}

int lsi_tile_launch(const SplatArgs& a, hipStream_t stream) {
  const LsiSplatDesc* d = &a.d;
  if (!a.canvas) return LSI_ENULL;
  if (a.ws_bytes < lsi_tile_workspace_bytes(d)) return LSI_EWORKSPACE;
  if ((reinterpret_cast<uintptr_t>(a.canvas) & 7) != 0) return LSI_EINVAL;
  float2* range = reinterpret_cast<float2*>(a.canvas);
  const int rvec4 = d->W % 4 == 0 && d->disp_sx == 1 && d->disp_sy % 4 == 0 &&
                    d->disp_sb % 4 == 0 && d->disp_sl % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.disp) & 15) == 0;
  hipLaunchKernelGGL(disp_range_kernel, dim3(d->B * d->L, LSI_RANGE_SLICES),
                     dim3(256), 0, stream, a, range, rvec4);
  const bool want_disp = (d->flags & LSI_WANT_DISP) != 0;
  // the sweep kernel; the gather kernel below remains for more than
  // LSI_SWEEP_MAXL composed layers (reserved bit 8 forces it, for A/B runs)
  if (!(d->reserved & 256) &&
      (!(d->flags & LSI_COMPOSE) || d->L <= LSI_SWEEP_MAXL))
    return lsi_sweep_launch(a, range, stream);
  TileCfg c;
  c.tw_log2 = d->Wt <= 32 ? 5 : (d->Wt <= 64 ? 6 : 7);
  const int TW = 1 << c.tw_log2;
  // Taller tiles re-project fewer source rows per output row ((TH + reach) / TH)
  // but leave fewer workgroups: the tallest tile that still gives every CU a
  // workgroup (tune_rows overrides: tile height in cells).
  c.cpt = 1;
  if (d->tune_rows > 0) {
    c.cpt = d->tune_rows * TW / TT;
    if (c.cpt < 1) c.cpt = 1;
    if (c.cpt > MAXCPT) c.cpt = MAXCPT;
    if (c.cpt == 3) c.cpt = 2;
  } else {
    for (int k = 2; k <= MAXCPT; k *= 2) {
      const int th = k * TT / TW;
      const long nwg =
          (long)((d->Ht + th - 1) / th) * ((d->Wt + TW - 1) / TW) * d->B;
      if (nwg < 256 || th > 2 * d->Ht) break;
      c.cpt = k;
    }
  }
  c.th = c.cpt * TT / TW;
  c.tiles_x = (d->Wt + TW - 1) / TW;
  const int tiles_y = (d->Ht + c.th - 1) / c.th;
  const size_t lds = (size_t)CH * 32 + (want_disp ? (size_t)CH * 4 : 0) +
                     (size_t)CH * 4 + (size_t)2 * (c.th + 1) * (TW + 1) * 4 + 16;
  const void* fn = want_disp ? (const void*)splat_tile_kernel<true>
                             : (const void*)splat_tile_kernel<false>;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return LSI_ELAUNCH;
  const float2* crange = range;
  void* kargs[3] = {const_cast<SplatArgs*>(&a), &c, &crange};
  if (hipLaunchKernel(fn, dim3(c.tiles_x * tiles_y, d->B), dim3(TT), kargs, lds,
                      stream) != hipSuccess)
    return LSI_ELAUNCH;
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
