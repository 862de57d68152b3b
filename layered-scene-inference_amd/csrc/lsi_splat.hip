// LDI forward-splat renderer for MI355X (gfx950): projection + soft z-buffer
// weight + 4-corner bilinear splat + normalise/compose, ldi.py:71-182 with
// sampling.py:171-254 and helpers.py:82-85,116-137,180-193 fused in.
//
// Two of the four kernel families behind lsi_splat_fwd (include/lsi_hip.h; the
// others: lsi_splat_stream.hip, lsi_splat_tile.hip + lsi_splat_sweep.hip):
//
//  LSI_PATH_ATOMIC   any projection matrix.  One thread per source pixel,
//                    fp32 global atomics (global_atomic_add_f32) into
//                    per-layer canvases held in the caller's workspace, then a
//                    per-target-pixel epilogue kernel.
//
//  LSI_PATH_ROWBAND  projection matrices whose target ROW does not depend on
//                    disparity (M[1][3] == M[2][3] == 0: rectified stereo and
//                    every pure-x translation, any rotation/intrinsics).  One
//                    workgroup owns a band of target rows across the full
//                    width, keeps the band's canvases in LDS (ds_add_f32),
//                    scans exactly the source rows that can reach the band
//                    (all layers), and writes every output pixel once,
//                    already normalised / composed.  No global atomics, no
//                    workspace, one launch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"

#pragma clang fp contract(off)

using namespace lsi;

#include "lsi_splat_internal.h"

namespace {

// ---------------------------------------------------------------------------
// LSI_PATH_ATOMIC
// ---------------------------------------------------------------------------
// grid (ceil(H*W/256), B, L); canvas [ncanv][B][nch][P] (planar: the lanes of a
// wave hit neighbouring dwords of one channel plane, which the L2 atomic units
// serve ~4x faster than a 4-channel interleaved canvas -- tools/microbench.hip)
// zero-initialised.  Adjacent lanes that map to the same target cell (the
// common case at trg_downsampling 0.5) are merged with one DPP exchange before
// the atomics, halving their number.
__global__ __launch_bounds__(256) void splat_atomic_kernel(SplatArgs a) {
  const LsiSplatDesc& d = a.d;
  const int b = blockIdx.y, l = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool inr = i < d.H * d.W;
  const int ii = inr ? i : 0;
  const int y = ii / d.W, x = ii - y * d.W;
  const float* __restrict__ m = a.M + 16 * b;
  const float dv = a.disp[l * d.disp_sl + b * d.disp_sb + y * d.disp_sy +
                          x * d.disp_sx];
  const float mk = (d.flags & LSI_HAS_MASK)
                       ? a.mask[l * d.mask_sl + b * d.mask_sb + y * d.mask_sy +
                                x * d.mask_sx]
                       : 1.0f;
  Proj p;
  project_px(m, (float)x + 0.5f, (float)y + 0.5f, dv, mk, d.trg_downsampling,
             d.max_disp, d.zbuf_scale, d.Ht, d.Wt, p);
  // a pixel with zero weight contributes exactly +0 everywhere
  const bool active = inr && p.ok && (p.pw != 0.0f);
  const float* tp =
      a.tex + l * d.tex_sl + b * d.tex_sb + y * d.tex_sy + x * d.tex_sx;
  const float pw = active ? p.pw : 0.0f;
  float ch[5];
  ch[0] = tp[0] * pw; ch[1] = tp[d.tex_sc] * pw; ch[2] = tp[2 * d.tex_sc] * pw;
  ch[3] = pw; ch[4] = active ? p.dd * pw : 0.0f;  // (dd may be NaN when dropped)
  const int nch = a.nch;
  // partner = the other lane of the pair (2m, 2m+1): quad_perm [1,0,3,2].
  // The pair is merged only when all four cells agree: tl and br are compared
  // exactly (two exchanges; tr and bl follow from them for valid footprints:
  // both lanes' cells are the clipped corners of the same two columns / rows).
  const int k0 = active ? p.idx[0] : -1 - (int)threadIdx.x;
  const int k3 = active ? p.idx[3] : -1 - (int)threadIdx.x;
  const int pk0 = __builtin_amdgcn_update_dpp(0, k0, 0xB1, 0xf, 0xf, true);
  const int pk3 = __builtin_amdgcn_update_dpp(0, k3, 0xB1, 0xf, 0xf, true);
  const int k1 = active ? p.idx[1] : -1, k2 = active ? p.idx[2] : -1;
  const int pk1 = __builtin_amdgcn_update_dpp(0, k1, 0xB1, 0xf, 0xf, true);
  const int pk2 = __builtin_amdgcn_update_dpp(0, k2, 0xB1, 0xf, 0xf, true);
  const bool same = (k0 == pk0) && (k3 == pk3) && (k1 == pk1) && (k2 == pk2);
  const bool odd = threadIdx.x & 1;
  const size_t P = (size_t)d.Ht * d.Wt;
  const int lc = a.shared ? 0 : l;
  float* cv = a.canvas + ((size_t)lc * d.B + b) * P * nch;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float wk = active ? p.w[k] : 0.0f;
    float* t = cv + p.idx[k];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      if (c >= nch) break;
      float v = ch[c] * wk;
      const float pv = __int_as_float(__builtin_amdgcn_update_dpp(
          0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
      if (same) v = odd ? 0.0f : v + pv;   // even lane carries the pair
      if (v != 0.0f) atomic_add_f32(t + (size_t)c * P, v);
    }
  }
}

// One thread per (b, target pixel): background init, per-layer disparity
// normalisation, compose, final normalisation (ldi.py:122-125,157-182).
__global__ __launch_bounds__(256) void splat_epilogue_kernel(SplatArgs a) {
  const LsiSplatDesc& d = a.d;
  const size_t P = (size_t)d.Ht * d.Wt;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= P * d.B) return;
  const int b = (int)(gid / P);
  const size_t p = gid - (size_t)b * P;
  const float bg = d.bg_wt;
  const bool compose = d.flags & LSI_COMPOSE;
  const bool want_disp = d.flags & LSI_WANT_DISP;
  if (a.shared) {  // compose, no disparity: one shared canvas, L x bg
    const float* c = a.canvas + (size_t)b * P * a.nch + p;  // planar
    const float lbg = (float)d.L * bg;
    const float w = c[3 * P] + lbg;
    const float wd = safe_den(w);
    const size_t o = (size_t)b * P + p;
    a.out_img[3 * o + 0] = div_rn(c[0] + lbg, wd);
    a.out_img[3 * o + 1] = div_rn(c[P] + lbg, wd);
    a.out_img[3 * o + 2] = div_rn(c[2 * P] + lbg, wd);
    a.out_wts[o] = w;
    return;
  }
  float A0 = 0.f, A1 = 0.f, A2 = 0.f, W = 0.f, dmax = 0.f;
  for (int l = 0; l < d.L; ++l) {
    const float* c = a.canvas + ((size_t)l * d.B + b) * P * a.nch + p;  // planar
    const float a0 = bg + c[0], a1 = bg + c[P], a2 = bg + c[2 * P],
                w = bg + c[3 * P];
    const float dl = want_disp ? div_rn(c[4 * P], safe_den(w)) : 0.0f;
    if (compose) {
      if (l == 0) { A0 = a0; A1 = a1; A2 = a2; W = w; dmax = dl; }
      else { A0 += a0; A1 += a1; A2 += a2; W += w; dmax = fmaxf(dmax, dl); }
    } else {
      const size_t o = ((size_t)l * d.B + b) * P + p;
      const float wd = safe_den(w);
      a.out_img[3 * o + 0] = div_rn(a0, wd);
      a.out_img[3 * o + 1] = div_rn(a1, wd);
      a.out_img[3 * o + 2] = div_rn(a2, wd);
      a.out_wts[o] = w;
      if (want_disp) a.out_disp[o] = dl;
    }
  }
  if (compose) {
    const size_t o = (size_t)b * P + p;
    const float wd = safe_den(W);
    a.out_img[3 * o + 0] = div_rn(A0, wd);
    a.out_img[3 * o + 1] = div_rn(A1, wd);
    a.out_img[3 * o + 2] = div_rn(A2, wd);
    a.out_wts[o] = W;
    if (want_disp) a.out_disp[o] = dmax;
  }
}

// ---------------------------------------------------------------------------
// LSI_PATH_ROWBAND
// ---------------------------------------------------------------------------
// Workgroup (band, b): target rows [band*R, band*R + R) x all Wt columns.
// LDS layout: tile[R][Wt][NCH] floats (+ fold[R][Wt][5] when disparity is
// composed over layers) + 2 ints (source row range).
//
// Source rows that can reach the band: the target row of a source pixel is
// floor(Y), floor(Y)+1 with Y = (q1/n')*s - 0.5, independent of disparity on
// this path, and for a fixed source row monotone in x (n' > 0 is part of the
// path's precondition), so the extreme target rows of a source row are reached
// at its two end pixels.  Every thread evaluates those two end pixels for a few
// source rows with the SAME fp32 sequence the splat uses, widened by a safety
// margin, and the workgroup scans the bounding range [y_lo, y_hi].
template <int NCH>
__global__ __launch_bounds__(1024) void splat_rowband_kernel(SplatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const LsiSplatDesc& d = a.d;
  const int R = a.band_rows;
  const int band = blockIdx.x, b = blockIdx.y;
  const int T = blockDim.x, tid = threadIdx.x;
  const int Wt = d.Wt, Ht = d.Ht;
  const int row0 = band * R;
  const int rows = min(R, Ht - row0);
  const int tile_elems = R * Wt * NCH;
  const bool compose = d.flags & LSI_COMPOSE;
  const bool want_disp = (NCH == 5);
  float* tile = smem;
  float* fold = smem + tile_elems;                      // only compose+disp
  int* yrange = reinterpret_cast<int*>(
      smem + tile_elems + ((compose && want_disp) ? R * Wt * 5 : 0));

  const float* __restrict__ m = a.M + 16 * b;
  const float s = d.trg_downsampling;

  // ---- source row range ------------------------------------------------
  if (tid == 0) { yrange[0] = d.H; yrange[1] = -1; }
  __syncthreads();
  {
    int lo = d.H, hi = -1;
    const float xe[2] = {0.5f, (float)d.W - 0.5f};
    for (int y = tid; y < d.H; y += T) {
      const float py = (float)y + 0.5f;
      float ymin = __builtin_inff(), ymax = -__builtin_inff();
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float q1 = mrow(m, 1, xe[e], py, 0.0f);
        const float n = safe_den(mrow(m, 2, xe[e], py, 0.0f));
        const float Y = div_rn(q1, n) * s - 0.5f;
        ymin = fminf(ymin, Y);
        ymax = fmaxf(ymax, Y);
      }
      const float mar = 0.01f + 1e-5f * fmaxf(fabsf(ymin), fabsf(ymax));
      // target rows touched: floor(ymin - mar) .. floor(ymax + mar) + 1
      const float tlo = floorf(ymin - mar), thi = floorf(ymax + mar) + 1.0f;
      const bool hit = (thi >= (float)row0) && (tlo <= (float)(row0 + rows - 1));
      // NaN (cannot happen on this path) would compare false => row skipped.
      if (hit) { lo = min(lo, y); hi = max(hi, y); }
    }
    if (hi >= 0) {
      atomicMin(&yrange[0], lo);
      atomicMax(&yrange[1], hi);
    }
  }
  __syncthreads();
  const int y_lo = yrange[0], y_hi = yrange[1];

  const float bg = d.bg_wt;
  const size_t P = (size_t)Ht * Wt;
  const int band_px = rows * Wt;
  const int nlayer_pass = compose && !want_disp ? 1 : d.L;  // tile resets

  for (int pass = 0; pass < nlayer_pass; ++pass) {
    // zero the tile
    for (int i = tid; i < tile_elems; i += T) tile[i] = 0.0f;
    if (compose && want_disp && pass == 0)
      for (int i = tid; i < R * Wt * 5; i += T) fold[i] = 0.0f;
    __syncthreads();

    const int l_begin = (compose && !want_disp) ? 0 : pass;
    const int l_end = (compose && !want_disp) ? d.L : pass + 1;
    if (y_hi >= y_lo) {
      const int nrow = y_hi - y_lo + 1;
      for (int l = l_begin; l < l_end; ++l) {
        const float* dl = a.disp + l * d.disp_sl + b * d.disp_sb;
        const float* tl = a.tex + l * d.tex_sl + b * d.tex_sb;
        const float* ml = (d.flags & LSI_HAS_MASK)
                              ? a.mask + l * d.mask_sl + b * d.mask_sb
                              : nullptr;
        // thread -> (row, column) by flat index over the band's source rows
        for (int j = tid; j < nrow * d.W; j += T) {
          const int yr = j / d.W;
          const int x = j - yr * d.W;
          const int y = y_lo + yr;
          const float dv = dl[y * d.disp_sy + x * d.disp_sx];
          const float mk = ml ? ml[y * d.mask_sy + x * d.mask_sx] : 1.0f;
          Proj p;
          project_px(m, (float)x + 0.5f, (float)y + 0.5f, dv, mk, s, d.max_disp,
                     d.zbuf_scale, Ht, Wt, p);
          if (!p.ok || p.pw == 0.0f) continue;
          const float* tp = tl + y * d.tex_sy + x * d.tex_sx;
          const float r = tp[0] * p.pw, g = tp[d.tex_sc] * p.pw,
                      bl = tp[2 * d.tex_sc] * p.pw;
          const float dw = p.dd * p.pw;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float wk = p.w[k];
            const int rel = p.idx[k] - row0 * Wt;  // flat offset inside band
            if (wk == 0.0f || rel < 0 || rel >= band_px) continue;
            float* t = tile + rel * NCH;
            atomic_add_f32(t + 0, r * wk);
            atomic_add_f32(t + 1, g * wk);
            atomic_add_f32(t + 2, bl * wk);
            atomic_add_f32(t + 3, p.pw * wk);
            if (NCH == 5) atomic_add_f32(t + 4, dw * wk);
          }
        }
      }
    }
    __syncthreads();

    // ---- epilogue for this pass ---------------------------------------
    if (compose && !want_disp) {
      const float lbg = (float)d.L * bg;
      for (int i = tid; i < band_px; i += T) {
        const float* c = tile + i * NCH;
        const float w = c[3] + lbg;
        const float wd = safe_den(w);
        const size_t o = (size_t)b * P + (size_t)row0 * Wt + i;
        a.out_img[3 * o + 0] = div_rn(c[0] + lbg, wd);
        a.out_img[3 * o + 1] = div_rn(c[1] + lbg, wd);
        a.out_img[3 * o + 2] = div_rn(c[2] + lbg, wd);
        a.out_wts[o] = w;
      }
    } else if (!compose) {
      const int l = pass;
      for (int i = tid; i < band_px; i += T) {
        const float* c = tile + i * NCH;
        const float w = bg + c[3];
        const float wd = safe_den(w);
        const size_t o = ((size_t)l * d.B + b) * P + (size_t)row0 * Wt + i;
        a.out_img[3 * o + 0] = div_rn(bg + c[0], wd);
        a.out_img[3 * o + 1] = div_rn(bg + c[1], wd);
        a.out_img[3 * o + 2] = div_rn(bg + c[2], wd);
        a.out_wts[o] = w;
        if (NCH == 5) a.out_disp[o] = div_rn(c[4], wd);
      }
    } else {  // compose with disparity: fold this layer, emit after the last
      const bool last = (pass == d.L - 1);
      for (int i = tid; i < band_px; i += T) {
        const float* c = tile + i * NCH;
        float* f = fold + i * 5;
        const float a0 = bg + c[0], a1 = bg + c[1], a2 = bg + c[2],
                    w = bg + c[3];
        const float dl = div_rn(c[NCH - 1], safe_den(w));
        float A0, A1, A2, W, dm;
        if (pass == 0) { A0 = a0; A1 = a1; A2 = a2; W = w; dm = dl; }
        else {
          A0 = f[0] + a0; A1 = f[1] + a1; A2 = f[2] + a2; W = f[3] + w;
          dm = fmaxf(f[4], dl);
        }
        if (!last) { f[0] = A0; f[1] = A1; f[2] = A2; f[3] = W; f[4] = dm; }
        else {
          const size_t o = (size_t)b * P + (size_t)row0 * Wt + i;
          const float wd = safe_den(W);
          a.out_img[3 * o + 0] = div_rn(A0, wd);
          a.out_img[3 * o + 1] = div_rn(A1, wd);
          a.out_img[3 * o + 2] = div_rn(A2, wd);
          a.out_wts[o] = W;
          a.out_disp[o] = dm;
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Projection debug view (lsi_project_indices)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void project_indices_kernel(
    SplatArgs a, int32_t* __restrict__ idx4, float* __restrict__ upd4) {
  const LsiSplatDesc& d = a.d;
  const int b = blockIdx.y, l = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.H * d.W) return;
  const int y = i / d.W, x = i - y * d.W;
  const float dv = a.disp[l * d.disp_sl + b * d.disp_sb + y * d.disp_sy +
                          x * d.disp_sx];
  const float mk = (d.flags & LSI_HAS_MASK)
                       ? a.mask[l * d.mask_sl + b * d.mask_sb + y * d.mask_sy +
                                x * d.mask_sx]
                       : 1.0f;
  Proj p;
  project_px(a.M + 16 * b, (float)x + 0.5f, (float)y + 0.5f, dv, mk,
             d.trg_downsampling, d.max_disp, d.zbuf_scale, d.Ht, d.Wt, p);
  const size_t o = (((size_t)l * d.B + b) * ((size_t)d.H * d.W) + i) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    idx4[o + k] = p.idx[k];
    upd4[o + k] = p.pw * p.w[k];
  }
}

// ---------------------------------------------------------------------------
// Backward (gather): see lsi_splat_bwd in include/lsi_hip.h.
// ---------------------------------------------------------------------------
// Pre-pass, one thread per output pixel: gradient w.r.t. the un-normalised
// canvases A (3 ch) and W:   img = A / W',  wts = W,  W' = W + 1e-8[W == 0]
//   gA = g_img / W'          gW = g_wts - sum_c g_img_c * img_c / W'
__global__ __launch_bounds__(256) void splat_bwd_pre_kernel(
    size_t n, const float* __restrict__ out_img,
    const float* __restrict__ out_wts, const float* __restrict__ g_img,
    const float* __restrict__ g_wts, float4* __restrict__ G) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float wd = safe_den(out_wts[i]);
  const float inv = 1.0f / wd;
  const float g0 = g_img[3 * i], g1 = g_img[3 * i + 1], g2 = g_img[3 * i + 2];
  float gw = g_wts ? g_wts[i] : 0.0f;
  gw -= (g0 * out_img[3 * i] + g1 * out_img[3 * i + 1] +
         g2 * out_img[3 * i + 2]) * inv;
  G[i] = make_float4(g0 * inv, g1 * inv, g2 * inv, gw);
}

// grid (ceil(W / 256), ceil(H / BWD_ROWS), B * L): no integer division per thread.  SIMPLE_M
// (decided per batch element from M itself, uniform per block): normaliser == 1
// and target disparity == source disparity, i.e. M rows 2, 3 = (0,0,1,0),
// (0,0,0,1) -- every rectified pair.  Then u = q0 * s exactly (x / 1 == x), so
// the corner cells are the forward's bit for bit, and the projection and its
// gradient need no division at all.
struct BwdPx { float gt0, gt1, gt2, gm, gd; };

template <bool SIMPLE_M>
__device__ __forceinline__ BwdPx splat_bwd_core(
    const LsiSplatDesc& d, const float* __restrict__ m, int y, int x, float dv,
    float mk, float t0, float t1, float t2, const float4* __restrict__ Gb) {
  const float s = d.trg_downsampling;
  const float inv_md = div_rn(1.0f, d.max_disp);
  Proj p;
  if (SIMPLE_M) {
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    p.q0 = mrow(m, 0, px, py, dv);
    p.q1 = mrow(m, 1, px, py, dv);
    p.q3 = dv;
    p.nden = 1.0f;
    p.dd = dv;
    p.zw = zbuffer_weight(dv * inv_md, d.zbuf_scale);
    p.pw = p.zw * mk;
    const float X = p.q0 * s - 0.5f, Y = p.q1 * s - 0.5f;
    p.ok = finite_f(X) && finite_f(Y);
    p.ax = splat_axis(X, (float)d.Wt - 1.0f);
    p.ay = splat_axis(Y, (float)d.Ht - 1.0f);
    const float wt = (float)d.Wt;
    p.w[0] = clamp_small(p.ax.w0 * p.ay.w0);
    p.w[1] = clamp_small(p.ax.w1 * p.ay.w0);
    p.w[2] = clamp_small(p.ax.w0 * p.ay.w1);
    p.w[3] = clamp_small(p.ax.w1 * p.ay.w1);
    if (p.ok) {
      p.idx[0] = (int)(p.ax.c0s + p.ay.c0s * wt);
      p.idx[1] = (int)(p.ax.c1s + p.ay.c0s * wt);
      p.idx[2] = (int)(p.ax.c0s + p.ay.c1s * wt);
      p.idx[3] = (int)(p.ax.c1s + p.ay.c1s * wt);
    } else {
      p.idx[0] = p.idx[1] = p.idx[2] = p.idx[3] = 0;
      p.w[0] = p.w[1] = p.w[2] = p.w[3] = 0.0f;
    }
  } else {
    project_px(m, (float)x + 0.5f, (float)y + 0.5f, dv, mk, s, d.max_disp,
               d.zbuf_scale, d.Ht, d.Wt, p);
  }
  float gt0 = 0.f, gt1 = 0.f, gt2 = 0.f, gpw = 0.f;
  float gwk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    gwk[k] = 0.0f;
    if (!p.ok || p.w[k] == 0.0f) continue;  // clamped / invalid: zero grad
    const float4 gg = Gb[p.idx[k]];
    const float S = t0 * gg.x + t1 * gg.y + t2 * gg.z + gg.w;
    gt0 += p.w[k] * gg.x;
    gt1 += p.w[k] * gg.y;
    gt2 += p.w[k] * gg.z;
    gpw += p.w[k] * S;
    gwk[k] = p.pw * S;
  }
  BwdPx r;
  r.gt0 = gt0 * p.pw;
  r.gt1 = gt1 * p.pw;
  r.gt2 = gt2 * p.pw;
  r.gm = gpw * p.zw;

  // corner weights -> X, Y.  w_tl = wx0*wy0 etc.; d wx0/dX = -v0, d wx1/dX = +v1
  const float gX = -p.ax.v0 * (gwk[0] * p.ay.w0 + gwk[2] * p.ay.w1) +
                   p.ax.v1 * (gwk[1] * p.ay.w0 + gwk[3] * p.ay.w1);
  const float gY = -p.ay.v0 * (gwk[0] * p.ax.w0 + gwk[1] * p.ax.w1) +
                   p.ay.v1 * (gwk[2] * p.ax.w0 + gwk[3] * p.ax.w1);
  // pw = zw * mask ; zw = exp((clip(xn,0,1)-.5)*scale)*[xn>0], xn = D/max_disp
  const float xn = SIMPLE_M ? p.dd * inv_md : p.dd / d.max_disp;
  const float inr = (xn >= 0.0f && xn <= 1.0f) ? 1.0f : 0.0f;
  const float gD = SIMPLE_M
                       ? gpw * mk * p.zw * d.zbuf_scale * inr * inv_md
                       : gpw * mk * p.zw * d.zbuf_scale * inr / d.max_disp;
  // u = q0/n' * s, v = q1/n' * s, D = q3/n'
  float gd;
  if (SIMPLE_M) {  // n' == 1, M[2][3] == 0, M[3][3] == 1
    gd = (gX * s) * m[3] + (gY * s) * m[7] + gD;
  } else {
    const float inv_n = 1.0f / p.nden;
    const float gq0 = gX * s * inv_n, gq1 = gY * s * inv_n, gq3 = gD * inv_n;
    const float gn = -(gq0 * p.q0 + gq1 * p.q1 + gq3 * p.q3) * inv_n;
    gd = gq0 * m[3] + gq1 * m[7] + gn * m[11] + gq3 * m[15];
  }
  if (!p.ok) gd = 0.0f;
  r.gd = gd;
  return r;
}

// One thread = one column x of BWD_ROWS consecutive source rows.  A wave with a
// single pixel per lane is a chain of dependent memory round trips (arguments,
// inputs, the four gathers, stores) with nothing to overlap them; here the next
// row's inputs are in flight while the current row gathers and computes, and
// the per-wave setup is paid once per BWD_ROWS rows (one pixel per wave-lane:
// 294 us at config 3; this loop: see DESIGN.md 4.5).
constexpr int BWD_ROWS = 8;

struct BwdIn { float dv, mk, t0, t1, t2; };

template <bool SIMPLE_M>
__device__ __forceinline__ void splat_bwd_rows(
    const SplatArgs& a, const float* __restrict__ m, int b, int l, int y0, int x,
    const float4* __restrict__ G, float* __restrict__ g_tex,
    float* __restrict__ g_disp, float* __restrict__ g_mask) {
  const LsiSplatDesc& d = a.d;
  const bool has_mask = d.flags & LSI_HAS_MASK;
  const float* dbase = a.disp + l * d.disp_sl + b * d.disp_sb + x * d.disp_sx;
  const float* tbase = a.tex + l * d.tex_sl + b * d.tex_sb + x * d.tex_sx;
  const float* mbase =
      has_mask ? a.mask + l * d.mask_sl + b * d.mask_sb + x * d.mask_sx : nullptr;
  auto load = [&](int y) {  // (rows past the end re-read the last row: no branch)
    const int yc = min(y, d.H - 1);
    BwdIn in;
    in.dv = dbase[yc * d.disp_sy];
    const float* tp = tbase + yc * d.tex_sy;
    if (d.tex_sc == 1) {  // channels last: one 12-byte load per lane
      const float3 t3 = *reinterpret_cast<const float3*>(tp);
      in.t0 = t3.x; in.t1 = t3.y; in.t2 = t3.z;
    } else {
      in.t0 = tp[0]; in.t1 = tp[d.tex_sc]; in.t2 = tp[2 * d.tex_sc];
    }
    in.mk = has_mask ? mbase[yc * d.mask_sy] : 1.0f;
    return in;
  };
  const size_t P = (size_t)d.Ht * d.Wt;
  const int lo = (d.flags & LSI_COMPOSE) ? 0 : l;
  const float4* Gb = G + ((size_t)lo * d.B + b) * P;
  const size_t obase = ((size_t)l * d.B + b) * ((size_t)d.H * d.W) + x;
  BwdIn cur = load(y0);
#pragma unroll
  for (int r = 0; r < BWD_ROWS; ++r) {
    const int y = y0 + r;
    const BwdIn nxt = load(y + 1);
    if (y < d.H) {
      const BwdPx g = splat_bwd_core<SIMPLE_M>(d, m, y, x, cur.dv, cur.mk, cur.t0,
                                               cur.t1, cur.t2, Gb);
      const size_t o = obase + (size_t)y * d.W;
      // one 12-byte store per lane: a wave writes 768 contiguous bytes
      *reinterpret_cast<float3*>(g_tex + 3 * o) = make_float3(g.gt0, g.gt1, g.gt2);
      if (g_mask) g_mask[o] = g.gm;
      g_disp[o] = g.gd;
    }
    cur = nxt;
  }
}

__global__ __launch_bounds__(256) void splat_bwd_kernel(
    SplatArgs a, float inv_l, const float4* __restrict__ G,
    float* __restrict__ g_tex, float* __restrict__ g_disp,
    float* __restrict__ g_mask) {
  const LsiSplatDesc& d = a.d;
  const int y0 = blockIdx.y * BWD_ROWS;
  // blockIdx.z = b * L + l: the layers of a batch element run back to back and
  // find its gradient canvas in the L2
  int b = (int)((float)blockIdx.z * inv_l);
  int l = (int)blockIdx.z - b * d.L;
  if (l < 0) { --b; l += d.L; }
  if (l >= d.L) { ++b; l -= d.L; }
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= d.W) return;
  const float* __restrict__ m = a.M + 16 * b;
  const bool simple_m = m[8] == 0.0f && m[9] == 0.0f && m[10] == 1.0f &&
                        m[11] == 0.0f && m[12] == 0.0f && m[13] == 0.0f &&
                        m[14] == 0.0f && m[15] == 1.0f;
  if (simple_m)
    splat_bwd_rows<true>(a, m, b, l, y0, x, G, g_tex, g_disp, g_mask);
  else
    splat_bwd_rows<false>(a, m, b, l, y0, x, G, g_tex, g_disp, g_mask);
}

static void launch_bwd(const SplatArgs& a, const float4* G, float* g_tex,
                       float* g_disp, float* g_mask, hipStream_t stream) {
  const LsiSplatDesc* d = &a.d;
  hipLaunchKernelGGL(splat_bwd_kernel,
                     dim3((d->W + 255) / 256, (d->H + BWD_ROWS - 1) / BWD_ROWS,
                          d->L * d->B),
                     dim3(256), 0, stream, a, 1.0f / (float)d->L, G, g_tex, g_disp,
                     g_mask);
}

// Pre-pass of lsi_splat_bwd_both: layer l's canvas receives the gradient of
// its own (independent) output plus the composed output's, whose canvas is the
// sum of the layers' canvases (ldi.py:167-171).
__global__ __launch_bounds__(256) void splat_bwd_pre_both_kernel(
    size_t n1, int L, const float* __restrict__ img_i,
    const float* __restrict__ wts_i, const float* __restrict__ img_c,
    const float* __restrict__ wts_c, const float* __restrict__ g_img_i,
    const float* __restrict__ g_wts_i, const float* __restrict__ g_img_c,
    const float* __restrict__ g_wts_c, float4* __restrict__ G) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // (b, pixel)
  if (i >= n1) return;
  float4 gc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g_img_c) {
    const float inv = 1.0f / safe_den(wts_c[i]);
    const float g0 = g_img_c[3 * i], g1 = g_img_c[3 * i + 1], g2 = g_img_c[3 * i + 2];
    float gw = g_wts_c ? g_wts_c[i] : 0.0f;
    gw -= (g0 * img_c[3 * i] + g1 * img_c[3 * i + 1] + g2 * img_c[3 * i + 2]) * inv;
    gc = make_float4(g0 * inv, g1 * inv, g2 * inv, gw);
  }
  for (int l = 0; l < L; ++l) {
    const size_t j = (size_t)l * n1 + i;
    float4 g = gc;
    if (g_img_i) {
      const float inv = 1.0f / safe_den(wts_i[j]);
      const float g0 = g_img_i[3 * j], g1 = g_img_i[3 * j + 1], g2 = g_img_i[3 * j + 2];
      float gw = g_wts_i ? g_wts_i[j] : 0.0f;
      gw -= (g0 * img_i[3 * j] + g1 * img_i[3 * j + 1] + g2 * img_i[3 * j + 2]) * inv;
      g.x += g0 * inv; g.y += g1 * inv; g.z += g2 * inv; g.w += gw;
    }
    G[j] = g;
  }
}

// Composed outputs from the per-layer ones, for the kernel families that do
// not produce both in one sweep: canvas_l = img_l * W'_l (the normalisation
// undone; one rounding more than the fused sum), ldi.py:167-174.
__global__ __launch_bounds__(256) void compose_from_layers_kernel(
    size_t n1, int L, const float* __restrict__ img_i,
    const float* __restrict__ wts_i, float* __restrict__ img_c,
    float* __restrict__ wts_c) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n1) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, w = 0.f;
  for (int l = 0; l < L; ++l) {
    const size_t j = (size_t)l * n1 + i;
    const float wl = wts_i[j], wd = safe_den(wl);
    a0 += img_i[3 * j] * wd; a1 += img_i[3 * j + 1] * wd; a2 += img_i[3 * j + 2] * wd;
    w += wl;
  }
  const float wd = safe_den(w);
  img_c[3 * i] = div_rn(a0, wd); img_c[3 * i + 1] = div_rn(a1, wd);
  img_c[3 * i + 2] = div_rn(a2, wd);
  wts_c[i] = w;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int check_desc(const LsiSplatDesc* d) {
  if (!d) return LSI_ENULL;
  if (d->L <= 0 || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ht <= 0 ||
      d->Wt <= 0)
    return LSI_EINVAL;
  if (d->B > 65535 || d->L > 65535) return LSI_EINVAL;
  if ((int64_t)d->H * d->W >= (1 << 24) || (int64_t)d->Ht * d->Wt >= (1 << 24))
    return LSI_EINVAL;  // fp32 index arithmetic exact only below 2^24
  if (!(d->max_disp > 0.0f)) return LSI_EINVAL;
  // the device exp (lsi_common.h: exp_accurate) is valid for |argument| < ~80,
  // i.e. |zbuf_scale| / 2: beyond that the weights would overflow / flush
  // differently from the reference's exp
  if (!(fabsf(d->zbuf_scale) <= 160.0f)) return LSI_EINVAL;
  return LSI_OK;
}

// LSI_PACKED_RGBD is the caller's statement about its pointers: verified here
bool packed_ok(const LsiSplatDesc* d, const float* tex, const float* disp) {
  if (!(d->flags & LSI_PACKED_RGBD)) return true;
  return disp == tex + 3 && d->tex_sc == 1 && d->tex_sx == 4 && d->disp_sx == 4 &&
         d->tex_sl == d->disp_sl && d->tex_sb == d->disp_sb &&
         d->tex_sy == d->disp_sy && d->tex_sl % 4 == 0 && d->tex_sb % 4 == 0 &&
         d->tex_sy % 4 == 0 && (reinterpret_cast<uintptr_t>(tex) & 15) == 0 &&
         !(d->flags & LSI_HAS_MASK);
}

int canvas_channels(const LsiSplatDesc* d) {
  return (d->flags & LSI_WANT_DISP) ? 5 : 4;
}

int canvas_count(const LsiSplatDesc* d) {
  return ((d->flags & LSI_COMPOSE) && !(d->flags & LSI_WANT_DISP)) ? 1 : d->L;
}

size_t rowband_lds_bytes(const LsiSplatDesc* d, int R) {
  const int nch = canvas_channels(d);
  size_t fl = (size_t)R * d->Wt * nch;
  if ((d->flags & LSI_COMPOSE) && (d->flags & LSI_WANT_DISP))
    fl += (size_t)R * d->Wt * 5;
  return fl * sizeof(float) + 16;
}

// Band height: as tall as the LDS budget allows while keeping >= ~2 workgroups
// per CU in flight when the problem is large enough.
int pick_band_rows(const LsiSplatDesc* d) {
  const size_t budget = 64 * 1024;  // two workgroups per CU (160 KiB LDS)
  int best = 1;
  for (int R = 1; R <= d->Ht; R *= 2) {
    if (rowband_lds_bytes(d, R) > budget) break;
    const long groups = (long)((d->Ht + R - 1) / R) * d->B;
    if (R > 1 && groups < 512) break;
    best = R;
  }
  return best;
}

}  // namespace

extern "C" {

int lsi_rowband_ok(const LsiSplatDesc* d, const float* M) {
  if (check_desc(d) != LSI_OK || !M) return 0;
  if (rowband_lds_bytes(d, 1) > 160 * 1024 - 256) return 0;
  for (int b = 0; b < d->B; ++b) {
    const float* m = M + 16 * b;
    if (m[7] != 0.0f || m[11] != 0.0f) return 0;  // M[1][3], M[2][3]
    // normaliser n = m20*x + m21*y + m22 must be safely positive over the
    // image (it is linear: check the four corner pixel centres).
    float nmin = INFINITY, nmax = 0.0f;
    const float xs[2] = {0.5f, (float)d->W - 0.5f};
    const float ys[2] = {0.5f, (float)d->H - 0.5f};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) {
        const float n = m[8] * xs[i] + m[9] * ys[j] + m[10];
        if (!(n == n)) return 0;
        nmin = fminf(nmin, n);
        nmax = fmaxf(nmax, fabsf(n));
      }
    if (!(nmin > 1e-3f * nmax) || !(nmin > 0.0f)) return 0;
    for (int k = 0; k < 16; ++k)
      if (!isfinite(m[k])) return 0;
  }
  return 1;
}

size_t lsi_splat_workspace_bytes(const LsiSplatDesc* d) {
  if (check_desc(d) != LSI_OK) return 0;
  // ATOMIC-path canvases (the ROWBAND path needs none; sized for the worst
  // case so that one allocation serves either path).
  const size_t atomic_need = (size_t)canvas_count(d) * d->B * d->Ht * d->Wt *
                             canvas_channels(d) * sizeof(float);
  // STREAM-path boundary-row exchange area (worst case band height)
  const size_t stream_need = lsi_stream_workspace_bytes(d);
  // TILE-path disparity ranges (larger than the canvases only for targets of
  // a few cells)
  const size_t tile_need = lsi_tile_workspace_bytes(d);
  // A descriptor that names its path asks for that path's need only (STREAM may
  // hand a request over to TILE); AUTO / ATOMIC: one allocation serves any path.
  if (d->path == LSI_PATH_STREAM)
    return stream_need > tile_need ? stream_need : tile_need;
  if (d->path == LSI_PATH_TILE) return tile_need;
  size_t need = atomic_need > stream_need ? atomic_need : stream_need;
  if (tile_need > need) need = tile_need;
  return need;
}

int lsi_splat_fwd(const LsiSplatDesc* d, const float* tex, const float* disp,
                  const float* mask, const float* M, float* out_img,
                  float* out_wts, float* out_disp, void* workspace,
                  size_t workspace_bytes, lsi_stream_t stream_) {
  int rc = check_desc(d);
  if (rc != LSI_OK) return rc;
  if (!tex || !disp || !M || !out_img || !out_wts) return LSI_ENULL;
  if (!packed_ok(d, tex, disp)) return LSI_EINVAL;
  if ((d->flags & LSI_WANT_DISP) && !out_disp) return LSI_ENULL;
  if ((d->flags & LSI_HAS_MASK) && !mask) return LSI_ENULL;
  hipStream_t stream = (hipStream_t)stream_;
  SplatArgs a;
  a.d = *d;
  a.tex = tex; a.disp = disp; a.mask = mask; a.M = M;
  a.out_img = out_img; a.out_wts = out_wts; a.out_disp = out_disp;
  a.canvas = (float*)workspace;
  a.ws_bytes = workspace_bytes;
  a.nch = canvas_channels(d);
  a.ncanv = canvas_count(d);
  a.shared = (d->flags & LSI_COMPOSE) && !(d->flags & LSI_WANT_DISP);
  a.band_rows = 0;
  a.out_img_c = a.out_wts_c = nullptr;

  int path = d->path;
  // AUTO without a host copy of M: the any-pose tile path (callers that have
  // one ask lsi_stream_ok first)
  if (path == LSI_PATH_AUTO) path = LSI_PATH_TILE;
  if (path == LSI_PATH_TILE) return lsi_tile_launch(a, stream);
  if (path == LSI_PATH_ROWBAND) {
    const int R = d->tune_rows > 0 ? d->tune_rows : pick_band_rows(d);
    const size_t lds = rowband_lds_bytes(d, R);
    if (lds > 160 * 1024) return LSI_EINVAL;
    a.band_rows = R;
    const int threads = d->tune_threads > 0 ? d->tune_threads : 512;
    dim3 grid((d->Ht + R - 1) / R, d->B);
    if (a.nch == 5) {
      if (hipFuncSetAttribute((const void*)splat_rowband_kernel<5>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return LSI_ELAUNCH;
      hipLaunchKernelGGL(splat_rowband_kernel<5>, grid, dim3(threads), lds,
                         stream, a);
    } else {
      if (hipFuncSetAttribute((const void*)splat_rowband_kernel<4>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return LSI_ELAUNCH;
      hipLaunchKernelGGL(splat_rowband_kernel<4>, grid, dim3(threads), lds,
                         stream, a);
    }
    return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
  }
  if (path == LSI_PATH_STREAM) return lsi_stream_launch(a, stream);
  if (path != LSI_PATH_ATOMIC) return LSI_EINVAL;

  const size_t need = (size_t)canvas_count(d) * d->B * d->Ht * d->Wt *
                      canvas_channels(d) * sizeof(float);
  if (!workspace) return LSI_ENULL;
  if (workspace_bytes < need) return LSI_EWORKSPACE;
  if (hipMemsetAsync(workspace, 0, need, stream) != hipSuccess)
    return LSI_ELAUNCH;
  const int npx = d->H * d->W;
  hipLaunchKernelGGL(splat_atomic_kernel, dim3((npx + 255) / 256, d->B, d->L),
                     dim3(256), 0, stream, a);
  const size_t nt = (size_t)d->B * d->Ht * d->Wt;
  hipLaunchKernelGGL(splat_epilogue_kernel, dim3((unsigned)((nt + 255) / 256)),
                     dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

int lsi_project_indices(const LsiSplatDesc* d, const float* disp,
                        const float* mask, const float* M, int32_t* idx4,
                        float* upd4, lsi_stream_t stream_) {
  int rc = check_desc(d);
  if (rc != LSI_OK) return rc;
  if (!disp || !M || !idx4 || !upd4) return LSI_ENULL;
  if ((d->flags & LSI_HAS_MASK) && !mask) return LSI_ENULL;
  SplatArgs a;
  a.d = *d;
  a.tex = nullptr; a.disp = disp; a.mask = mask; a.M = M;
  a.out_img = a.out_wts = a.out_disp = a.canvas = nullptr;
  a.ws_bytes = 0;
  a.nch = 4; a.ncanv = 1; a.shared = 0; a.band_rows = 0;
  a.out_img_c = a.out_wts_c = nullptr;
  const int npx = d->H * d->W;
  hipLaunchKernelGGL(project_indices_kernel,
                     dim3((npx + 255) / 256, d->B, d->L), dim3(256), 0,
                     (hipStream_t)stream_, a, idx4, upd4);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

size_t lsi_splat_bwd_workspace_bytes(const LsiSplatDesc* d) {
  if (check_desc(d) != LSI_OK) return 0;
  const int nl = (d->flags & LSI_COMPOSE) ? 1 : d->L;
  return (size_t)nl * d->B * d->Ht * d->Wt * sizeof(float4);
}

int lsi_splat_bwd(const LsiSplatDesc* d, const float* tex, const float* disp,
                     const float* mask, const float* M, const float* out_img,
                     const float* out_wts, const float* g_img,
                     const float* g_wts, float* g_tex, float* g_disp_in,
                     float* g_mask, void* workspace, size_t workspace_bytes,
                     lsi_stream_t stream_) {
  int rc = check_desc(d);
  if (rc != LSI_OK) return rc;
  if (!tex || !disp || !M || !out_img || !out_wts || !g_img || !g_tex ||
      !g_disp_in || !workspace)
    return LSI_ENULL;
  if ((d->flags & LSI_HAS_MASK) && !mask) return LSI_ENULL;
  if (!packed_ok(d, tex, disp)) return LSI_EINVAL;
  if (workspace_bytes < lsi_splat_bwd_workspace_bytes(d)) return LSI_EWORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  // rectified pairs rendered by STREAM: the streamed gather (16-byte loads and
  // stores, the band's gradient-canvas rows built in LDS; no pre-pass)
  if (lsi_bwd_stream_applies(d, tex, disp, mask, g_tex, g_disp_in, g_mask)) {
    const LsiBwdCanvas ci = {out_img, out_wts, g_img, g_wts};
    return lsi_bwd_stream_launch(d, tex, disp, mask, M, &ci, nullptr, g_tex,
                                 g_disp_in, g_mask, stream);
  }
  const int nl = (d->flags & LSI_COMPOSE) ? 1 : d->L;
  const size_t n = (size_t)nl * d->B * d->Ht * d->Wt;
  hipLaunchKernelGGL(splat_bwd_pre_kernel, dim3((unsigned)((n + 255) / 256)),
                     dim3(256), 0, stream, n, out_img, out_wts, g_img, g_wts,
                     (float4*)workspace);
  SplatArgs a;
  a.d = *d;
  a.tex = tex; a.disp = disp; a.mask = mask; a.M = M;
  a.out_img = a.out_wts = a.out_disp = a.canvas = nullptr;
  a.ws_bytes = 0;
  a.nch = 4; a.ncanv = 1; a.shared = 0; a.band_rows = 0;
  a.out_img_c = a.out_wts_c = nullptr;
  if ((long)d->B * d->L > 65535 || d->H > 65535) return LSI_EINVAL;  // grid.y / z
  launch_bwd(a, (const float4*)workspace, g_tex, g_disp_in,
             (d->flags & LSI_HAS_MASK) ? g_mask : nullptr, stream);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

int lsi_splat_fwd_both(const LsiSplatDesc* d, const float* tex,
                       const float* disp, const float* mask, const float* M,
                       float* out_img, float* out_wts, float* out_img_c,
                       float* out_wts_c, void* workspace,
                       size_t workspace_bytes, lsi_stream_t stream_) {
  int rc = check_desc(d);
  if (rc != LSI_OK) return rc;
  if (d->flags & (LSI_COMPOSE | LSI_WANT_DISP)) return LSI_EINVAL;
  if (!out_img_c || !out_wts_c || !out_img || !out_wts) return LSI_ENULL;
  if (tex && disp && !packed_ok(d, tex, disp)) return LSI_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (d->path == LSI_PATH_STREAM) {
    // one sweep: every layer's tile is written out and added to a second tile
    if (!tex || !disp || !M) return LSI_ENULL;
    if ((d->flags & LSI_HAS_MASK) && !mask) return LSI_ENULL;
    SplatArgs a;
    a.d = *d;
    a.tex = tex; a.disp = disp; a.mask = mask; a.M = M;
    a.out_img = out_img; a.out_wts = out_wts; a.out_disp = nullptr;
    a.canvas = (float*)workspace; a.ws_bytes = workspace_bytes;
    a.nch = 4; a.ncanv = d->L; a.shared = 0; a.band_rows = 0;
    a.out_img_c = out_img_c; a.out_wts_c = out_wts_c;
    return lsi_stream_launch(a, stream);
  }
  // other kernel families: the per-layer pass, then the layers are summed
  rc = lsi_splat_fwd(d, tex, disp, mask, M, out_img, out_wts, nullptr,
                     workspace, workspace_bytes, stream_);
  if (rc != LSI_OK) return rc;
  const size_t n1 = (size_t)d->B * d->Ht * d->Wt;
  hipLaunchKernelGGL(compose_from_layers_kernel,
                     dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream,
                     n1, d->L, out_img, out_wts, out_img_c, out_wts_c);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

int lsi_splat_bwd_both(const LsiSplatDesc* d, const float* tex,
                       const float* disp, const float* mask, const float* M,
                       const float* out_img, const float* out_wts,
                       const float* out_img_c, const float* out_wts_c,
                       const float* g_img, const float* g_wts,
                       const float* g_img_c, const float* g_wts_c, float* g_tex,
                       float* g_disp_in, float* g_mask, void* workspace,
                       size_t workspace_bytes, lsi_stream_t stream_) {
  int rc = check_desc(d);
  if (rc != LSI_OK) return rc;
  if (d->flags & LSI_COMPOSE) return LSI_EINVAL;
  if (!tex || !disp || !M || !g_tex || !g_disp_in || !workspace) return LSI_ENULL;
  if (!packed_ok(d, tex, disp)) return LSI_EINVAL;
  if (g_img && (!out_img || !out_wts)) return LSI_ENULL;
  if (g_img_c && (!out_img_c || !out_wts_c)) return LSI_ENULL;
  if ((d->flags & LSI_HAS_MASK) && !mask) return LSI_ENULL;
  if (workspace_bytes < lsi_splat_bwd_workspace_bytes(d)) return LSI_EWORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (lsi_bwd_stream_applies(d, tex, disp, mask, g_tex, g_disp_in, g_mask)) {
    const LsiBwdCanvas ci = {out_img, out_wts, g_img, g_wts};
    const LsiBwdCanvas cc = {out_img_c, out_wts_c, g_img_c, g_wts_c};
    return lsi_bwd_stream_launch(d, tex, disp, mask, M, &ci, &cc, g_tex, g_disp_in,
                                 g_mask, stream);
  }
  const size_t n1 = (size_t)d->B * d->Ht * d->Wt;
  hipLaunchKernelGGL(splat_bwd_pre_both_kernel,
                     dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream,
                     n1, d->L, out_img, out_wts, out_img_c, out_wts_c, g_img,
                     g_wts, g_img_c, g_wts_c, (float4*)workspace);
  SplatArgs a;
  a.d = *d;
  a.tex = tex; a.disp = disp; a.mask = mask; a.M = M;
  a.out_img = a.out_wts = a.out_disp = a.canvas = nullptr;
  a.ws_bytes = 0;
  a.nch = 4; a.ncanv = 1; a.shared = 0; a.band_rows = 0;
  a.out_img_c = a.out_wts_c = nullptr;
  if ((long)d->B * d->L > 65535 || d->H > 65535) return LSI_EINVAL;  // grid.y / z
  launch_bwd(a, (const float4*)workspace, g_tex, g_disp_in,
             (d->flags & LSI_HAS_MASK) ? g_mask : nullptr, stream);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

}  // extern "C"
