// Backward of the forward splat for rectified pairs, streamed like the forward
// STREAM path (gfx950).  Reference semantics: the gradient TF1 derives for
// ldi.py:129-182 + sampling.py:161-241 (zero gradient through the
// floor / clip / 1e-3 clamp decisions), closed form as in lsi_splat.hip's
// splat_bwd_core<SIMPLE_M = true>.
//
// When it applies (decided by lsi_splat_bwd / lsi_splat_bwd_both from the
// descriptor the forward was launched with): path STREAM with the SIMPLE bit
// (every batch element: M rows 2, 3 = (0,0,1,0), (0,0,0,1), M[1][0] = M[1][3]
// = 0 -- the target row depends on the source row only, normaliser 1, target
// disparity = source disparity), channels-last contiguous textures (or RGBD
// pixels), W % 4 == 0; an optional mask with unit pixel stride.
//
// * Workgroup = (band of RS consecutive SOURCE rows, batch element[, layer]).
//   Every source pixel is written exactly once whatever the map does.  The
//   rows of the gradient canvas G the band can touch (Y(y) is monotone in y:
//   the band's first and last row bound them) are staged in LDS once, so the
//   four corner gathers of a pixel are `ds_read_b128`s; a band whose rows do
//   not fit (vertical zoom-in) gathers from global memory instead.
// * Item = (source row, 256-pixel segment, layer): a lane loads 4 consecutive
//   pixels with four 16-byte loads (disparities, 3 x rgb) and stores their
//   gradients with four 16-byte stores; the next item's loads are in flight
//   while this one computes.  Items go round-robin over the waves.
// * Row-uniform terms (the y axis of the footprint: cells, weights, border
//   masks) are computed once per item from scalars.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"
#include "lsi_splat_internal.h"

#pragma clang fp contract(off)

using namespace lsi;

namespace {

#ifndef LSI_BS_WPE
#define LSI_BS_WPE 2          // waves per SIMD the register budget allows (4: spills, 1.7x slower)
#endif
#ifndef LSI_BS_FENCE
#define LSI_BS_FENCE 1        // pixels gathered at a time: 1 << LSI_BS_FENCE (0 1 2)
#endif
#ifndef LSI_BS_T
#define LSI_BS_T 512
#endif
constexpr int BS_T = LSI_BS_T;  // threads per workgroup
constexpr int BS_NW = BS_T / 64;
constexpr int BS_SEG = 256;   // source pixels per item

// One set of forward outputs and their incoming gradients (g_img == NULL: not
// present): the per-layer (or the only) canvases, and lsi_splat_bwd_both's
// composed one.
struct BSCanvas {
  const float* img;
  const float* wts;
  const float* g_img;
  const float* g_wts;  // may be NULL
};

struct BSArgs {
  const float* tex;
  const float* disp;
  const float* M;
  const float* mask;  // L x B x H x W (unit pixel stride) or NULL
  float* g_mask;      // contiguous L x B x H x W, with mask
  int mask_sb, mask_sl, mask_sy;
  BSCanvas ci, cc;
  int vec4;  // canvas rows can be read 4 cells at a time (16-byte loads)
  float* g_tex;
  float* g_disp;
  int nt;             // streaming (non-temporal) stores of the gradients
  int B, H, W, Ht, Wt, L, nseg;
  int tex_sb, tex_sl, tex_sy, disp_sb, disp_sl, disp_sy;
  float s, max_disp, zscale, zA, zB;
  int compose;  // 1: one canvas for all layers (grid.z = 1), 0: grid.z = layer
  int remap;    // see the kernel: the layers of a band as neighbours on one XCD
  int RS;       // source rows per band
  int GR;       // canvas rows the LDS tile holds (0: always gather from global)
};

struct BSIn { float4 d4, t0, t1, t2, mk; };

// Gradient w.r.t. the un-normalised canvases A (3 ch) and W of one cell:
//   img = A / W',  wts = W,  W' = W + 1e-8 [W == 0]
//   gA = g_img / W'          gW = g_wts - sum_c g_img_c * img_c / W'
__device__ __forceinline__ float4 bs_pre(float i0, float i1, float i2, float w,
                                         float g0, float g1, float g2, float gw) {
  const float inv = 1.0f / safe_den(w);
  return make_float4(g0 * inv, g1 * inv, g2 * inv,
                     gw - (g0 * i0 + g1 * i1 + g2 * i2) * inv);
}
// ... of cell `o` of one canvas set (o: flat index into that set's arrays)
__device__ __forceinline__ float4 bs_cell(const BSCanvas& c, size_t o) {
  return bs_pre(c.img[3 * o], c.img[3 * o + 1], c.img[3 * o + 2], c.wts[o],
                c.g_img[3 * o], c.g_img[3 * o + 1], c.g_img[3 * o + 2],
                c.g_wts ? c.g_wts[o] : 0.0f);
}
// The gradient canvas G of (layer canvas lc, batch element b) at cell `cell`:
// the layer's own outputs' share plus the composed outputs' (whose canvas is
// the sum of the layers' canvases, ldi.py:167-171).
__device__ __forceinline__ float4 bs_G(const BSArgs& a, size_t oi, size_t oc) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.cc.g_img) g = bs_cell(a.cc, oc);
  if (a.ci.g_img) {
    const float4 h = bs_cell(a.ci, oi);
    g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
  }
  return g;
}
// Four consecutive cells at once (16-byte loads; o % 4 == 0, aligned arrays)
__device__ __forceinline__ void bs_cell4(const BSCanvas& c, size_t o, float4 (&g)[4],
                                         bool add) {
  const float4 ia = *reinterpret_cast<const float4*>(c.img + 3 * o);
  const float4 ib = *reinterpret_cast<const float4*>(c.img + 3 * o + 4);
  const float4 ic = *reinterpret_cast<const float4*>(c.img + 3 * o + 8);
  const float4 w = *reinterpret_cast<const float4*>(c.wts + o);
  const float4 ga = *reinterpret_cast<const float4*>(c.g_img + 3 * o);
  const float4 gb = *reinterpret_cast<const float4*>(c.g_img + 3 * o + 4);
  const float4 gc = *reinterpret_cast<const float4*>(c.g_img + 3 * o + 8);
  float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c.g_wts) gw = *reinterpret_cast<const float4*>(c.g_wts + o);
  float4 h[4];
  h[0] = bs_pre(ia.x, ia.y, ia.z, w.x, ga.x, ga.y, ga.z, gw.x);
  h[1] = bs_pre(ia.w, ib.x, ib.y, w.y, ga.w, gb.x, gb.y, gw.y);
  h[2] = bs_pre(ib.z, ib.w, ic.x, w.z, gb.z, gb.w, gc.x, gw.z);
  h[3] = bs_pre(ic.y, ic.z, ic.w, w.w, gc.y, gc.z, gc.w, gw.w);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (add) { g[k].x += h[k].x; g[k].y += h[k].y; g[k].z += h[k].z; g[k].w += h[k].w; }
    else g[k] = h[k];
  }
}

// One item's inputs: 4 disparities + 4 x rgb, or (PACK) 4 RGBD pixels, in which
// case d4, t0, t1, t2 hold the lane's pixels 0 .. 3 as (r, g, b, d).
template <bool PACK>
__device__ __forceinline__ void bs_load(BSIn& o, const float* pd, const float* pt) {
  if (PACK) {
    o.d4 = *reinterpret_cast<const float4*>(pt);
    o.t0 = *reinterpret_cast<const float4*>(pt + 4);
    o.t1 = *reinterpret_cast<const float4*>(pt + 8);
    o.t2 = *reinterpret_cast<const float4*>(pt + 12);
  } else {
    o.d4 = *reinterpret_cast<const float4*>(pd);
    o.t0 = *reinterpret_cast<const float4*>(pt);
    o.t1 = *reinterpret_cast<const float4*>(pt + 4);
    o.t2 = *reinterpret_cast<const float4*>(pt + 8);
  }
}

typedef float __attribute__((ext_vector_type(4))) bs_f4v;
typedef const __attribute__((address_space(3))) bs_f4v bs_lds_f4;

// The wave's items, two register sets of loads in flight.  IN_LDS: the band's
// canvas rows glo .. are in the LDS tile `gt`; else gathered from Gb.
template <bool IN_LDS, bool PACK, bool MASK>
__device__ __forceinline__ void bs_run(const BSArgs& a, const float (&m)[8],
                                       BSIn (&set)[2], const float4* gt,
                                       size_t obi, size_t obc, int b,
                                       int l_lo, int NL, int ys, int nitem,
                                       int glo, int wave, int lane) {
  const int Wt = a.Wt, nseg = a.nseg;
  const int half = (Wt + 1) >> 1;
  const float s = a.s;
  const float xmax = (float)Wt - 1.0f, ymax = (float)a.Ht - 1.0f;
  const float inv_md = div_rn(1.0f, a.max_disp);
  const float zs_md = a.zscale * inv_md;
  // (lanes past the end of a row -- its last segment may be partial -- read the
  // row's last four pixels instead and store nothing)
  const float* const g_tex_in = a.tex + (long)b * a.tex_sb + (long)l_lo * a.tex_sl;
  const float* const g_disp_in = a.disp + (long)b * a.disp_sb + (long)l_lo * a.disp_sl;
  const int px_last = a.W - 4;
  const float* const g_mask_in =
      MASK ? a.mask + (long)b * a.mask_sb + (long)l_lo * a.mask_sl : nullptr;
  float* const o_mask =
      MASK ? a.g_mask + ((size_t)l_lo * a.B + b) * ((size_t)a.H * a.W) + 4 * lane : nullptr;
  float* const o_tex = a.g_tex + ((size_t)l_lo * a.B + b) * ((size_t)a.H * a.W) * 3 + 12 * lane;
  float* const o_disp = a.g_disp + ((size_t)l_lo * a.B + b) * ((size_t)a.H * a.W) + 4 * lane;
  const size_t lay_px = (size_t)a.B * a.H * a.W;
  bs_lds_f4* const gt3 = (bs_lds_f4*)gt;

  // item -> (row, segment, layer), layers innermost; kept incrementally
  // (a wave's items are BS_NW apart)
  struct Pos { int r, sg, l; };
  auto advance = [&](Pos& p, int n) {
    p.l += n;
    while (p.l >= NL) { p.l -= NL; ++p.sg; }
    while (p.sg >= nseg) { p.sg -= nseg; ++p.r; }
  };
  auto load_item = [&](BSIn& o, const Pos& p, bool live) {
    const Pos q = live ? p : Pos{0, 0, 0};  // past the end: a harmless re-read
    const int px = min(q.sg * BS_SEG + 4 * lane, px_last);
    bs_load<PACK>(o, g_disp_in + (long)q.l * a.disp_sl + (long)(ys + q.r) * a.disp_sy + px,
                  g_tex_in + (long)q.l * a.tex_sl + (long)(ys + q.r) * a.tex_sy + (PACK ? 4 : 3) * px);
    if (MASK)
      o.mk = *reinterpret_cast<const float4*>(
          g_mask_in + (long)q.l * a.mask_sl + (long)(ys + q.r) * a.mask_sy + px);
  };
  auto item = [&](const BSIn& in, const Pos& p) {
    const int y = ys + p.r, sg = p.sg, l = p.l;
    // ---- row-uniform: the y axis of the footprint ---------------------------
    const float py = (float)y + 0.5f;
    const float Y = mrow(m, 1, 0.5f, py, 0.0f) * s - 0.5f;
    const Axis ay = splat_axis(Y, ymax);
    const bool yok = finite_f(Y);
    const int r0 = yok ? (int)ay.c0s : glo, r1 = yok ? (int)ay.c1s : glo;
    const float pym01 = py * m[1];
    const float dvn[4] = {in.d4.x, in.d4.y, in.d4.z, in.d4.w};
    const float dvp[4] = {in.d4.w, in.t0.w, in.t1.w, in.t2.w};
    const float txn[12] = {in.t0.x, in.t0.y, in.t0.z, in.t0.w, in.t1.x, in.t1.y,
                           in.t1.z, in.t1.w, in.t2.x, in.t2.y, in.t2.z, in.t2.w};
    const float txp[12] = {in.d4.x, in.d4.y, in.d4.z, in.t0.x, in.t0.y, in.t0.z,
                           in.t1.x, in.t1.y, in.t1.z, in.t2.x, in.t2.y, in.t2.z};
    const float (&dv)[4] = PACK ? dvp : dvn;
    const float (&tx)[12] = PACK ? txp : txn;
    const float mkv[4] = {in.mk.x, in.mk.y, in.mk.z, in.mk.w};
    float ot[12], od[4], om[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = dv[i];
      const float t0 = tx[3 * i], t1 = tx[3 * i + 1], t2 = tx[3 * i + 2];
      const float px = (float)(sg * BS_SEG + 4 * lane + i) + 0.5f;
      // q0 = ((px*m00 + py*m01) + m02) + d*m03, each op rounded (the forward's)
      const float q0 = ((px * m[0] + pym01) + m[2]) + d * m[3];
      const float X = q0 * s - 0.5f;
      // (Y carries d * 0: a non-finite disparity drops the point)
      const bool ok = yok && finite_f(X) && finite_f(d);
      const Axis ax = splat_axis(X, xmax);
      float w0 = clamp_small(ax.w0 * ay.w0), w1 = clamp_small(ax.w1 * ay.w0);
      float w2 = clamp_small(ax.w0 * ay.w1), w3 = clamp_small(ax.w1 * ay.w1);
      if (!ok) { w0 = 0.f; w1 = 0.f; w2 = 0.f; w3 = 0.f; }
      const int c0 = ok ? (int)ax.c0s : 0, c1 = ok ? (int)ax.c1s : 0;
      float4 g0, g1, g2, g3;
      if (IN_LDS) {
        // (cells of a row are stored even ones first, odd ones in the second
        // half: lanes two cells apart -- 4 pixels at s = 0.5 -- read
        // consecutive 16-byte slots instead of every other one)
        bs_lds_f4* ra = gt3 + (r0 - glo) * Wt;
        bs_lds_f4* rb = gt3 + (r1 - glo) * Wt;
        const int s0 = (c0 >> 1) + (c0 & 1) * half, s1 = (c1 >> 1) + (c1 & 1) * half;
        const bs_f4v v0 = ra[s0], v1 = ra[s1], v2 = rb[s0], v3 = rb[s1];
        g0 = make_float4(v0.x, v0.y, v0.z, v0.w); g1 = make_float4(v1.x, v1.y, v1.z, v1.w);
        g2 = make_float4(v2.x, v2.y, v2.z, v2.w); g3 = make_float4(v3.x, v3.y, v3.z, v3.w);
      } else {  // (the band's canvas rows exceed the tile: from the arrays)
        const size_t ra = (size_t)r0 * Wt, rb = (size_t)r1 * Wt;
        g0 = bs_G(a, obi + ra + c0, obc + ra + c0);
        g1 = bs_G(a, obi + ra + c1, obc + ra + c1);
        g2 = bs_G(a, obi + rb + c0, obc + rb + c0);
        g3 = bs_G(a, obi + rb + c1, obc + rb + c1);
      }
      // exp((clip(d/max,0,1) - 0.5)*scale) [d/max > 0] = exp2(clip(d,0,max)*zA + zB)
      // (the forward compact instance's form; ~1e-6 relative)
      const float xn = d * inv_md;
      const float ez = __builtin_amdgcn_exp2f(
          __fmaf_rn(__builtin_amdgcn_fmed3f(d, 0.0f, a.max_disp), a.zA, a.zB));
      const float zw = xn > 0.0f ? ez : 0.0f;
      // S_k = <tex, G_k.rgb> + G_k.w : gradient w.r.t. the corner's weight / pw
      const float S0 = __fmaf_rn(t0, g0.x, __fmaf_rn(t1, g0.y, __fmaf_rn(t2, g0.z, g0.w)));
      const float S1 = __fmaf_rn(t0, g1.x, __fmaf_rn(t1, g1.y, __fmaf_rn(t2, g1.z, g1.w)));
      const float S2 = __fmaf_rn(t0, g2.x, __fmaf_rn(t1, g2.y, __fmaf_rn(t2, g2.z, g2.w)));
      const float S3 = __fmaf_rn(t0, g3.x, __fmaf_rn(t1, g3.y, __fmaf_rn(t2, g3.z, g3.w)));
      // (a clamped / masked corner has weight exactly 0 and no gradient; its
      // canvas value is read but multiplied by that zero)
      const float a0 = __fmaf_rn(w3, g3.x, __fmaf_rn(w2, g2.x, __fmaf_rn(w1, g1.x, w0 * g0.x)));
      const float a1 = __fmaf_rn(w3, g3.y, __fmaf_rn(w2, g2.y, __fmaf_rn(w1, g1.y, w0 * g0.y)));
      const float a2 = __fmaf_rn(w3, g3.z, __fmaf_rn(w2, g2.z, __fmaf_rn(w1, g1.z, w0 * g0.z)));
      const float gpw = __fmaf_rn(w3, S3, __fmaf_rn(w2, S2, __fmaf_rn(w1, S1, w0 * S0)));
      // pixel weight = soft z-buffer weight * mask (ldi.py:145-146)
      const float pw = MASK ? zw * mkv[i] : zw;
      ot[3 * i] = a0 * pw; ot[3 * i + 1] = a1 * pw; ot[3 * i + 2] = a2 * pw;
      if (MASK) om[i] = gpw * zw;
      const float k0 = w0 != 0.0f ? pw * S0 : 0.0f, k1 = w1 != 0.0f ? pw * S1 : 0.0f;
      const float k2 = w2 != 0.0f ? pw * S2 : 0.0f, k3 = w3 != 0.0f ? pw * S3 : 0.0f;
      // corner weights -> X: d wx0/dX = -v0, d wx1/dX = +v1
      const float gX = -ax.v0 * (k0 * ay.w0 + k2 * ay.w1) + ax.v1 * (k1 * ay.w0 + k3 * ay.w1);
      const float inr = (xn >= 0.0f && xn <= 1.0f) ? 1.0f : 0.0f;
      const float gD = gpw * pw * zs_md * inr;
      // (M[1][3] == 0: the row coordinate does not move with the disparity)
      const float gd = (gX * s) * m[3] + gD;
      od[i] = ok ? gd : 0.0f;
      // (the gathers of the next pixels stay behind this pixel's arithmetic:
      // 16 of them in flight at once cost more registers than they hide)
      if (((i + 1) & ((1 << LSI_BS_FENCE) - 1)) == 0 && i < 3) asm volatile("" ::: "memory");
    }
    const size_t po = (size_t)l * lay_px + (size_t)y * a.W + (size_t)sg * BS_SEG;
    if (sg * BS_SEG + 4 * lane < a.W) {
      float* pt = o_tex + 3 * po;
      if (a.nt) {  // gradients are written once: streaming stores
        store_stream_f4(pt, ot[0], ot[1], ot[2], ot[3]);
        store_stream_f4(pt + 4, ot[4], ot[5], ot[6], ot[7]);
        store_stream_f4(pt + 8, ot[8], ot[9], ot[10], ot[11]);
        store_stream_f4(o_disp + po, od[0], od[1], od[2], od[3]);
        if (MASK) store_stream_f4(o_mask + po, om[0], om[1], om[2], om[3]);
      } else {
        *reinterpret_cast<float4*>(pt) = make_float4(ot[0], ot[1], ot[2], ot[3]);
        *reinterpret_cast<float4*>(pt + 4) = make_float4(ot[4], ot[5], ot[6], ot[7]);
        *reinterpret_cast<float4*>(pt + 8) = make_float4(ot[8], ot[9], ot[10], ot[11]);
        *reinterpret_cast<float4*>(o_disp + po) = make_float4(od[0], od[1], od[2], od[3]);
        if (MASK)
          *reinterpret_cast<float4*>(o_mask + po) = make_float4(om[0], om[1], om[2], om[3]);
      }
    }
  };

  Pos p0{0, 0, 0}, p1{0, 0, 0};
  advance(p0, wave);
  advance(p1, wave + BS_NW);
  for (int it = wave; it < nitem; it += 2 * BS_NW) {
    item(set[0], p0);
    advance(p0, 2 * BS_NW);
    load_item(set[0], p0, it + 2 * BS_NW < nitem);
    if (it + BS_NW < nitem) item(set[1], p1);
    advance(p1, 2 * BS_NW);
    load_item(set[1], p1, it + 3 * BS_NW < nitem);
  }
}

template <bool PACK, bool MASK>
__global__ __launch_bounds__(BS_T, LSI_BS_WPE) void splat_bwd_stream_kernel(BSArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* const gt = reinterpret_cast<float4*>(smem);  // [GR][Wt]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Which (band, batch element, layer).  One workgroup per layer (both outputs:
  // the composed canvas' rows are read by every layer's workgroup of a band):
  // the L workgroups of a band are made neighbours in the dispatch order of ONE
  // XCD (workgroup i runs on XCD i % 8), so that they are resident together and
  // the composed rows they share come from that XCD's L2 once, not L times from
  // memory (a.remap; LSI_BWD_REMAP=0: the natural order).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.remap) {
    const unsigned nbx = gridDim.x, L = gridDim.z;
    const unsigned lin = blockIdx.x + nbx * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = lin & 7u, slot = lin >> 3;
    const unsigned l = slot % L, pair = (slot / L) * 8u + xcd;
    bz = (int)l;
    by = (int)(pair / nbx);
    bx = (int)(pair - (unsigned)by * nbx);
  }
  const int b = by;
  const int ys = bx * a.RS, ye = min(a.H, ys + a.RS);
  const int l_lo = a.compose ? 0 : bz;
  const int NL = a.compose ? a.L : 1;
  const int Wt = a.Wt;
  const int nseg = a.nseg;
  const int nitem = (ye - ys) * nseg * NL;

  // ---- loads of the wave's first two items, before anything else -------------
  BSIn set[2];
  {
    const float* const g_tex_in = a.tex + (long)b * a.tex_sb + (long)l_lo * a.tex_sl;
    const float* const g_disp_in = a.disp + (long)b * a.disp_sb + (long)l_lo * a.disp_sl;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int it = wave + k * BS_NW;
      const int itc = it < nitem ? it : 0;
      const int rs = itc / NL, l = itc - rs * NL;
      const int r = rs / nseg, sg = rs - r * nseg;
      const int px = min(sg * BS_SEG + 4 * lane, a.W - 4);
      bs_load<PACK>(set[k], g_disp_in + (long)l * a.disp_sl + (long)(ys + r) * a.disp_sy + px,
                    g_tex_in + (long)l * a.tex_sl + (long)(ys + r) * a.tex_sy + (PACK ? 4 : 3) * px);
      if (MASK)
        set[k].mk = *reinterpret_cast<const float4*>(
            a.mask + (long)b * a.mask_sb + (long)(l_lo + l) * a.mask_sl +
            (long)(ys + r) * a.mask_sy + px);
    }
  }

  // ---- the projection, the band's canvas rows --------------------------------
  const float* __restrict__ mg = a.M + 16 * b;
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = mg[k];
  const float s = a.s;
  const float ymax = (float)a.Ht - 1.0f;
  auto row_Y = [&](int y) {
    const float py = (float)y + 0.5f;
    return mrow(m, 1, 0.5f, py, 0.0f) * s - 0.5f;
  };
  int glo = 0, ghi = 0;
  {
    const Axis a0 = splat_axis(row_Y(ys), ymax), a1 = splat_axis(row_Y(max(ye - 1, ys)), ymax);
    // (non-finite Y: no pixel of the row has a gradient; any rows do)
    const float lo = fminf(fminf(a0.c0s, a1.c0s), fminf(a0.c1s, a1.c1s));
    const float hi = fmaxf(fmaxf(a0.c0s, a1.c0s), fmaxf(a0.c1s, a1.c1s));
    glo = finite_f(lo) ? (int)lo : 0;
    ghi = finite_f(hi) ? (int)hi : 0;
  }
  const int grow = ghi - glo + 1;
  const bool in_lds = grow <= a.GR;
  const size_t P = (size_t)a.Ht * Wt;
  // cell 0 of this workgroup's canvases in the per-layer / composed arrays
  const size_t obi = ((size_t)l_lo * a.B + b) * P, obc = (size_t)b * P;
  if (in_lds) {
    // the gradient canvas of the band's rows, straight from the forward's
    // outputs and their incoming gradients (no pre-pass, no G in memory)
    const size_t c0 = (size_t)glo * Wt;
    const int n = grow * Wt;
    if (a.vec4) {
      const int half = (Wt + 1) >> 1;
      const float inv_wt = 1.0f / (float)Wt;
      for (int i = 4 * tid; i < n; i += 4 * BS_T) {
        float4 g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.cc.g_img) bs_cell4(a.cc, obc + c0 + i, g, false);
        if (a.ci.g_img) bs_cell4(a.ci, obi + c0 + i, g, a.cc.g_img != nullptr);
        // (Wt % 4 == 0: the four cells are in one row; slot = even/odd halves)
        int row = (int)((float)i * inv_wt);  // i < 2^22: exact after one fix-up
        int c = i - row * Wt;
        if (c >= Wt) { ++row; c -= Wt; }
        if (c < 0) { --row; c += Wt; }
        float4* dst = gt + row * Wt + (c >> 1);
        dst[0] = g[0]; dst[half] = g[1]; dst[1] = g[2]; dst[half + 1] = g[3];
      }
    } else {
      const int half = (Wt + 1) >> 1;
      for (int i = tid; i < n; i += BS_T) {
        const int row = i / Wt, c = i - row * Wt;
        gt[row * Wt + (c >> 1) + (c & 1) * half] = bs_G(a, obi + c0 + i, obc + c0 + i);
      }
    }
  }
  __syncthreads();
  if (nitem <= 0) return;
  if (in_lds)
    bs_run<true, PACK, MASK>(a, m, set, gt, obi, obc, b, l_lo, NL, ys, nitem, glo, wave, lane);
  else
    bs_run<false, PACK, MASK>(a, m, set, gt, obi, obc, b, l_lo, NL, ys, nitem, glo, wave, lane);
}

bool aligned16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Descriptor-level test: what lsi_splat_bwd / lsi_splat_bwd_both ask before
// they take this path (the forward's lsi_stream_ok looked at M on the host and
// left its verdict in the descriptor).
bool lsi_bwd_stream_applies(const LsiSplatDesc* d, const float* tex,
                            const float* disp, const float* mask,
                            const float* g_tex, const float* g_disp,
                            const float* g_mask) {
  if (const char* e = getenv("LSI_BWD_STREAM"))
    if (e[0] == '0') return false;
  if (d->path != LSI_PATH_STREAM || !(d->tune_window & LSI_STREAM_SIMPLE_BIT))
    return false;
  if (d->flags & LSI_WANT_DISP) return false;
  if (d->flags & LSI_HAS_MASK) {
    if (!mask || !g_mask || (d->flags & LSI_PACKED_RGBD) || d->mask_sx != 1) return false;
    const int64_t ms[] = {d->mask_sl, d->mask_sb, d->mask_sy};
    for (int64_t v : ms)
      if (v < 0 || v > 0x7fffffffLL || v % 4) return false;
    if (!aligned16b(mask) || !aligned16b(g_mask)) return false;
  }
  if (d->W % 4 != 0 || d->W < 4 || d->L < 1) return false;
  // (LSI_PACKED_RGBD: the entry points have verified the caller's statement)
  const bool pack = (d->flags & LSI_PACKED_RGBD) != 0;
  if (!pack && (d->tex_sc != 1 || d->tex_sx != 3 || d->disp_sx != 1)) return false;
  const int64_t st[] = {d->tex_sl, d->tex_sb, d->tex_sy, d->disp_sl, d->disp_sb, d->disp_sy};
  for (int64_t v : st)
    if (v < 0 || v > 0x7fffffffLL || v % 4) return false;
  if (!aligned16b(tex) || (!pack && !aligned16b(disp)) || !aligned16b(g_tex) ||
      !aligned16b(g_disp))
    return false;
  if ((long)d->B > 65535 || (long)d->L > 65535) return false;
  return true;
}

int lsi_bwd_stream_launch(const LsiSplatDesc* d, const float* tex,
                          const float* disp, const float* mask, const float* M,
                          const LsiBwdCanvas* ci, const LsiBwdCanvas* cc,
                          float* g_tex, float* g_disp, float* g_mask,
                          hipStream_t stream) {
  BSArgs a;
  a.tex = tex; a.disp = disp; a.M = M; a.g_tex = g_tex; a.g_disp = g_disp;
  {
    // Plain stores: a lane's four 16-byte stores are a third of the 48 bytes it
    // owns per instruction; as non-temporal stores those partial lines are not
    // merged -- WRITE_SIZE 476 MB instead of 403 MB per launch at config 3, the
    // same time (profiles/r04/pmc_bwd_stream_cfg3.txt; LSI_BWD_NT=1 to compare)
    static const char* nt_env = getenv("LSI_BWD_NT");
    a.nt = nt_env ? atoi(nt_env) : 0;
  }
  const bool has_mask = (d->flags & LSI_HAS_MASK) != 0;
  a.mask = has_mask ? mask : nullptr; a.g_mask = has_mask ? g_mask : nullptr;
  a.mask_sb = (int)d->mask_sb; a.mask_sl = (int)d->mask_sl; a.mask_sy = (int)d->mask_sy;
  const LsiBwdCanvas none = {nullptr, nullptr, nullptr, nullptr};
  const LsiBwdCanvas& i_ = ci ? *ci : none;
  const LsiBwdCanvas& c_ = cc ? *cc : none;
  a.ci = {i_.img, i_.wts, i_.g_img, i_.g_wts};
  a.cc = {c_.img, c_.wts, c_.g_img, c_.g_wts};
  a.vec4 = d->Wt % 4 == 0;
  for (const BSCanvas* c : {&a.ci, &a.cc})
    if (c->g_img)
      a.vec4 = a.vec4 && aligned16b(c->img) && aligned16b(c->wts) &&
               aligned16b(c->g_img) && aligned16b(c->g_wts);
  a.B = d->B; a.H = d->H; a.W = d->W; a.Ht = d->Ht; a.Wt = d->Wt; a.L = d->L;
  a.nseg = (d->W + BS_SEG - 1) / BS_SEG;
  a.tex_sb = (int)d->tex_sb; a.tex_sl = (int)d->tex_sl; a.tex_sy = (int)d->tex_sy;
  a.disp_sb = (int)d->disp_sb; a.disp_sl = (int)d->disp_sl; a.disp_sy = (int)d->disp_sy;
  a.s = d->trg_downsampling; a.max_disp = d->max_disp; a.zscale = d->zbuf_scale;
  const double l2e = 1.4426950408889634;
  a.zA = (float)((double)d->zbuf_scale * l2e / (double)d->max_disp);
  a.zB = (float)(-0.5 * (double)d->zbuf_scale * l2e);
  a.compose = (d->flags & LSI_COMPOSE) ? 1 : 0;
  // Source rows per band: the largest of 32 / 16 / 8 whose canvas rows
  // (RS * s + 3: the same-intrinsics estimate; the kernel checks the real
  // span) fit the LDS, halved while the grid would leave CUs without work.
  auto rows_for = [&](int rs) { return (int)ceilf((float)rs * d->trg_downsampling) + 3; };
  auto bytes_for = [&](int rs) { return (size_t)rows_for(rs) * d->Wt * 16; };
  const size_t cap = 156 * 1024;
  int rs = 0;
  if (const char* e = getenv("LSI_BWD_STREAM_ROWS")) rs = atoi(e);
  if (rs <= 0) {
    rs = 8;
    for (int c : {32, 16, 8})
      if (bytes_for(c) <= cap) { rs = c; break; }
    const long nz = a.compose ? 1 : d->L;
    while (rs > 8 && (long)((d->H + rs - 1) / rs) * d->B * nz < 256) rs >>= 1;
  }
  a.RS = rs;
  a.GR = rows_for(rs);
  size_t lds = bytes_for(rs);
  if (lds > cap) { a.GR = 0; lds = 0; }
  const void* fn = (d->flags & LSI_PACKED_RGBD)
                       ? (const void*)splat_bwd_stream_kernel<true, false>
                       : (has_mask ? (const void*)splat_bwd_stream_kernel<false, true>
                                   : (const void*)splat_bwd_stream_kernel<false, false>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(lds > 0 ? lds : 16)) != hipSuccess) {
    // a device that does not grant the LDS asked for: corners from the arrays
    (void)hipGetLastError();
    a.GR = 0; lds = 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 16) !=
        hipSuccess)
      return LSI_ELAUNCH;
  }
  const dim3 grid((d->H + rs - 1) / rs, d->B, a.compose ? 1 : d->L);
  {
    static const char* rm = getenv("LSI_BWD_REMAP");
    const long nwg = (long)grid.x * grid.y * grid.z;
    a.remap = (!a.compose && grid.z > 1 && nwg % (8L * grid.z) == 0 &&
               !(rm && rm[0] == '0')) ? 1 : 0;
  }
  void* kargs[1] = {&a};
  if (hipLaunchKernel(fn, grid, dim3(BS_T), kargs, lds, stream) != hipSuccess)
    return LSI_ELAUNCH;
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
