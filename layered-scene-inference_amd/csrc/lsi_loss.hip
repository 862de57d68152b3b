// Fused single-pass kernels for the losses on LDIs and rendered views, and for
// layer composition (gfx950).  Each loss reads its inputs once and reduces to a
// scalar; each backward reads them once more and writes the gradients.  The
// reference builds every one of these from 10-20 stock TF elementwise / reduce
// ops (one memory round trip each, plus the autodiff mirror):
//   zbuffer_composition_loss   lsi/loss/loss.py:66-115
//   disp_smoothness_loss       lsi/geometry/ldi.py:33-68
//   decreasing_disp_loss       lsi/loss/loss.py:48-63
//   view-synthesis loss        ldi_enc_dec.py:337-357 (inline)
//   compose / compose_depth    lsi/geometry/layers.py:29-115 with
//   soft_z_buffering           lsi/nnutils/helpers.py:140-160
// Bound: HBM streaming (no scatter).  Reductions: per-thread fp32, per-block and
// final sums in fp64, in a fixed order (run-to-run reproducible).
// Gradient conventions are TensorFlow's: abs'(0) = 0, relu'(0) = 0,
// clip_by_value passes the gradient on the closed interval, comparisons and
// stop_gradient have none, reduce_min splits the gradient evenly among ties.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"

using namespace lsi;

namespace {

constexpr int TPB = 256;
constexpr int MAXBLK = 2048;  // partial sums per scalar

// Block-wide sum of NV per-thread values; thread 0 stores them (fp64) to
// part[v * MAXBLK + blockIdx.x].
template <int NV>
__device__ __forceinline__ void block_store_partials(const float (&v)[NV],
                                                     double* part) {
  __shared__ double sm[NV][TPB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double x = (double)v[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) sm[k][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double s = 0.0;
      for (int w = 0; w < TPB / 64; ++w) s += sm[k][w];
      part[(size_t)k * MAXBLK + blockIdx.x] = s;
    }
  }
}

// out[k] = scale[k] * sum of part[k][0..nblk)   (one block)
struct Scales { double s[5]; };
__global__ __launch_bounds__(TPB) void finish_kernel(const double* part, int nblk,
                                                     int nv, Scales sc,
                                                     float* out, int combine) {
  __shared__ double sm[TPB];
  double total = 0.0;
  for (int k = 0; k < nv; ++k) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblk; i += TPB) x += part[(size_t)k * MAXBLK + i];
    sm[threadIdx.x] = x;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
      if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
      __syncthreads();
    }
    const double r = sm[0] * sc.s[k];
    __syncthreads();
    if (combine < 0) {
      if (threadIdx.x == 0) out[k] = (float)r;
    } else {
      // the first `combine` scalars are summed into out[0] in fp32, in order,
      // as the reference adds its four means; the rest follow one by one
      if (k < combine) {
        total = (k == 0) ? (double)(float)r : (double)((float)total + (float)r);
        if (k == combine - 1 && threadIdx.x == 0) out[0] = (float)total;
      } else if (threadIdx.x == 0) {
        out[1 + k - combine] = (float)r;
      }
    }
  }
}

int grid_for(long n) {
  long g = (n + TPB - 1) / TPB;
  if (g > MAXBLK) g = MAXBLK;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------
// zbuffer_composition_loss (loss.py:66-115)
// ---------------------------------------------------------------------------
struct ZArgs {
  LsiLossDesc d;
  const float* imgs; const float* masks; const float* disps; const float* trg;
};

// per pixel: layer weights w_l = zw(d_l / max_disp) * m_l, the white background
// layer at bg_layer_disp (mask 1), S = sum, cost = sum_l (w_l / S') e_l with
// e_l = sum_c (img_lc - trg_c)^2
template <bool BWD>
__global__ __launch_bounds__(TPB) void zbuf_comp_kernel(
    ZArgs a, double* part, const float* g_loss, float* g_imgs,
    float* g_masks, float* g_disps) {
  const LsiLossDesc& d = a.d;
  const long N = (long)d.B * d.H * d.W;
  const float md = d.max_disp, zs = d.zbuf_scale;
  // the appended background layer: disparity bg_layer_disp, mask 1
  // (loss.py:99-107; an fp32 tensor division there, unlike ldi.py:115)
  const float bg_w = zbuffer_weight(div_rn(d.bg_layer_disp, md), zs);
  float acc[1] = {0.0f};
  for (long p = (long)blockIdx.x * TPB + threadIdx.x; p < N;
       p += (long)gridDim.x * TPB) {
    const int x = (int)(p % d.W);
    const long q = p / d.W;
    const int y = (int)(q % d.H), b = (int)(q / d.H);
    const float* tp = a.trg + (long)b * d.trg_sb + (long)y * d.trg_sy +
                      (long)x * d.trg_sx;
    const float t0 = tp[0], t1 = tp[d.trg_sc], t2 = tp[2 * d.trg_sc];
    const long io = (long)b * d.img_sb + (long)y * d.img_sy + (long)x * d.img_sx;
    const long dd = (long)b * d.disp_sb + (long)y * d.disp_sy + (long)x * d.disp_sx;
    const long mo = (long)b * d.mask_sb + (long)y * d.mask_sy + (long)x * d.mask_sx;
    // pass 1: S and sum_l w_l e_l
    float S = 0.0f, num = 0.0f;
    for (int l = 0; l < d.L; ++l) {
      const float dl = a.disps[dd + (long)l * d.disp_sl];
      const float m = a.masks ? a.masks[mo + (long)l * d.mask_sl] : 1.0f;
      const float w = zbuffer_weight(div_rn(dl, md), zs) * m;
      const float* ip = a.imgs + io + (long)l * d.img_sl;
      const float e0 = ip[0] - t0, e1 = ip[d.img_sc] - t1, e2 = ip[2 * d.img_sc] - t2;
      S += w;
      num += w * ((e0 * e0 + e1 * e1) + e2 * e2);
    }
    const float eb = ((1.0f - t0) * (1.0f - t0) + (1.0f - t1) * (1.0f - t1)) +
                     (1.0f - t2) * (1.0f - t2);
    S += bg_w;
    num += bg_w * eb;
    const float Sd = safe_den(S);
    const float cost = div_rn(num, Sd);
    if (!BWD) {
      acc[0] += cost;
    } else {
      // loss = 0.5 * sum_p cost_p / (N * 3)
      const float gs = g_loss[0] * (0.5f / (float)(N * 3));
      const float rS = div_rn(1.0f, Sd);
      const long P = N;  // contiguous outputs: [l][p][c], [l][p]
      for (int l = 0; l < d.L; ++l) {
        const float dl = a.disps[dd + (long)l * d.disp_sl];
        const float m = a.masks ? a.masks[mo + (long)l * d.mask_sl] : 1.0f;
        const float xn = div_rn(dl, md);
        const float zw = zbuffer_weight(xn, zs);
        const float w = zw * m;
        const float pl = w * rS;
        const float* ip = a.imgs + io + (long)l * d.img_sl;
        const float e0 = ip[0] - t0, e1 = ip[d.img_sc] - t1, e2 = ip[2 * d.img_sc] - t2;
        const float el = (e0 * e0 + e1 * e1) + e2 * e2;
        float* gi = g_imgs + ((long)l * P + p) * 3;
        gi[0] = gs * 2.0f * e0 * pl;
        gi[1] = gs * 2.0f * e1 * pl;
        gi[2] = gs * 2.0f * e2 * pl;
        const float gw = gs * (el - cost) * rS;  // d cost / d w_l
        if (g_masks) g_masks[(long)l * P + p] = gw * zw;
        // d w / d disp: exp's derivative inside the clip's closed interval
        const float inside = (xn >= 0.0f && xn <= 1.0f) ? 1.0f : 0.0f;
        g_disps[(long)l * P + p] = gw * m * zw * inside * div_rn(zs, md);
      }
    }
  }
  if (!BWD) block_store_partials<1>(acc, part);
}

// ---------------------------------------------------------------------------
// disp_smoothness_loss (ldi.py:33-68) + decreasing_disp_loss (loss.py:48-63)
// ---------------------------------------------------------------------------
struct DArgs {
  int L, B, H, W;
  long sl, sb, sy, sx;
  const float* disp;
};

__device__ __forceinline__ float sgn(float v) {
  return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
}

// second differences anchored at (y, x): the reference's gradient() applied
// twice (forward differences of forward differences, each rounded)
struct Stencil {
  const float* p; long sy, sx; int H, W;
  __device__ __forceinline__ float at(int y, int x) const {
    return p[(long)y * sy + (long)x * sx];
  }
  __device__ __forceinline__ bool has_xx(int y, int x) const {
    return y >= 0 && y < H && x >= 0 && x + 2 < W;
  }
  __device__ __forceinline__ bool has_yy(int y, int x) const {
    return x >= 0 && x < W && y >= 0 && y + 2 < H;
  }
  __device__ __forceinline__ bool has_xy(int y, int x) const {
    return y >= 0 && x >= 0 && y + 1 < H && x + 1 < W;
  }
  __device__ __forceinline__ float xx(int y, int x) const {
    return (at(y, x + 2) - at(y, x + 1)) - (at(y, x + 1) - at(y, x));
  }
  __device__ __forceinline__ float yy(int y, int x) const {
    return (at(y + 2, x) - at(y + 1, x)) - (at(y + 1, x) - at(y, x));
  }
  // gradient(dx) along y: dx[y+1, x] - dx[y, x]
  __device__ __forceinline__ float xy(int y, int x) const {
    return (at(y + 1, x + 1) - at(y + 1, x)) - (at(y, x + 1) - at(y, x));
  }
  // gradient(dy) along x: dy[y, x+1] - dy[y, x]
  __device__ __forceinline__ float yx(int y, int x) const {
    return (at(y + 1, x + 1) - at(y, x + 1)) - (at(y + 1, x) - at(y, x));
  }
};

template <bool BWD>
__global__ __launch_bounds__(TPB) void disp_reg_kernel(DArgs a, double* part,
                                                       const float* g2,
                                                       float* g_disp) {
  const long N = (long)a.L * a.B * a.H * a.W;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // xx, xy, yx, yy, decreasing
  // divisors of the five means
  const double n_xx = (double)a.L * a.B * a.H * (a.W - 2);
  const double n_xy = (double)a.L * a.B * (a.H - 1) * (a.W - 1);
  const double n_yy = (double)a.L * a.B * (a.H - 2) * a.W;
  const double n_dc = (double)(a.L - 1) * a.B * a.H * a.W;
  float c_xx = 0.f, c_xy = 0.f, c_yy = 0.f, c_dc = 0.f;
  if (BWD) {
    c_xx = n_xx > 0 ? (float)(g2[0] / n_xx) : 0.f;
    c_xy = n_xy > 0 ? (float)(g2[0] / n_xy) : 0.f;
    c_yy = n_yy > 0 ? (float)(g2[0] / n_yy) : 0.f;
    c_dc = n_dc > 0 ? (float)(g2[1] / n_dc) : 0.f;
  }
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < N;
       i += (long)gridDim.x * TPB) {
    const int x = (int)(i % a.W);
    long q = i / a.W;
    const int y = (int)(q % a.H);
    q /= a.H;
    const int b = (int)(q % a.B), l = (int)(q / a.B);
    Stencil s;
    s.p = a.disp + (long)l * a.sl + (long)b * a.sb; s.sy = a.sy; s.sx = a.sx;
    s.H = a.H; s.W = a.W;
    if (!BWD) {
      if (s.has_xx(y, x)) acc[0] += fabsf(s.xx(y, x));
      if (s.has_xy(y, x)) { acc[1] += fabsf(s.xy(y, x)); acc[2] += fabsf(s.yx(y, x)); }
      if (s.has_yy(y, x)) acc[3] += fabsf(s.yy(y, x));
      if (l + 1 < a.L) {
        const float nxt = a.disp[(long)(l + 1) * a.sl + (long)b * a.sb +
                                 (long)y * a.sy + (long)x * a.sx];
        acc[4] += fmaxf(nxt - s.at(y, x), 0.0f);
      }
    } else {
      // The 13 values the gradient at (y, x) depends on -- row y and column x
      // two pixels each way, the 3 x 3 block around it -- loaded ONCE (zero
      // outside the image: a term that would use it is switched off by its
      // has_*); the differences below are the Stencil's, on registers.  (One
      // at() per use was ~50 loads with 64-bit address arithmetic each: 201 us
      // for the 6.3 M disparities of a 4-layer 256 x 768 pair against 71 us
      // for the forward.)
      float v[5][5];
      const float* const c = s.p + (long)y * s.sy + (long)x * s.sx;
      auto ld = [&](int dy, int dx) {
        const int yy = y + dy, xx = x + dx;
        v[dy + 2][dx + 2] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
                                ? c[(long)dy * s.sy + (long)dx * s.sx] : 0.0f;
      };
#pragma unroll
      for (int d = -2; d <= 2; ++d) { ld(0, d); if (d) ld(d, 0); }
      ld(-1, -1); ld(-1, 1); ld(1, -1); ld(1, 1);
#define LSI_V(dy, dx) v[(dy) + 2][(dx) + 2]
      // (the Stencil's xx / yy / xy / yx anchored at (y + dy, x + dx))
      auto XX = [&](int dy, int dx) {
        return (LSI_V(dy, dx + 2) - LSI_V(dy, dx + 1)) - (LSI_V(dy, dx + 1) - LSI_V(dy, dx));
      };
      auto YY = [&](int dy, int dx) {
        return (LSI_V(dy + 2, dx) - LSI_V(dy + 1, dx)) - (LSI_V(dy + 1, dx) - LSI_V(dy, dx));
      };
      auto XY = [&](int dy, int dx) {
        return (LSI_V(dy + 1, dx + 1) - LSI_V(dy + 1, dx)) - (LSI_V(dy, dx + 1) - LSI_V(dy, dx));
      };
      auto YX = [&](int dy, int dx) {
        return (LSI_V(dy + 1, dx + 1) - LSI_V(dy, dx + 1)) - (LSI_V(dy + 1, dx) - LSI_V(dy, dx));
      };
      float g = 0.0f;
      // d[y, x] enters xx(y, x) with +1, xx(y, x-1) with -2, xx(y, x-2) with +1
      if (s.has_xx(y, x)) g += c_xx * sgn(XX(0, 0));
      if (s.has_xx(y, x - 1)) g -= 2.0f * c_xx * sgn(XX(0, -1));
      if (s.has_xx(y, x - 2)) g += c_xx * sgn(XX(0, -2));
      if (s.has_yy(y, x)) g += c_yy * sgn(YY(0, 0));
      if (s.has_yy(y - 1, x)) g -= 2.0f * c_yy * sgn(YY(-1, 0));
      if (s.has_yy(y - 2, x)) g += c_yy * sgn(YY(-2, 0));
      // mixed terms: +1 at (y,x), -1 at (y,x+1), -1 at (y+1,x), +1 at (y+1,x+1)
      if (s.has_xy(y, x)) g += c_xy * (sgn(XY(0, 0)) + sgn(YX(0, 0)));
      if (s.has_xy(y, x - 1)) g -= c_xy * (sgn(XY(0, -1)) + sgn(YX(0, -1)));
      if (s.has_xy(y - 1, x)) g -= c_xy * (sgn(XY(-1, 0)) + sgn(YX(-1, 0)));
      if (s.has_xy(y - 1, x - 1)) g += c_xy * (sgn(XY(-1, -1)) + sgn(YX(-1, -1)));
#undef LSI_V
      // decreasing loss: only the farther layer of each pair gets a gradient
      if (l >= 1) {
        const float pre = a.disp[(long)(l - 1) * a.sl + (long)b * a.sb +
                                 (long)y * a.sy + (long)x * a.sx];
        if (v[2][2] - pre > 0.0f) g += c_dc;
      }
      g_disp[i] = g;
    }
  }
  if (!BWD) block_store_partials<5>(acc, part);
}

// ---------------------------------------------------------------------------
// view-synthesis loss (ldi_enc_dec.py:337-357)
// ---------------------------------------------------------------------------
struct VArgs {
  int nl, B, Ht, Wt, H, W, x_min, y_min;
  const float* recons;  // [nl, B, Ht, Wt, 3] contiguous
  const float* target;  // [B, H, W, 3] element strides below
  long t_sb, t_sy, t_sx, t_sc;
};

// AREA resize for integer factors: box mean, rows then columns in order
__device__ __forceinline__ void area_px(const VArgs& a, int b, int yt, int xt,
                                        float (&t)[3]) {
  const int fy = a.H / a.Ht, fx = a.W / a.Wt;
  t[0] = t[1] = t[2] = 0.0f;
  for (int dy = 0; dy < fy; ++dy)
    for (int dx = 0; dx < fx; ++dx) {
      const float* p = a.target + (long)b * a.t_sb +
                       (long)(yt * fy + dy) * a.t_sy + (long)(xt * fx + dx) * a.t_sx;
      t[0] += p[0]; t[1] += p[a.t_sc]; t[2] += p[2 * a.t_sc];
    }
  const float inv = 1.0f / (float)(fy * fx);
  t[0] *= inv; t[1] *= inv; t[2] *= inv;
}

__device__ __forceinline__ float layer_l1(const VArgs& a, int l, int b, int yt,
                                          int xt, const float (&t)[3]) {
  const float* r = a.recons + ((((long)l * a.B + b) * a.Ht + yt) * a.Wt + xt) * 3;
  // mean over the three channels (tf.reduce_mean: sum, then / 3)
  return ((fabsf(t[0] - r[0]) + fabsf(t[1] - r[1])) + fabsf(t[2] - r[2])) / 3.0f;
}

template <bool BWD>
__global__ __launch_bounds__(TPB) void view_synth_kernel(VArgs a, double* part,
                                                         const float* g_loss,
                                                         float* g_recons) {
  const int hc = a.Ht - 2 * a.y_min, wc = a.Wt - 2 * a.x_min;
  float acc[1] = {0.0f};
  if (!BWD) {
    const long N = (long)a.B * hc * wc;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < N;
         i += (long)gridDim.x * TPB) {
      const int xt = a.x_min + (int)(i % wc);
      const long q = i / wc;
      const int yt = a.y_min + (int)(q % hc), b = (int)(q / hc);
      float t[3];
      area_px(a, b, yt, xt, t);
      float best = layer_l1(a, 0, b, yt, xt, t);
      for (int l = 1; l < a.nl; ++l) best = fminf(best, layer_l1(a, l, b, yt, xt, t));
      acc[0] += best;
    }
    block_store_partials<1>(acc, part);
  } else {
    const long N = (long)a.B * a.Ht * a.Wt;
    const float gs = g_loss[0] / (float)((long)a.B * hc * wc);
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < N;
         i += (long)gridDim.x * TPB) {
      const int xt = (int)(i % a.Wt);
      const long q = i / a.Wt;
      const int yt = (int)(q % a.Ht), b = (int)(q / a.Ht);
      const bool in = xt >= a.x_min && xt < a.Wt - a.x_min && yt >= a.y_min &&
                      yt < a.Ht - a.y_min;
      float t[3] = {0.f, 0.f, 0.f};
      float best = 0.0f;
      int ties = 0;
      if (in) {
        area_px(a, b, yt, xt, t);
        best = layer_l1(a, 0, b, yt, xt, t);
        for (int l = 1; l < a.nl; ++l) best = fminf(best, layer_l1(a, l, b, yt, xt, t));
        for (int l = 0; l < a.nl; ++l)
          ties += layer_l1(a, l, b, yt, xt, t) == best ? 1 : 0;
      }
      for (int l = 0; l < a.nl; ++l) {
        const long o = ((((long)l * a.B + b) * a.Ht + yt) * a.Wt + xt) * 3;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (in && layer_l1(a, l, b, yt, xt, t) == best) {
          const float c = gs / (3.0f * (float)ties);
          const float* r = a.recons + o;
          g0 = c * sgn(r[0] - t[0]); g1 = c * sgn(r[1] - t[1]);
          g2 = c * sgn(r[2] - t[2]);
        }
        g_recons[o] = g0; g_recons[o + 1] = g1; g_recons[o + 2] = g2;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// compose / compose_depth (layers.py:29-115, helpers.py:140-160)
// ---------------------------------------------------------------------------
struct CArgs {
  int L, C; long N;
  const float* imgs;   // [L, N, C] (compose) -- NULL for compose_depth
  const float* masks;  // [L, N]
  const float* dmaps;  // [L, N]
  float min_disp, temp, dmax;  // dmax: max over all dmaps incl. bg (bg_layer)
  int soft, depth_mode, bg_layer;
};

// log-probability of layer l (l == L: the background layer at min_disp)
__device__ __forceinline__ float layer_logp(const CArgs& a, long n, int l) {
  float m = 1.0f, dsel;
  if (l < a.L) {
    m = a.masks[(long)l * a.N + n];
    const float dm = fmaxf(a.dmaps[(long)l * a.N + n], 0.0f);  // relu
    dsel = (a.depth_mode && a.bg_layer) ? a.dmax - dm : dm;
  } else {
    dsel = a.min_disp;
  }
  dsel = fmaxf(dsel, 0.0f);                                   // helpers.py:152
  const float depth = div_rn(1.0f, safe_den(dsel));           // divide_safe(1, d)
  const float lp = div_rn(-depth, a.temp);
  return logf(m + 1e-8f) + lp;
}

__global__ __launch_bounds__(TPB) void compose_kernel(CArgs a, float* out) {
  for (long n = (long)blockIdx.x * TPB + threadIdx.x; n < a.N;
       n += (long)gridDim.x * TPB) {
    // soft_z_buffering: probabilities exp(logp - max) / sum; the hard variants
    // take the FIRST maximum of the probabilities (tf.argmax), which can tie
    // where the log-probabilities still differ
    float mx = layer_logp(a, n, 0);
    for (int l = 1; l <= a.L; ++l) mx = fmaxf(mx, layer_logp(a, n, l));
    float sum = 0.0f;
    for (int l = 0; l <= a.L; ++l) sum += expf(layer_logp(a, n, l) - mx);
    int arg = 0;
    float pbest = div_rn(expf(layer_logp(a, n, 0) - mx), sum);
    for (int l = 1; l <= a.L; ++l) {
      const float pl = div_rn(expf(layer_logp(a, n, l) - mx), sum);
      if (pl > pbest) { pbest = pl; arg = l; }
    }
    if (a.depth_mode) {
      out[n] = arg < a.L ? fmaxf(a.dmaps[(long)arg * a.N + n], 0.0f) : a.min_disp;
      continue;
    }
    if (!a.soft) {
      for (int c = 0; c < a.C; ++c)
        out[n * a.C + c] = arg < a.L ? a.imgs[((long)arg * a.N + n) * a.C + c] : 1.0f;
      continue;
    }
    for (int c = 0; c < a.C; ++c) {
      float o = 0.0f;
      for (int l = 0; l <= a.L; ++l) {
        const float p = div_rn(expf(layer_logp(a, n, l) - mx), sum);
        o += p * (l < a.L ? a.imgs[((long)l * a.N + n) * a.C + c] : 1.0f);
      }
      out[n * a.C + c] = o;
    }
  }
}

int rc_of_launch() {
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

bool loss_desc_ok(const LsiLossDesc* d) {
  return d && d->L > 0 && d->B > 0 && d->H > 0 && d->W > 0 && d->max_disp != 0.0f;
}

}  // namespace

extern "C" {

size_t lsi_loss_workspace_bytes(void) { return (size_t)5 * MAXBLK * sizeof(double); }

int lsi_zbuf_comp_loss_fwd(const LsiLossDesc* d, const float* imgs,
                           const float* masks, const float* disps,
                           const float* trg, float* out_loss, void* ws,
                           size_t ws_bytes, lsi_stream_t stream) {
  if (!loss_desc_ok(d)) return LSI_EINVAL;
  if (!imgs || !disps || !trg || !out_loss || !ws) return LSI_ENULL;
  if (ws_bytes < lsi_loss_workspace_bytes()) return LSI_EWORKSPACE;
  ZArgs a; a.d = *d; a.imgs = imgs; a.masks = masks; a.disps = disps; a.trg = trg;
  const long N = (long)d->B * d->H * d->W;
  const int g = grid_for(N);
  hipLaunchKernelGGL(zbuf_comp_kernel<false>, dim3(g), dim3(TPB), 0,
                     (hipStream_t)stream, a, (double*)ws, nullptr, nullptr,
                     nullptr, nullptr);
  Scales sc; sc.s[0] = 0.5 / ((double)N * 3.0);
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream,
                     (const double*)ws, g, 1, sc, out_loss, -1);
  return rc_of_launch();
}

int lsi_zbuf_comp_loss_bwd(const LsiLossDesc* d, const float* imgs,
                           const float* masks, const float* disps,
                           const float* trg, const float* g_loss, float* g_imgs,
                           float* g_masks, float* g_disps, lsi_stream_t stream) {
  if (!loss_desc_ok(d)) return LSI_EINVAL;
  if (!imgs || !disps || !trg || !g_loss || !g_imgs || !g_disps) return LSI_ENULL;
  ZArgs a; a.d = *d; a.imgs = imgs; a.masks = masks; a.disps = disps; a.trg = trg;
  const long N = (long)d->B * d->H * d->W;
  hipLaunchKernelGGL(zbuf_comp_kernel<true>, dim3(grid_for(N)), dim3(TPB), 0,
                     (hipStream_t)stream, a, nullptr, g_loss, g_imgs,
                     masks ? g_masks : nullptr, g_disps);
  return rc_of_launch();
}

int lsi_disp_reg_loss_fwd(int32_t L, int32_t B, int32_t H, int32_t W,
                          int64_t sl, int64_t sb, int64_t sy, int64_t sx,
                          const float* disp, float* out2, void* ws,
                          size_t ws_bytes, lsi_stream_t stream) {
  if (L <= 0 || B <= 0 || H <= 0 || W <= 0) return LSI_EINVAL;
  if (!disp || !out2 || !ws) return LSI_ENULL;
  if (ws_bytes < lsi_loss_workspace_bytes()) return LSI_EWORKSPACE;
  DArgs a; a.L = L; a.B = B; a.H = H; a.W = W; a.sl = sl; a.sb = sb; a.sy = sy;
  a.sx = sx; a.disp = disp;
  const long N = (long)L * B * H * W;
  const int g = grid_for(N);
  hipLaunchKernelGGL(disp_reg_kernel<false>, dim3(g), dim3(TPB), 0,
                     (hipStream_t)stream, a, (double*)ws, nullptr, nullptr);
  auto inv = [](double n) { return n > 0 ? 1.0 / n : 0.0; };
  Scales sc;
  sc.s[0] = inv((double)L * B * H * (W - 2));
  sc.s[1] = inv((double)L * B * (H - 1) * (W - 1));
  sc.s[2] = sc.s[1];
  sc.s[3] = inv((double)L * B * (H - 2) * W);
  sc.s[4] = inv((double)(L - 1) * B * H * W);
  // out2[0] = dx2 + dxdy + dydx + dy2 (ldi.py:68), out2[1] = decreasing loss
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream,
                     (const double*)ws, g, 5, sc, out2, 4);
  return rc_of_launch();
}

int lsi_disp_reg_loss_bwd(int32_t L, int32_t B, int32_t H, int32_t W,
                          int64_t sl, int64_t sb, int64_t sy, int64_t sx,
                          const float* disp, const float* g2, float* g_disp,
                          lsi_stream_t stream) {
  if (L <= 0 || B <= 0 || H <= 0 || W <= 0) return LSI_EINVAL;
  if (!disp || !g2 || !g_disp) return LSI_ENULL;
  DArgs a; a.L = L; a.B = B; a.H = H; a.W = W; a.sl = sl; a.sb = sb; a.sy = sy;
  a.sx = sx; a.disp = disp;
  hipLaunchKernelGGL(disp_reg_kernel<true>, dim3(grid_for((long)L * B * H * W)),
                     dim3(TPB), 0, (hipStream_t)stream, a, nullptr, g2, g_disp);
  return rc_of_launch();
}

static int vargs_of(int32_t nl, int32_t B, int32_t Ht, int32_t Wt, int32_t H,
                    int32_t W, int32_t x_min, int32_t y_min, const float* recons,
                    const float* target, int64_t t_sb, int64_t t_sy, int64_t t_sx,
                    int64_t t_sc, VArgs* a) {
  if (nl <= 0 || B <= 0 || Ht <= 0 || Wt <= 0 || H <= 0 || W <= 0 ||
      H % Ht || W % Wt || x_min < 0 || y_min < 0 || 2 * x_min >= Wt ||
      2 * y_min >= Ht)
    return LSI_EINVAL;
  a->nl = nl; a->B = B; a->Ht = Ht; a->Wt = Wt; a->H = H; a->W = W;
  a->x_min = x_min; a->y_min = y_min; a->recons = recons; a->target = target;
  a->t_sb = t_sb; a->t_sy = t_sy; a->t_sx = t_sx; a->t_sc = t_sc;
  return LSI_OK;
}

int lsi_view_synth_loss_fwd(int32_t nl, int32_t B, int32_t Ht, int32_t Wt,
                            int32_t H, int32_t W, int32_t x_min, int32_t y_min,
                            const float* recons, const float* target,
                            int64_t t_sb, int64_t t_sy, int64_t t_sx,
                            int64_t t_sc, float* out_loss, void* ws,
                            size_t ws_bytes, lsi_stream_t stream) {
  VArgs a;
  const int rc = vargs_of(nl, B, Ht, Wt, H, W, x_min, y_min, recons, target,
                          t_sb, t_sy, t_sx, t_sc, &a);
  if (rc != LSI_OK) return rc;
  if (!recons || !target || !out_loss || !ws) return LSI_ENULL;
  if (ws_bytes < lsi_loss_workspace_bytes()) return LSI_EWORKSPACE;
  const long N = (long)B * (Ht - 2 * y_min) * (Wt - 2 * x_min);
  const int g = grid_for(N);
  hipLaunchKernelGGL(view_synth_kernel<false>, dim3(g), dim3(TPB), 0,
                     (hipStream_t)stream, a, (double*)ws, nullptr, nullptr);
  Scales sc; sc.s[0] = 1.0 / (double)N;
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream,
                     (const double*)ws, g, 1, sc, out_loss, -1);
  return rc_of_launch();
}

int lsi_view_synth_loss_bwd(int32_t nl, int32_t B, int32_t Ht, int32_t Wt,
                            int32_t H, int32_t W, int32_t x_min, int32_t y_min,
                            const float* recons, const float* target,
                            int64_t t_sb, int64_t t_sy, int64_t t_sx,
                            int64_t t_sc, const float* g_loss, float* g_recons,
                            lsi_stream_t stream) {
  VArgs a;
  const int rc = vargs_of(nl, B, Ht, Wt, H, W, x_min, y_min, recons, target,
                          t_sb, t_sy, t_sx, t_sc, &a);
  if (rc != LSI_OK) return rc;
  if (!recons || !target || !g_loss || !g_recons) return LSI_ENULL;
  hipLaunchKernelGGL(view_synth_kernel<true>, dim3(grid_for((long)B * Ht * Wt)),
                     dim3(TPB), 0, (hipStream_t)stream, a, nullptr, g_loss,
                     g_recons);
  return rc_of_launch();
}

int lsi_compose_fwd(int32_t L, int64_t N, int32_t C, const float* imgs,
                    const float* masks, const float* dmaps, int32_t soft,
                    float min_disp, float depth_softmax_temp, float* out,
                    lsi_stream_t stream) {
  if (L <= 0 || N <= 0 || C <= 0 || depth_softmax_temp == 0.0f) return LSI_EINVAL;
  if (!imgs || !masks || !dmaps || !out) return LSI_ENULL;
  CArgs a; a.L = L; a.C = C; a.N = N; a.imgs = imgs; a.masks = masks;
  a.dmaps = dmaps; a.min_disp = min_disp; a.temp = depth_softmax_temp;
  a.dmax = 0.0f; a.soft = soft; a.depth_mode = 0; a.bg_layer = 0;
  hipLaunchKernelGGL(compose_kernel, dim3(grid_for(N)), dim3(TPB), 0,
                     (hipStream_t)stream, a, out);
  return rc_of_launch();
}

int lsi_compose_depth_fwd(int32_t L, int64_t N, const float* masks,
                          const float* dmaps, int32_t bg_layer, float dmax,
                          float min_disp, float depth_softmax_temp, float* out,
                          lsi_stream_t stream) {
  if (L <= 0 || N <= 0 || depth_softmax_temp == 0.0f) return LSI_EINVAL;
  if (!masks || !dmaps || !out) return LSI_ENULL;
  CArgs a; a.L = L; a.C = 1; a.N = N; a.imgs = nullptr; a.masks = masks;
  a.dmaps = dmaps; a.min_disp = min_disp; a.temp = depth_softmax_temp;
  a.dmax = dmax; a.soft = 0; a.depth_mode = 1; a.bg_layer = bg_layer;
  hipLaunchKernelGGL(compose_kernel, dim3(grid_for(N)), dim3(TPB), 0,
                     (hipStream_t)stream, a, out);
  return rc_of_launch();
}

}  // extern "C"
