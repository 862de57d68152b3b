// lsi.geometry.sampling on MI355X (gfx950): generic bilinear splat
// (sampling.py:171-254), batched scatter-add (sampling.py:257-313) and bilinear
// gather (sampling.py:41-168), with their gradients, plus the small host-only
// entry points of the C ABI (include/lsi_hip.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"

#pragma clang fp contract(off)

using namespace lsi;

namespace {

struct Corners {
  Axis ax, ay;
  float w[4];
  int idx[4];
  bool ok;
};

// sampling.py:183-241 for one point (u, v) on an Ht x Wt canvas.
__device__ __forceinline__ void corners_of(float u, float v, int Ht, int Wt,
                                           Corners& c) {
  const float X = u - 0.5f, Y = v - 0.5f;
  c.ok = finite_f(X) && finite_f(Y);
  c.ax = splat_axis(X, (float)Wt - 1.0f);
  c.ay = splat_axis(Y, (float)Ht - 1.0f);
  const float wt = (float)Wt;
  c.w[0] = clamp_small(c.ax.w0 * c.ay.w0);
  c.w[1] = clamp_small(c.ax.w1 * c.ay.w0);
  c.w[2] = clamp_small(c.ax.w0 * c.ay.w1);
  c.w[3] = clamp_small(c.ax.w1 * c.ay.w1);
  if (c.ok) {
    c.idx[0] = (int)(c.ax.c0s + c.ay.c0s * wt);
    c.idx[1] = (int)(c.ax.c1s + c.ay.c0s * wt);
    c.idx[2] = (int)(c.ax.c0s + c.ay.c1s * wt);
    c.idx[3] = (int)(c.ax.c1s + c.ay.c1s * wt);
  } else {
    c.idx[0] = c.idx[1] = c.idx[2] = c.idx[3] = 0;
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0.0f;
  }
}

__global__ __launch_bounds__(256) void splat_generic_kernel(
    int Ns, int C, int Ht, int Wt, const float* __restrict__ src,
    const float* __restrict__ coords, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Ns) return;
  const size_t si = (size_t)b * Ns + i;
  Corners c;
  corners_of(coords[2 * si], coords[2 * si + 1], Ht, Wt, c);
  if (!c.ok) return;
  float* ob = out + (size_t)b * Ht * Wt * C;
  const float* sp = src + si * C;
  for (int ch = 0; ch < C; ++ch) {
    const float v = sp[ch];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float upd = v * c.w[k];
      if (upd != 0.0f) atomic_add_f32(ob + (size_t)c.idx[k] * C + ch, upd);
    }
  }
}

__global__ __launch_bounds__(256) void splat_generic_bwd_kernel(
    int Ns, int C, int Ht, int Wt, const float* __restrict__ src,
    const float* __restrict__ coords, const float* __restrict__ g_out,
    float* __restrict__ g_src, float* __restrict__ g_coords) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Ns) return;
  const size_t si = (size_t)b * Ns + i;
  Corners c;
  corners_of(coords[2 * si], coords[2 * si + 1], Ht, Wt, c);
  const float* gb = g_out + (size_t)b * Ht * Wt * C;
  float gwk[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < C; ++ch) {
    const float v = src[si * C + ch];
    float gs = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c.w[k] == 0.0f) continue;
      const float g = gb[(size_t)c.idx[k] * C + ch];
      gs += c.w[k] * g;
      gwk[k] += v * g;
    }
    if (g_src) g_src[si * C + ch] = gs;
  }
  if (g_coords) {
    const float gX = -c.ax.v0 * (gwk[0] * c.ay.w0 + gwk[2] * c.ay.w1) +
                     c.ax.v1 * (gwk[1] * c.ay.w0 + gwk[3] * c.ay.w1);
    const float gY = -c.ay.v0 * (gwk[0] * c.ax.w0 + gwk[1] * c.ax.w1) +
                     c.ay.v1 * (gwk[2] * c.ax.w0 + gwk[3] * c.ax.w1);
    g_coords[2 * si] = c.ok ? gX : 0.0f;
    g_coords[2 * si + 1] = c.ok ? gY : 0.0f;
  }
}

__global__ __launch_bounds__(256) void scatter_add_kernel(
    int64_t P, int64_t N, const int32_t* __restrict__ idx,
    const float* __restrict__ upd, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int32_t j = idx[b * N + i];
  if (j < 0 || j >= P) return;
  atomic_add_f32(out + b * P + j, upd[b * N + i]);
}

// sampling.py:54-107: the four taps of one sampling point on an Hs x Ws image.
struct Taps {
  float wx0, wx1, wy0, wy1;  // un-masked interpolation weights
  float vx0, vx1, vy0, vy1;  // validity masks
  int i00, i01, i10, i11;    // flat x + y*Ws (00: x0,y0  01: x0,y1  10: x1,y0)
  bool ok;
};

__device__ __forceinline__ void taps_of(float u, float v, int Hs, int Ws,
                                        Taps& t) {
  const float x = u - 0.5f, y = v - 0.5f;
  t.ok = finite_f(x) && finite_f(y);
  const float x0 = floorf(x), x1 = x0 + 1.0f, y0 = floorf(y), y1 = y0 + 1.0f;
  const float xm = (float)(Ws - 1), ym = (float)(Hs - 1);
  const float x0s = fminf(fmaxf(x0, 0.f), xm), x1s = fminf(fmaxf(x1, 0.f), xm);
  const float y0s = fminf(fmaxf(y0, 0.f), ym), y1s = fminf(fmaxf(y1, 0.f), ym);
  t.wx0 = x1 - x; t.wx1 = x - x0; t.wy0 = y1 - y; t.wy1 = y - y0;
  t.vx0 = x0 == x0s ? 1.f : 0.f; t.vx1 = x1 == x1s ? 1.f : 0.f;
  t.vy0 = y0 == y0s ? 1.f : 0.f; t.vy1 = y1 == y1s ? 1.f : 0.f;
  const float w = (float)Ws;
  if (t.ok) {
    t.i00 = (int)(x0s + y0s * w); t.i01 = (int)(x0s + y1s * w);
    t.i10 = (int)(x1s + y0s * w); t.i11 = (int)(x1s + y1s * w);
  } else {
    t.i00 = t.i01 = t.i10 = t.i11 = 0;
  }
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(
    int Hs, int Ws, int C, int Nt, const float* __restrict__ imgs,
    const float* __restrict__ coords, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nt) return;
  const size_t ti = (size_t)b * Nt + i;
  Taps t;
  taps_of(coords[2 * ti], coords[2 * ti + 1], Hs, Ws, t);
  const float* ib = imgs + (size_t)b * Hs * Ws * C;
  // sampling.py:118-123: valid_x * valid_y * wt_x * wt_y * im, summed in the
  // order 00, 01, 10, 11.
  const float c00 = t.vx0 * t.vy0 * t.wx0 * t.wy0;
  const float c01 = t.vx0 * t.vy1 * t.wx0 * t.wy1;
  const float c10 = t.vx1 * t.vy0 * t.wx1 * t.wy0;
  const float c11 = t.vx1 * t.vy1 * t.wx1 * t.wy1;
  for (int ch = 0; ch < C; ++ch) {
    float o = 0.0f;
    if (t.ok) {
      o = c00 * ib[(size_t)t.i00 * C + ch];
      o = o + c01 * ib[(size_t)t.i01 * C + ch];
      o = o + c10 * ib[(size_t)t.i10 * C + ch];
      o = o + c11 * ib[(size_t)t.i11 * C + ch];
    }
    out[ti * C + ch] = o;
  }
}

// sampling.py:124-130 (compose=False): the four border-masked taps and the four
// un-masked weights, tap order (x0,y0), (x0,y1), (x1,y0), (x1,y1).
__global__ __launch_bounds__(256) void bilinear_taps_kernel(
    int Hs, int Ws, int C, int Nt, int B, const float* __restrict__ imgs,
    const float* __restrict__ coords, float* __restrict__ taps,
    float* __restrict__ wts) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nt) return;
  const size_t ti = (size_t)b * Nt + i;
  const size_t N = (size_t)B * Nt;  // elements of one tap / weight plane
  Taps t;
  taps_of(coords[2 * ti], coords[2 * ti + 1], Hs, Ws, t);
  const float* ib = imgs + (size_t)b * Hs * Ws * C;
  const float m[4] = {t.vx0 * t.vy0, t.vx0 * t.vy1, t.vx1 * t.vy0, t.vx1 * t.vy1};
  const int idx[4] = {t.i00, t.i01, t.i10, t.i11};
  wts[0 * N + ti] = t.wx0 * t.wy0;
  wts[1 * N + ti] = t.wx0 * t.wy1;
  wts[2 * N + ti] = t.wx1 * t.wy0;
  wts[3 * N + ti] = t.wx1 * t.wy1;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    for (int ch = 0; ch < C; ++ch)
      taps[(k * N + ti) * C + ch] = t.ok ? m[k] * ib[(size_t)idx[k] * C + ch] : 0.0f;
}

__global__ __launch_bounds__(256) void bilinear_bwd_kernel(
    int Hs, int Ws, int C, int Nt, const float* __restrict__ imgs,
    const float* __restrict__ coords, const float* __restrict__ g_out,
    float* __restrict__ g_imgs, float* __restrict__ g_coords) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nt) return;
  const size_t ti = (size_t)b * Nt + i;
  Taps t;
  taps_of(coords[2 * ti], coords[2 * ti + 1], Hs, Ws, t);
  if (!t.ok) {
    if (g_coords) { g_coords[2 * ti] = 0.f; g_coords[2 * ti + 1] = 0.f; }
    return;
  }
  const float* ib = imgs + (size_t)b * Hs * Ws * C;
  float* gb = g_imgs ? g_imgs + (size_t)b * Hs * Ws * C : nullptr;
  const float m00 = t.vx0 * t.vy0, m01 = t.vx0 * t.vy1, m10 = t.vx1 * t.vy0,
              m11 = t.vx1 * t.vy1;
  float gx = 0.f, gy = 0.f;
  for (int ch = 0; ch < C; ++ch) {
    const float g = g_out[ti * C + ch];
    const float a00 = ib[(size_t)t.i00 * C + ch], a01 = ib[(size_t)t.i01 * C + ch],
                a10 = ib[(size_t)t.i10 * C + ch], a11 = ib[(size_t)t.i11 * C + ch];
    if (gb) {
      const float u00 = g * m00 * t.wx0 * t.wy0, u01 = g * m01 * t.wx0 * t.wy1,
                  u10 = g * m10 * t.wx1 * t.wy0, u11 = g * m11 * t.wx1 * t.wy1;
      if (u00 != 0.f) atomic_add_f32(gb + (size_t)t.i00 * C + ch, u00);
      if (u01 != 0.f) atomic_add_f32(gb + (size_t)t.i01 * C + ch, u01);
      if (u10 != 0.f) atomic_add_f32(gb + (size_t)t.i10 * C + ch, u10);
      if (u11 != 0.f) atomic_add_f32(gb + (size_t)t.i11 * C + ch, u11);
    }
    // d wt_x0/dx = -1, d wt_x1/dx = +1 (floor has zero gradient)
    gx += g * (-(m00 * t.wy0 * a00 + m01 * t.wy1 * a01) +
               (m10 * t.wy0 * a10 + m11 * t.wy1 * a11));
    gy += g * (-(m00 * t.wx0 * a00 + m10 * t.wx1 * a10) +
               (m01 * t.wx0 * a01 + m11 * t.wx1 * a11));
  }
  if (g_coords) { g_coords[2 * ti] = gx; g_coords[2 * ti + 1] = gy; }
}

// TF autodiff of sampling.py:124-130 (compose=False).  The taps are gathers times
// a 0 / 1 mask: their gradient is a scatter-add into the image (floor / clip /
// equal carry no gradient to the coordinates); the weights are products of
// (x1 - x), (x - x0), (y1 - y), (y - y0) with d/dx = -1, +1 and d/dy = -1, +1.
__global__ __launch_bounds__(256) void bilinear_taps_bwd_kernel(
    int Hs, int Ws, int C, int Nt, int B, const float* __restrict__ coords,
    const float* __restrict__ g_taps, const float* __restrict__ g_wts,
    float* __restrict__ g_imgs, float* __restrict__ g_coords) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nt) return;
  const size_t ti = (size_t)b * Nt + i;
  const size_t N = (size_t)B * Nt;
  Taps t;
  taps_of(coords[2 * ti], coords[2 * ti + 1], Hs, Ws, t);
  if (g_coords) {
    float gx = 0.f, gy = 0.f;
    if (g_wts) {
      const float g0 = g_wts[0 * N + ti], g1 = g_wts[1 * N + ti], g2 = g_wts[2 * N + ti],
                  g3 = g_wts[3 * N + ti];
      // wts: wx0 wy0, wx0 wy1, wx1 wy0, wx1 wy1
      gx = (g2 * t.wy0 + g3 * t.wy1) - (g0 * t.wy0 + g1 * t.wy1);
      gy = (g1 * t.wx0 + g3 * t.wx1) - (g0 * t.wx0 + g2 * t.wx1);
    }
    g_coords[2 * ti] = gx;
    g_coords[2 * ti + 1] = gy;
  }
  if (!g_imgs || !g_taps || !t.ok) return;
  float* gb = g_imgs + (size_t)b * Hs * Ws * C;
  const float m[4] = {t.vx0 * t.vy0, t.vx0 * t.vy1, t.vx1 * t.vy0, t.vx1 * t.vy1};
  const int idx[4] = {t.i00, t.i01, t.i10, t.i11};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (m[k] == 0.f) continue;
    for (int ch = 0; ch < C; ++ch) {
      const float u = g_taps[(k * N + ti) * C + ch];
      if (u != 0.f) atomic_add_f32(gb + (size_t)idx[k] * C + ch, u);
    }
  }
}

inline int launch_rc() {
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

}  // namespace

extern "C" {

int lsi_version(void) { return LSI_VERSION; }

const char* lsi_strerror(int code) {
  switch (code) {
    case LSI_OK: return "ok";
    case LSI_EINVAL: return "invalid shape, stride or flag combination";
    case LSI_ENULL: return "required pointer is NULL";
    case LSI_EWORKSPACE: return "workspace too small";
    case LSI_ELAUNCH: return "HIP kernel launch failed";
    case LSI_EUNSUPPORTED: return "not implemented in this build";
    default: return "unknown lsi error code";
  }
}

float lsi_bg_weight(double bg_layer_disp, double max_disp, double zbuf_scale) {
  // ldi.py:115-116 + helpers.py:180-193 on the host, fp32 after the division.
  float x = (float)(bg_layer_disp / max_disp);
  const float pos = x > 0.0f ? 1.0f : 0.0f;
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  x = x - 0.5f;
  return expf(x * (float)zbuf_scale) * pos;
}

int lsi_splat_generic(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                      int32_t Wt, const float* src, const float* coords,
                      float* out, lsi_stream_t stream) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || Ht <= 0 || Wt <= 0 ||
      B > 65535 || (int64_t)Ht * Wt >= (1 << 24))
    return LSI_EINVAL;
  if (!src || !coords || !out) return LSI_ENULL;
  const int Ns = Hs * Ws;
  hipLaunchKernelGGL(splat_generic_kernel, dim3((Ns + 255) / 256, B), dim3(256),
                     0, (hipStream_t)stream, Ns, C, Ht, Wt, src, coords, out);
  return launch_rc();
}

int lsi_splat_generic_bwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                          int32_t Ht, int32_t Wt, const float* src,
                          const float* coords, const float* g_out, float* g_src,
                          float* g_coords, lsi_stream_t stream) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || Ht <= 0 || Wt <= 0 ||
      B > 65535 || (int64_t)Ht * Wt >= (1 << 24))
    return LSI_EINVAL;
  if (!src || !coords || !g_out) return LSI_ENULL;
  const int Ns = Hs * Ws;
  hipLaunchKernelGGL(splat_generic_bwd_kernel, dim3((Ns + 255) / 256, B),
                     dim3(256), 0, (hipStream_t)stream, Ns, C, Ht, Wt, src,
                     coords, g_out, g_src, g_coords);
  return launch_rc();
}

int lsi_scatter_add(int32_t B, int64_t P, int64_t N, const int32_t* idx,
                    const float* upd, float* out, lsi_stream_t stream) {
  if (B <= 0 || P <= 0 || N < 0 || B > 65535) return LSI_EINVAL;
  if (N == 0) return LSI_OK;
  if (!idx || !upd || !out) return LSI_ENULL;
  hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)((N + 255) / 256), B),
                     dim3(256), 0, (hipStream_t)stream, P, N, idx, upd, out);
  return launch_rc();
}

int lsi_bilinear_fwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                     int32_t Wt, const float* imgs, const float* coords,
                     float* out, lsi_stream_t stream) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || Ht <= 0 || Wt <= 0 ||
      B > 65535 || (int64_t)Hs * Ws >= (1 << 24))
    return LSI_EINVAL;
  if (!imgs || !coords || !out) return LSI_ENULL;
  const int Nt = Ht * Wt;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3((Nt + 255) / 256, B), dim3(256),
                     0, (hipStream_t)stream, Hs, Ws, C, Nt, imgs, coords, out);
  return launch_rc();
}

int lsi_bilinear_taps(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                      int32_t Wt, const float* imgs, const float* coords,
                      float* taps, float* wts, lsi_stream_t stream) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || Ht <= 0 || Wt <= 0 ||
      B > 65535 || (int64_t)Hs * Ws >= (1 << 24))
    return LSI_EINVAL;
  if (!imgs || !coords || !taps || !wts) return LSI_ENULL;
  const int Nt = Ht * Wt;
  hipLaunchKernelGGL(bilinear_taps_kernel, dim3((Nt + 255) / 256, B), dim3(256),
                     0, (hipStream_t)stream, Hs, Ws, C, Nt, B, imgs, coords, taps,
                     wts);
  return launch_rc();
}

int lsi_bilinear_taps_bwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                          int32_t Wt, const float* coords, const float* g_taps,
                          const float* g_wts, float* g_imgs, float* g_coords,
                          lsi_stream_t stream) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || Ht <= 0 || Wt <= 0 ||
      B > 65535 || (int64_t)Hs * Ws >= (1 << 24))
    return LSI_EINVAL;
  if (!coords) return LSI_ENULL;
  if (g_imgs && !g_taps) return LSI_ENULL;
  const int Nt = Ht * Wt;
  hipLaunchKernelGGL(bilinear_taps_bwd_kernel, dim3((Nt + 255) / 256, B), dim3(256),
                     0, (hipStream_t)stream, Hs, Ws, C, Nt, B, coords, g_taps, g_wts,
                     g_imgs, g_coords);
  return launch_rc();
}

int lsi_bilinear_bwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                     int32_t Wt, const float* imgs, const float* coords,
                     const float* g_out, float* g_imgs, float* g_coords,
                     lsi_stream_t stream) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || Ht <= 0 || Wt <= 0 ||
      B > 65535 || (int64_t)Hs * Ws >= (1 << 24))
    return LSI_EINVAL;
  if (!imgs || !coords || !g_out) return LSI_ENULL;
  const int Nt = Ht * Wt;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((Nt + 255) / 256, B), dim3(256),
                     0, (hipStream_t)stream, Hs, Ws, C, Nt, imgs, coords, g_out,
                     g_imgs, g_coords);
  return launch_rc();
}

}  // extern "C"
