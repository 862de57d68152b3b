// Device-side building blocks shared by the LDI renderer kernels (gfx950).
//
// Floating-point contract (shared with oracle/lsi_oracle.py and
// oracle/lsi_ref_cpu.c -- it is what makes projected pixel indices bit-exact):
//   q_j = ((x*M[j][0] + y*M[j][1]) + 1*M[j][2]) + d*M[j][3], every product and
//         sum rounded to fp32 on its own (no FMA contraction), in this order
//         (helpers.py:116-137 as TF's k=4 matmul evaluates it);
//   n' = n + 1e-8f*[n==0]                                   (helpers.py:82-85)
//   u = (q0/n')*s, v = (q1/n')*s with IEEE-correct division  (ldi.py:138-139)
//   X = u-0.5f; x0 = floor(X); x1 = x0+1; clip; border masks; corner weights;
//   1e-3 clamp; idx = int(x_safe + y_safe*Wt) in fp32        (sampling.py:183-241)
// This translation unit is compiled with -ffp-contract=off and the pragma below;
// fused multiply-adds appear only where written explicitly (__fmaf_rn) in code
// that does not feed an index or a threshold decision.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace lsi {

// helpers.py:82-85
__device__ __forceinline__ float safe_den(float den) {
  return den + 1e-8f * (den == 0.0f ? 1.0f : 0.0f);
}

// IEEE-correct fp32 division (v_div_scale/v_rcp/v_div_fmas/v_div_fixup).
__device__ __forceinline__ float div_rn(float a, float b) {
  return __fdiv_rn(a, b);
}

// exp(a) to ~1.5 ulp without the libm call: a*log2(e) split so that the
// rounding of the product is compensated, then the hardware 2^x.
// Valid for |a| < ~80 (callers pass |a| <= zbuf_scale/2).
__device__ __forceinline__ float exp_accurate(float a) {
  const float L2E_HI = 1.44269502e+00f;   // fl(log2 e)
  const float L2E_LO = 1.92596299e-08f;   // log2 e - L2E_HI
  const float LN2 = 6.93147182e-01f;
  const float t = a * L2E_HI;
  float r = __fmaf_rn(a, L2E_HI, -t);     // exact residual of the product
  r = __fmaf_rn(a, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return __fmaf_rn(e, r * LN2, e);
}

// helpers.py:180-193: exp((clip(x,0,1) - 0.5)*scale) * [x > 0]
__device__ __forceinline__ float zbuffer_weight(float x, float scale) {
  const float pos = x > 0.0f ? 1.0f : 0.0f;
  float c = fminf(fmaxf(x, 0.0f), 1.0f);
  c = c - 0.5f;
  return exp_accurate(c * scale) * pos;
}

// One axis of the bilinear splat footprint (sampling.py:193-211 for x; same for
// y): given the continuous target coordinate c = u - 0.5 (or v - 0.5) returns
// the two clipped integer cells and their masked weights.
struct Axis {
  float c0s, c1s;  // clipped cell coordinates (fp32, integral)
  float w0, w1;    // weights incl. border masks
  float v0, v1;    // border-validity masks (1/0), needed by the backward
};

__device__ __forceinline__ Axis splat_axis(float c, float cmax) {
  Axis a;
  const float c0 = floorf(c);
  const float c1 = c0 + 1.0f;
  a.c0s = fminf(fmaxf(c0, 0.0f), cmax);
  a.c1s = fminf(fmaxf(c1, 0.0f), cmax);
  a.v0 = (c0 == a.c0s) ? 1.0f : 0.0f;
  a.v1 = (c1 == a.c1s) ? 1.0f : 0.0f;
  a.w0 = (c1 - c) * a.v0;
  a.w1 = (c - c0) * a.v1;
  return a;
}

__device__ __forceinline__ bool finite_f(float x) {
  return fabsf(x) < __builtin_inff();  // false for NaN and +-Inf
}

// sampling.py:218-222: weights <= 1e-3 are zeroed.
__device__ __forceinline__ float clamp_small(float w) {
  return w * (w > 1e-3f ? 1.0f : 0.0f);
}

// Row j of q = M p for p = (px, py, 1, d); see the contract above.
__device__ __forceinline__ float mrow(const float* __restrict__ m, int j,
                                      float px, float py, float d) {
  float acc = px * m[4 * j + 0] + py * m[4 * j + 1];
  acc = acc + 1.0f * m[4 * j + 2];
  acc = acc + d * m[4 * j + 3];
  return acc;
}

// Full per-source-pixel projection result.
struct Proj {
  float pw;      // zbuffer weight * mask            (ldi.py:145-146)
  float zw;      // zbuffer weight alone (backward)
  float dd;      // target-frame disparity           (ldi.py:140)
  float q0, q1, q3, nden;  // pre-division terms (backward)
  Axis ax, ay;
  float w[4];    // tl, tr, bl, br after the 1e-3 clamp
  int idx[4];    // flat x + y*Wt
  bool ok;       // false: non-finite coordinate, point dropped
};

__device__ __forceinline__ void project_px(const float* __restrict__ m,
                                           float px, float py, float d,
                                           float mk, float s, float max_disp,
                                           float zscale, int Ht, int Wt,
                                           Proj& o) {
  o.q0 = mrow(m, 0, px, py, d);
  o.q1 = mrow(m, 1, px, py, d);
  const float n = mrow(m, 2, px, py, d);
  o.q3 = mrow(m, 3, px, py, d);
  o.nden = safe_den(n);
  const float u = div_rn(o.q0, o.nden) * s;
  const float v = div_rn(o.q1, o.nden) * s;
  o.dd = div_rn(o.q3, o.nden);
  o.zw = zbuffer_weight(div_rn(o.dd, max_disp), zscale);
  o.pw = o.zw * mk;
  const float X = u - 0.5f, Y = v - 0.5f;
  o.ok = finite_f(X) && finite_f(Y);
  o.ax = splat_axis(X, (float)Wt - 1.0f);
  o.ay = splat_axis(Y, (float)Ht - 1.0f);
  const float wt = (float)Wt;
  o.w[0] = clamp_small(o.ax.w0 * o.ay.w0);
  o.w[1] = clamp_small(o.ax.w1 * o.ay.w0);
  o.w[2] = clamp_small(o.ax.w0 * o.ay.w1);
  o.w[3] = clamp_small(o.ax.w1 * o.ay.w1);
  if (o.ok) {
    o.idx[0] = (int)(o.ax.c0s + o.ay.c0s * wt);
    o.idx[1] = (int)(o.ax.c1s + o.ay.c0s * wt);
    o.idx[2] = (int)(o.ax.c0s + o.ay.c1s * wt);
    o.idx[3] = (int)(o.ax.c1s + o.ay.c1s * wt);
  } else {
    o.idx[0] = o.idx[1] = o.idx[2] = o.idx[3] = 0;
    o.w[0] = o.w[1] = o.w[2] = o.w[3] = 0.0f;
  }
}

// Streaming stores for outputs that are written once and not read again by the
// launch (rendered views, gradients): the non-temporal policy keeps them from
// lingering dirty in the XCD's L2 -- measured on MI355X: the composed view of
// config 3 (25 MB) written with plain stores costs the launch 4.8 us more
// (86.7 vs 81.9 us; write-through `sc1` stores are slower than either).
typedef float lsi_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream_f4(float* p, float x, float y, float z,
                                                float w) {
  const lsi_v4f v = {x, y, z, w};
  __builtin_nontemporal_store(v, reinterpret_cast<lsi_v4f*>(p));
}
__device__ __forceinline__ void store_stream_f1(float* p, float x) {
  __builtin_nontemporal_store(x, p);
}

// fp32 atomic add that lowers to global_atomic_add_f32 / ds_add_f32 (no CAS
// loop); the translation unit is built with -munsafe-fp-atomics.
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  atomicAdd(p, v);
}

}  // namespace lsi
