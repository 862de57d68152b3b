// Host-side helpers (no kernels): launch bookkeeping shared by the kernel
// families, and the projection matrices of
// lsi/geometry/projection.py computed for a whole batch in one call.  The
// Python mirror (lsi/geometry/projection.py of this repo) does the same
// arithmetic with a dozen small torch ops (~300 us for 32 cameras, three times
// the renderer's kernel); forward_splat calls this instead when the cameras
// are on the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"
#include "lsi_splat_internal.h"

#pragma clang fp contract(off)

namespace {

// 3x3 inverse in fp64 (adjugate / determinant), rounded ONCE to fp32: what
// projection._inv3 pins (a correctly rounded inverse; fp32 LU routines of
// different libraries differ in the last ulp).
void inv3(const float* k, float* out) {
  double a[9];
  for (int i = 0; i < 9; ++i) a[i] = (double)k[i];
  const double c00 = a[4] * a[8] - a[5] * a[7];
  const double c01 = a[5] * a[6] - a[3] * a[8];
  const double c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const double r = 1.0 / det;
  const double inv[9] = {
      c00 * r, (a[2] * a[7] - a[1] * a[8]) * r, (a[1] * a[5] - a[2] * a[4]) * r,
      c01 * r, (a[0] * a[8] - a[2] * a[6]) * r, (a[2] * a[3] - a[0] * a[5]) * r,
      c02 * r, (a[1] * a[6] - a[0] * a[7]) * r, (a[0] * a[4] - a[1] * a[3]) * r};
  for (int i = 0; i < 9; ++i) out[i] = (float)inv[i];
}

// 4x4 product accumulated sequentially over k, every multiply and add rounded
// to fp32 on its own (helpers.seq_matmul; TF-1.4's small matmul order)
void matmul4_seq(const float* a, const float* b, float* out) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = a[4 * i + 0] * b[0 + j];
      for (int k = 1; k < 4; ++k) {
        const float p = a[4 * i + k] * b[4 * k + j];
        acc = acc + p;
      }
      out[4 * i + j] = acc;
    }
}

void pad_intrinsic(const float* k, float* out) {  // projection.py:27-46
  for (int i = 0; i < 16; ++i) out[i] = 0.0f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[4 * i + j] = k[3 * i + j];
  out[15] = 1.0f;
}

void pad_extrinsic(const float* rot, const float* t, float* out) {  // :49-68
  for (int i = 0; i < 16; ++i) out[i] = 0.0f;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out[4 * i + j] = rot[3 * i + j];
    out[4 * i + 3] = t[i];
  }
  out[15] = 1.0f;
}

}  // namespace

extern "C" int lsi_projection_matrices(int32_t B, const float* k_s, const float* k_t,
                                       const float* rot, const float* t,
                                       int32_t inverse, float* M) {
  if (B <= 0) return LSI_EINVAL;
  if (!k_s || !k_t || !rot || !t || !M) return LSI_ENULL;
  for (int b = 0; b < B; ++b) {
    const float* ks = k_s + 9 * b;
    const float* kt = k_t + 9 * b;
    const float* r = rot + 9 * b;
    const float* tt = t + 3 * b;
    float kinv[9], A[16], E[16], Ki[16], EK[16];
    if (!inverse) {
      // projection.py:71-86: pad(K_t) ([R t; 0 1] pad(K_s^-1))
      inv3(ks, kinv);
      pad_intrinsic(kt, A);
      pad_extrinsic(r, tt, E);
    } else {
      // projection.py:89-106: pad(K_s) ([R^T  -R^T t; 0 1] pad(K_t^-1))
      inv3(kt, kinv);
      pad_intrinsic(ks, A);
      float rt[9], ti[3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rt[3 * i + j] = r[3 * j + i];
      for (int i = 0; i < 3; ++i) {  // -1 * seq_matmul(R^T, t)
        float acc = rt[3 * i + 0] * tt[0];
        acc = acc + rt[3 * i + 1] * tt[1];
        acc = acc + rt[3 * i + 2] * tt[2];
        ti[i] = -1.0f * acc;
      }
      pad_extrinsic(rt, ti, E);
    }
    pad_intrinsic(kinv, Ki);
    matmul4_seq(E, Ki, EK);
    matmul4_seq(A, EK, M + 16 * b);
  }
  return LSI_OK;
}

int lsi_ensure_dynamic_lds(const void* fn, size_t bytes) {
  struct Granted { int dev; const void* fn; size_t bytes; };
  static std::mutex mu;
  static std::vector<Granted> granted;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return LSI_ELAUNCH;
  std::lock_guard<std::mutex> g(mu);
  for (Granted& e : granted)
    if (e.dev == dev && e.fn == fn) {
      if (e.bytes >= bytes) return LSI_OK;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes) != hipSuccess)
        return LSI_ELAUNCH;
      e.bytes = bytes;
      return LSI_OK;
    }
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)bytes) != hipSuccess)
    return LSI_ELAUNCH;
  granted.push_back(Granted{dev, fn, bytes});
  return LSI_OK;
}

