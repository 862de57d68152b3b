// Fused batch norm (batch statistics) + beta + ReLU for channels-last
// activations -- slim.batch_norm(center=True, scale=False, epsilon=1e-3,
// is_training=True) followed by tf.nn.relu, the normaliser / activation of every
// slim.conv2d / conv2d_transpose of the reference's networks (nets.py:44-67,
// 95-111, 265-347).
//
// Why: in the bf16 training step the convolutions (MIOpen implicit GEMM) are
// followed by three MIOpen batch-norm kernels, a ReLU kernel and dtype casts per
// layer, and by five more in the backward (profiles/r02/train_step_summary.json:
// the BN / elementwise chain is ~25 % of the step's kernel time).  Here the
// forward is two passes over the activation (statistics; normalise + ReLU +
// store in the input's dtype) and the backward two (reductions; dx), each a
// pure HBM stream of 16-byte accesses.
//
// Layout: x is N*H*W pixels x C channels, C innermost (torch channels_last),
// fp32 or bf16; C a multiple of the 16-byte vector (4 fp32 / 8 bf16) with
// C / vector a power of two <= 256.  Statistics are accumulated in fp32 per
// thread (around a per-channel shift: the channel's first value, so that
// E[(x-k)^2] - E[x-k]^2 does not cancel), per workgroup in fp32 through LDS,
// across workgroups by fp32 device-scope atomic adds into 2*C accumulators; the
// workgroup that arrives LAST (a device-scope counter) turns the totals into
// the per-channel constants the second pass reads and leaves accumulators and
// counter zero for the next launch.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/lsi_hip.h"
#include "lsi_splat_internal.h"

namespace {

constexpr int BN_THREADS = 256;

template <bool BF16> struct Vec;
template <> struct Vec<false> {
  static constexpr int N = 4;
  typedef float4 raw;
  __device__ static void load(const void* p, long i, float* v) {
    const float4 r = reinterpret_cast<const float4*>(p)[i];
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  __device__ static void store(void* p, long i, const float* v) {
    reinterpret_cast<float4*>(p)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec<true> {
  static constexpr int N = 8;
  __device__ static void load(const void* p, long i, float* v) {
    const uint4 r = reinterpret_cast<const uint4*>(p)[i];
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = __uint_as_float(w[k] << 16);
      v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
  }
  __device__ static unsigned rne(float f) {  // fp32 -> bf16 bits, round to nearest even
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
  }
  __device__ static void store(void* p, long i, const float* v) {
    uint4 r;
    r.x = rne(v[0]) | (rne(v[1]) << 16);
    r.y = rne(v[2]) | (rne(v[3]) << 16);
    r.z = rne(v[4]) | (rne(v[5]) << 16);
    r.w = rne(v[6]) | (rne(v[7]) << 16);
    reinterpret_cast<uint4*>(p)[i] = r;
  }
};

// workspace layout per group (floats): [0] arrival counter (int), [16, 16 + 4096)
// the accumulators -- both zero between launches --, then 2*C constants of the
// second pass and (backward) the group's C sums of dz
constexpr int WS_ACC = LSI_BN_WS_ACC, WS_CONST = LSI_BN_WS_CONST, WS_STRIDE = LSI_BN_WS_STRIDE;

// Sums of the per-thread accumulators over the threads that hold the same
// channels (tid % lpp), added to the 2*C global accumulators; returns true in
// the workgroup that arrived last, with the totals in `tot` (LDS, [2*C]).
template <int NV>
__device__ bool reduce_all(const float* a0, const float* a1, int lpp, int C,
                           float* ws, float* tot) {
  __shared__ float red[BN_THREADS * 2 * 8];
  __shared__ int last;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    red[tid * 2 * NV + k] = a0[k];
    red[tid * 2 * NV + NV + k] = a1[k];
  }
  __syncthreads();
  float* acc = ws + WS_ACC;
  for (int t = tid; t < 2 * C; t += BN_THREADS) {
    const int q = t / C, c = t - q * C;
    const int lane = c / NV, k = c - lane * NV;
    float s = 0.0f;
    for (int r = lane; r < BN_THREADS; r += lpp) s += red[r * 2 * NV + q * NV + k];
    __hip_atomic_fetch_add(acc + t, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_s_waitcnt(0);   // the adds are performed ...
  __syncthreads();
  if (tid == 0)                    // ... before this workgroup is counted
    last = __hip_atomic_fetch_add(reinterpret_cast<int*>(ws), 1, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return false;
  // A convolution's statistics nobody consumed (lsi_conv2d_*_bnstats without the
  // lsi_bn_relu_norm behind it) have left the group's tag and sums in slots this
  // kernel never clears: the totals are not this tensor's.  NaN out (loud), and
  // clean up for the calls that follow.
  const bool dirty = __hip_atomic_load(reinterpret_cast<int*>(ws) + LSI_BN_WS_TAG,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  for (int t = tid; t < 2 * C; t += BN_THREADS) {
    // (read and clear in one atomic: the accumulators are zero again)
    const float v = __hip_atomic_exchange(acc + t, 0.0f, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
    tot[t] = dirty ? __builtin_nanf("") : v;
  }
  if (dirty) {
    for (int t = 2 * C + tid; t < WS_CONST - WS_ACC; t += BN_THREADS)
      __hip_atomic_store(acc + t, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0)
      __hip_atomic_store(reinterpret_cast<int*>(ws) + LSI_BN_WS_TAG, 0, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0)
    __hip_atomic_store(reinterpret_cast<int*>(ws), 0, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return true;
}

// group `blockIdx.y` of a launch: its slice of every per-group array
template <bool BF16>
__device__ __forceinline__ const void* grp_in(const void* p, long npix, int C) {
  return static_cast<const char*>(p) + (size_t)blockIdx.y * npix * C * (BF16 ? 2 : 4);
}
template <bool BF16>
__device__ __forceinline__ void* grp_out(void* p, long npix, int C) {
  return static_cast<char*>(p) + (size_t)blockIdx.y * npix * C * (BF16 ? 2 : 4);
}

// pass 2 of the forward: y = relu(x * a + b), a = rstd, b = beta - mean * rstd
template <bool BF16>
__device__ __forceinline__ void norm_pass(const void* __restrict__ x, void* __restrict__ y,
                                          const float* ab, long npix, int C, int relu) {
  constexpr int NV = Vec<BF16>::N;
  const int lpp = C / NV, rows = BN_THREADS / lpp;
  const int lane = threadIdx.x % lpp, row = threadIdx.x / lpp;
  float a[NV], b[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    a[k] = ab[lane * NV + k];
    b[k] = ab[C + lane * NV + k];
  }
  const long stride = (long)gridDim.x * rows;
  long p = (long)blockIdx.x * rows + row;
  for (; p + 3 * stride < npix; p += 4 * stride) {
    float v[4][NV];
#pragma unroll
    for (int u = 0; u < 4; ++u) Vec<BF16>::load(x, (p + u * stride) * lpp + lane, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float z = __fmaf_rn(v[u][k], a[k], b[k]);
        v[u][k] = (relu && z < 0.0f) ? 0.0f : z;   // (not fmaxf: NaN constants must show)
      }
      Vec<BF16>::store(y, (p + u * stride) * lpp + lane, v[u]);
    }
  }
  for (; p < npix; p += stride) {
    float v[NV];
    Vec<BF16>::load(x, p * lpp + lane, v);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float z = __fmaf_rn(v[k], a[k], b[k]);
      v[k] = (relu && z < 0.0f) ? 0.0f : z;
    }
    Vec<BF16>::store(y, p * lpp + lane, v);
  }
}

// pass 1 of the forward: sums of (x - k) and (x - k)^2 per channel, k = the
// channel's value at pixel 0
template <bool BF16>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(
    const void* __restrict__ x_, const float* __restrict__ beta,
    float* __restrict__ ws_, float* __restrict__ mean_rstd_, long npix, int C,
    float eps) {
  constexpr int NV = Vec<BF16>::N;
  const void* x = grp_in<BF16>(x_, npix, C);
  float* ws = ws_ + (size_t)blockIdx.y * WS_STRIDE;
  float* mean_rstd = mean_rstd_ + (size_t)blockIdx.y * 2 * C;
  const int lpp = C / NV;                      // lanes per pixel
  const int rows = BN_THREADS / lpp;           // pixels per workgroup step
  const int lane = threadIdx.x % lpp, row = threadIdx.x / lpp;
  float kk[NV], s[NV], q[NV];
  Vec<BF16>::load(x, lane, kk);
#pragma unroll
  for (int k = 0; k < NV; ++k) { s[k] = 0.f; q[k] = 0.f; }
  const long stride = (long)gridDim.x * rows;
  long p = (long)blockIdx.x * rows + row;
  for (; p + 7 * stride < npix; p += 8 * stride) {   // eight loads in flight
    float v[8][NV];
#pragma unroll
    for (int u = 0; u < 8; ++u) Vec<BF16>::load(x, (p + u * stride) * lpp + lane, v[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float d = v[u][k] - kk[k];
        s[k] += d; q[k] = __fmaf_rn(d, d, q[k]);
      }
  }
  for (; p < npix; p += stride) {
    float v[NV];
    Vec<BF16>::load(x, p * lpp + lane, v);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float d = v[k] - kk[k];
      s[k] += d; q[k] = __fmaf_rn(d, d, q[k]);
    }
  }
  __shared__ float tot[2 * 2048];
  if (reduce_all<NV>(s, q, lpp, C, ws, tot)) {
    // the last workgroup: mean, rstd, and y = x * a + b
    for (int c = threadIdx.x; c < C; c += BN_THREADS) {
      float k0[NV];
      Vec<BF16>::load(x, c / NV, k0);
      const double shift = (double)k0[c % NV];
      const double m1 = (double)tot[c] / (double)npix;
      double var = (double)tot[C + c] / (double)npix - m1 * m1;   // biased (tf.nn.moments)
      if (var < 0.0) var = 0.0;
      const double mean = shift + m1;
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      mean_rstd[c] = (float)mean;
      mean_rstd[C + c] = rstd;
      tot[c] = rstd;
      tot[C + c] = beta[c] - (float)mean * rstd;
    }
    __syncthreads();
    float* ab = ws + WS_CONST;
    for (int t = threadIdx.x; t < 2 * C; t += BN_THREADS) ab[t] = tot[t];
  }
}

template <bool BF16>
__global__ __launch_bounds__(BN_THREADS) void bn_norm_kernel(
    const void* __restrict__ x, void* __restrict__ y, const float* __restrict__ ws,
    long npix, int C, int relu) {
  norm_pass<BF16>(grp_in<BF16>(x, npix, C), grp_out<BF16>(y, npix, C),
                         ws + (size_t)blockIdx.y * WS_STRIDE + WS_CONST, npix, C, relu);
}

// The second pass behind a producer that left plain sums of x and x * x in the
// group's accumulator slots (the convolution kernels' epilogue,
// lsi_conv_igemm.hip): every workgroup folds the slots and forms the constants
// itself (2 C * slots floats out of L2: less than its first pixel step), the
// group's first workgroup leaves mean / rstd for the backward, and the
// workgroup that finishes last -- all of them have read the sums by then --
// clears the accumulators for the next producer.
template <bool BF16>
__global__ __launch_bounds__(BN_THREADS) void bn_norm_sums_kernel(
    const void* __restrict__ x, void* __restrict__ y, const float* __restrict__ beta,
    float* __restrict__ ws_, float* __restrict__ mean_rstd_, long npix, int C, int relu,
    float eps, int ns) {
  __shared__ float ab[2 * 2048];
  __shared__ float red[BN_THREADS];
  float* ws = ws_ + (size_t)blockIdx.y * WS_STRIDE;
  // (plain loads: the sums were added by the previous kernel on the stream; all
  // of a thread's loads are in flight together)
  const float* acc = ws + WS_ACC;
  const int n2 = 2 * C;
  // the producer's tag: the sums are those of a C-channel tensor in gridDim.y
  // groups, left by the kernel before this one -- anything else: NaN constants
  const bool handed = reinterpret_cast<const int*>(ws)[LSI_BN_WS_TAG] ==
                      LSI_BN_TAG(C, (int)gridDim.y);
  if (n2 <= BN_THREADS) {   // several threads per entry, a few slots each
    const int t = threadIdx.x % n2, s0 = threadIdx.x / n2, sstep = BN_THREADS / n2;
    float v = 0.0f;
    for (int sl = s0; sl < ns; sl += sstep) v += acc[sl * n2 + t];
    red[threadIdx.x] = v;
    __syncthreads();
    if ((int)threadIdx.x < n2) {
      v = 0.0f;
      for (int k = 0; k < sstep; ++k) v += red[k * n2 + threadIdx.x];
      ab[threadIdx.x] = v;
    }
  } else {
    for (int t = threadIdx.x; t < n2; t += BN_THREADS) {
      float v = 0.0f;
      for (int sl = 0; sl < ns; ++sl) v += acc[sl * n2 + t];
      ab[t] = v;
    }
  }
  __syncthreads();
  // This workgroup has read the sums: it is counted now, and looks at what the
  // counter returned only when it is done (the round trip hides behind the pass).
  int ticket = 0;
  if (threadIdx.x == 0)
    ticket = __hip_atomic_fetch_add(reinterpret_cast<int*>(ws_), 1, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
  __shared__ float cst[2 * 2048];
  const float inv_n = (float)(1.0 / (double)npix);
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    const float mean = ab[c] * inv_n;
    // biased variance (tf.nn.moments); fp32: the sums are fp32
    const float var = fmaxf(__fmaf_rn(-mean, mean, ab[C + c] * inv_n), 0.0f);
    const float rstd = handed ? 1.0f / sqrtf(var + eps) : __builtin_nanf("");
    if (blockIdx.x == 0) {
      float* mr = mean_rstd_ + (size_t)blockIdx.y * 2 * C;
      mr[c] = mean;
      mr[C + c] = rstd;
    }
    cst[c] = rstd;
    cst[C + c] = __fmaf_rn(-mean, rstd, beta[c]);
  }
  __syncthreads();
  norm_pass<BF16>(grp_in<BF16>(x, npix, C), grp_out<BF16>(y, npix, C), cst, npix, C, relu);
  // the workgroup that was counted last clears every group's accumulators (all
  // the others had read them when they were counted)
  __shared__ int last;
  if (threadIdx.x == 0) last = ticket == (int)(gridDim.x * gridDim.y) - 1;
  __syncthreads();
  if (last) {
    for (unsigned g = 0; g < gridDim.y; ++g) {
      float* a = ws_ + (size_t)g * WS_STRIDE + WS_ACC;
      // (a hand-over that was not this call's: whatever its producer's layout was)
      const int nclr = handed ? ns * 2 * C : WS_CONST - WS_ACC;
      for (int t = threadIdx.x; t < nclr; t += BN_THREADS) a[t] = 0.0f;
      if (threadIdx.x == 0)
        reinterpret_cast<int*>(ws_ + (size_t)g * WS_STRIDE)[LSI_BN_WS_TAG] = 0;
    }
    if (threadIdx.x == 0)
      __hip_atomic_store(reinterpret_cast<int*>(ws_), 0, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
}

// pass 2 of the backward: dx = rstd * (dz - mean(dz) - xhat * mean(dz * xhat))
template <bool BF16>
__device__ __forceinline__ void dx_pass(const void* __restrict__ x, const void* __restrict__ dy,
                                        const float* __restrict__ mean_rstd,
                                        const float* __restrict__ beta, const float* c12,
                                        void* __restrict__ dx, long npix, int C, int relu) {
  constexpr int NV = Vec<BF16>::N;
  const int lpp = C / NV, rows = BN_THREADS / lpp;
  const int lane = threadIdx.x % lpp, row = threadIdx.x / lpp;
  float mu[NV], rs[NV], be[NV], c1[NV], c2[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane * NV + k;
    mu[k] = mean_rstd[c]; rs[k] = mean_rstd[C + c];
    be[k] = beta[c] - mu[k] * rs[k];
    c1[k] = c12[c];
    c2[k] = c12[C + c];
  }
  auto one = [&](float* v, const float* g) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float xh = (v[k] - mu[k]) * rs[k];
      const float dz = (!relu || __fmaf_rn(v[k], rs[k], be[k]) > 0.0f) ? g[k] : 0.0f;
      v[k] = rs[k] * (dz - c1[k] - xh * c2[k]);
    }
  };
  const long stride = (long)gridDim.x * rows;
  long p = (long)blockIdx.x * rows + row;
  for (; p + stride < npix; p += 2 * stride) {
    float v[2][NV], g[2][NV];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      Vec<BF16>::load(x, (p + u * stride) * lpp + lane, v[u]);
      Vec<BF16>::load(dy, (p + u * stride) * lpp + lane, g[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      one(v[u], g[u]);
      Vec<BF16>::store(dx, (p + u * stride) * lpp + lane, v[u]);
    }
  }
  for (; p < npix; p += stride) {
    float v[NV], g[NV];
    Vec<BF16>::load(x, p * lpp + lane, v);
    Vec<BF16>::load(dy, p * lpp + lane, g);
    one(v, g);
    Vec<BF16>::store(dx, p * lpp + lane, v);
  }
}

// pass 1 of the backward: sums of dz = dy * [z > 0] and of dz * xhat
template <bool BF16>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_stats_kernel(
    const void* __restrict__ x_, const void* __restrict__ dy_,
    const float* __restrict__ mean_rstd_, const float* __restrict__ beta,
    float* __restrict__ ws_, long npix, int C, int relu) {
  constexpr int NV = Vec<BF16>::N;
  const void* x = grp_in<BF16>(x_, npix, C);
  const void* dy = grp_in<BF16>(dy_, npix, C);
  const float* mean_rstd = mean_rstd_ + (size_t)blockIdx.y * 2 * C;
  float* ws = ws_ + (size_t)blockIdx.y * WS_STRIDE;
  const int lpp = C / NV, rows = BN_THREADS / lpp;
  const int lane = threadIdx.x % lpp, row = threadIdx.x / lpp;
  float mu[NV], rs[NV], be[NV], s[NV], q[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    mu[k] = mean_rstd[lane * NV + k]; rs[k] = mean_rstd[C + lane * NV + k];
    // (the forward's z = x * rstd + (beta - mean * rstd): the same mask)
    be[k] = beta[lane * NV + k] - mu[k] * rs[k]; s[k] = 0.f; q[k] = 0.f;
  }
  const long stride = (long)gridDim.x * rows;
  long p = (long)blockIdx.x * rows + row;
  auto acc = [&](const float* v, const float* g) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float xh = (v[k] - mu[k]) * rs[k];
      const float dz = (!relu || __fmaf_rn(v[k], rs[k], be[k]) > 0.0f) ? g[k] : 0.0f;
      s[k] += dz;
      q[k] = __fmaf_rn(dz, xh, q[k]);
    }
  };
  for (; p + 3 * stride < npix; p += 4 * stride) {   // eight loads in flight
    float v[4][NV], g[4][NV];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      Vec<BF16>::load(x, (p + u * stride) * lpp + lane, v[u]);
      Vec<BF16>::load(dy, (p + u * stride) * lpp + lane, g[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc(v[u], g[u]);
  }
  for (; p < npix; p += stride) {
    float v[NV], g[NV];
    Vec<BF16>::load(x, p * lpp + lane, v);
    Vec<BF16>::load(dy, p * lpp + lane, g);
    acc(v, g);
  }
  __shared__ float tot[2 * 2048];
  if (reduce_all<NV>(s, q, lpp, C, ws, tot)) {
    const float inv_m = (float)(1.0 / (double)npix);
    float* c12 = ws + WS_CONST;
    for (int c = threadIdx.x; c < C; c += BN_THREADS) {
      c12[2 * C + c] = tot[c];   // the group's share of dbeta (added up in pass 2)
      tot[c] *= inv_m;
      tot[C + c] *= inv_m;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += BN_THREADS) c12[t] = tot[t];
  }
}

template <bool BF16>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_dx_kernel(
    const void* __restrict__ x, const void* __restrict__ dy,
    const float* __restrict__ mean_rstd, const float* __restrict__ beta,
    const float* __restrict__ ws, void* __restrict__ dx, float* __restrict__ dbeta,
    long npix, int C, int relu) {
  if (blockIdx.x == 0 && blockIdx.y == 0) {  // dbeta = the groups' sums of dz
    for (int c = threadIdx.x; c < C; c += BN_THREADS) {
      float t = 0.0f;
      for (unsigned g = 0; g < gridDim.y; ++g)
        t += ws[(size_t)g * WS_STRIDE + WS_CONST + 2 * C + c];
      dbeta[c] = t;
    }
  }
  dx_pass<BF16>(grp_in<BF16>(x, npix, C), grp_in<BF16>(dy, npix, C),
                       mean_rstd + (size_t)blockIdx.y * 2 * C, beta,
                       ws + (size_t)blockIdx.y * WS_STRIDE + WS_CONST,
                       grp_out<BF16>(dx, npix, C), npix, C, relu);
}

bool bn_shape_ok(long npix, int C, int bf16) {
  const int nv = bf16 ? 8 : 4;
  if (npix <= 0 || C <= 0 || C % nv) return false;
  const int lpp = C / nv;
  return lpp <= BN_THREADS && (lpp & (lpp - 1)) == 0 && C <= 2048;
}

// Workgroups per group.  (Both passes in ONE launch for small activations --
// last workgroup publishes the constants, the others wait for a flag -- was
// built and measured: 16.4 / 21.7 us against 13.8 / 19.5 us for two launches
// at 1.6 MB, 277-281 against 285 samples/s for the bf16 training step; dropped.)
// Workgroups per group of a pass.  >= 16 steps of `rows` pixels per workgroup (the
// reduction and its atomics amortised), at most 8 workgroups per CU.  Tensors of
// at most 2 M elements (the U-Net's 2 x 6 ... 16 x 48 maps) used to get 12 - 48
// workgroups that way, each a chain of 4 dependent rounds of loads on an idle
// chip: there the steps per workgroup go down to 2 while the launch has fewer
// than 512 workgroups (tools/bn_small_time.py, pair of kernels inside a graph:
// 18.9 -> 10.8 us backward, 14.7 -> 9.8 forward at 8 x 4 x 12 x 512; the 6 - 25 MB
// tensors get slower with more workgroups -- more atomics on the same lines --
// and keep 16).  LSI_BN_MIN_STEPS overrides the 2.
int bn_grid(long npix, int C, int bf16, int groups = 1) {
  static const char* env = getenv("LSI_BN_MIN_STEPS");   // experiments
  const int min_steps = env ? atoi(env) : 2;
  const int rows = BN_THREADS / (C / (bf16 ? 8 : 4));
  const long nstep = (npix + rows - 1) / rows;        // steps of the whole group
  long steps = 16;
  if (npix * C * groups <= (2l << 20))
    while (steps > min_steps && (nstep + steps - 1) / steps * groups < 512) steps >>= 1;
  if (steps < 1) steps = 1;
  long g = (nstep + steps - 1) / steps;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;                           // 8 workgroups per CU
  return (int)g;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// workspace: groups x (counters, accumulators, constants) -- shape independent
extern "C" size_t lsi_bn_workspace_floats(int64_t npix, int32_t C, int32_t bf16,
                                          int32_t groups) {
  if (!bn_shape_ok(npix, C, bf16) || groups < 1) return 0;
  return (size_t)WS_STRIDE * groups;
}

extern "C" int lsi_bn_relu_fwd(const void* x, void* y, const float* beta,
                               float* workspace, float* mean_rstd, int64_t npix,
                               int32_t C, int32_t bf16, int32_t relu, float eps,
                               int32_t groups, lsi_stream_t stream_) {
  if (!x || !y || !beta || !workspace || !mean_rstd) return LSI_ENULL;
  if (!bn_shape_ok(npix, C, bf16) || !al16(x) || !al16(y) || groups < 1 ||
      groups > 65535)
    return LSI_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  const dim3 grid(bn_grid(npix, C, bf16, groups), groups), blk(BN_THREADS);
  if (bf16) {
    hipLaunchKernelGGL(bn_stats_kernel<true>, grid, blk, 0, st, x, beta, workspace,
                       mean_rstd, (long)npix, C, eps);
    hipLaunchKernelGGL(bn_norm_kernel<true>, grid, blk, 0, st, x, y,
                       (const float*)workspace, (long)npix, C, relu);
  } else {
    hipLaunchKernelGGL(bn_stats_kernel<false>, grid, blk, 0, st, x, beta, workspace,
                       mean_rstd, (long)npix, C, eps);
    hipLaunchKernelGGL(bn_norm_kernel<false>, grid, blk, 0, st, x, y,
                       (const float*)workspace, (long)npix, C, relu);
  }
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

extern "C" int lsi_bn_relu_norm(const void* x, void* y, const float* beta, float* workspace,
                                float* mean_rstd, int64_t npix, int32_t C, int32_t bf16,
                                int32_t relu, float eps, int32_t groups, lsi_stream_t stream_) {
  if (!x || !y || !beta || !workspace || !mean_rstd) return LSI_ENULL;
  if (!bn_shape_ok(npix, C, bf16) || !al16(x) || !al16(y) || groups < 1 || groups > 65535)
    return LSI_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  const dim3 grid(bn_grid(npix, C, bf16, groups), groups), blk(BN_THREADS);
  const int ns = lsi_bn_stat_slots(C);
  if (bf16)
    hipLaunchKernelGGL(bn_norm_sums_kernel<true>, grid, blk, 0, st, x, y, beta, workspace,
                       mean_rstd, (long)npix, C, relu, eps, ns);
  else
    hipLaunchKernelGGL(bn_norm_sums_kernel<false>, grid, blk, 0, st, x, y, beta, workspace,
                       mean_rstd, (long)npix, C, relu, eps, ns);
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}

extern "C" int lsi_bn_stats_discard(float* workspace, int32_t groups, lsi_stream_t stream_) {
  if (!workspace) return LSI_ENULL;
  if (groups < 1 || groups > 65535) return LSI_EINVAL;
  for (int g = 0; g < groups; ++g)
    if (hipMemsetAsync(workspace + (size_t)g * WS_STRIDE, 0, (size_t)WS_CONST * sizeof(float),
                       (hipStream_t)stream_) != hipSuccess)
      return LSI_ELAUNCH;
  return LSI_OK;
}

extern "C" int lsi_bn_relu_bwd(const void* x, const void* dy, const float* mean_rstd,
                               const float* beta, void* dx, float* dbeta,
                               float* workspace, int64_t npix, int32_t C,
                               int32_t bf16, int32_t relu, int32_t groups,
                               lsi_stream_t stream_) {
  if (!x || !dy || !mean_rstd || !beta || !dx || !dbeta || !workspace) return LSI_ENULL;
  if (!bn_shape_ok(npix, C, bf16) || !al16(x) || !al16(dy) || !al16(dx) ||
      groups < 1 || groups > 65535)
    return LSI_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  const dim3 grid(bn_grid(npix, C, bf16, groups), groups), blk(BN_THREADS);
  if (bf16) {
    hipLaunchKernelGGL(bn_bwd_stats_kernel<true>, grid, blk, 0, st, x, dy, mean_rstd,
                       beta, workspace, (long)npix, C, relu);
    hipLaunchKernelGGL(bn_bwd_dx_kernel<true>, grid, blk, 0, st, x, dy, mean_rstd, beta,
                       (const float*)workspace, dx, dbeta, (long)npix, C, relu);
  } else {
    hipLaunchKernelGGL(bn_bwd_stats_kernel<false>, grid, blk, 0, st, x, dy, mean_rstd,
                       beta, workspace, (long)npix, C, relu);
    hipLaunchKernelGGL(bn_bwd_dx_kernel<false>, grid, blk, 0, st, x, dy, mean_rstd, beta,
                       (const float*)workspace, dx, dbeta, (long)npix, C, relu);
  }
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
