// Micro-benchmarks that size the splat kernel design on MI355X:
//   (1) LDS fp32 atomic (ds_add_f32) throughput for the access patterns of a
//       4-corner x 4-channel splat,
//   (2) global fp32 atomic throughput (canvas flush / atomic path),
//   (3) streaming read bandwidth at the BASELINE config sizes.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: lane-unique address, stride 1 dword     (conflict free)
// mode 1: lanes 2k,2k+1 share an address           (2 source px per target cell)
// mode 2: address = (lane/2)*4 + ch, 4 channels    (interleaved RGBW canvas)
// mode 3: address = (lane/2) + ch*PLANE            (planar canvas)
// mode 4: lane-unique, stride 4 dwords             (interleaved, 1 px per cell)
template <int MODE>
__global__ void lds_atomic_kernel(float* out, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const float v = 1.0f + tid * 1e-6f;
  for (int it = 0; it < iters; ++it) {
    const int base = ((wave * 97 + it * 13) & 15) * 256;  // move around the tile
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // 4 corners
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) { // 4 channels
        int a;
        if (MODE == 0) a = base + lane + 64 * ((k * 4 + ch) & 3);
        else if (MODE == 1) a = base + (lane >> 1) + 32 * ((k * 4 + ch) & 7);
        else if (MODE == 2) a = base + ((lane >> 1) + (k & 1) + 40 * (k >> 1)) * 4 + ch;
        else if (MODE == 3) a = base + (lane >> 1) + (k & 1) + 40 * (k >> 1) + ch * 2048;
        else a = (base + (lane + (k & 1) + 72 * (k >> 1)) * 4 + ch) & 8191;
        atomicAdd(&lds[a & 8191], v);
      }
    }
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = lds[5];
}

__global__ void global_atomic_kernel(float* buf, size_t n, int iters, int stride_mode) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (int it = 0; it < iters; ++it) {
    size_t i = (gid + (size_t)it * nthreads);
    if (stride_mode == 1) i = i * 4;            // one channel of an interleaved canvas
    atomicAdd(&buf[i % n], 1.0f);
  }
}

__global__ void read_kernel(const float4* __restrict__ in, size_t n4, float* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = in[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <typename F>
float time_ms(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 20));
  const int iters = 200, blocks = 256 * 4, threads = 256;
  const char* names[5] = {"unique stride1", "pairs share addr", "interleaved RGBW pairs",
                          "planar pairs", "interleaved unique"};
  for (int mode = 0; mode < 5; ++mode) {
    auto launch = [&]() {
      switch (mode) {
        case 0: hipLaunchKernelGGL(lds_atomic_kernel<0>, dim3(blocks), dim3(threads), 32768, 0, out, iters); break;
        case 1: hipLaunchKernelGGL(lds_atomic_kernel<1>, dim3(blocks), dim3(threads), 32768, 0, out, iters); break;
        case 2: hipLaunchKernelGGL(lds_atomic_kernel<2>, dim3(blocks), dim3(threads), 32768, 0, out, iters); break;
        case 3: hipLaunchKernelGGL(lds_atomic_kernel<3>, dim3(blocks), dim3(threads), 32768, 0, out, iters); break;
        default: hipLaunchKernelGGL(lds_atomic_kernel<4>, dim3(blocks), dim3(threads), 32768, 0, out, iters); break;
      }
    };
    const float ms = time_ms(launch, 5);
    const double atomics = (double)blocks * threads * iters * 16;
    printf("LDS ds_add_f32 [%-24s] %.3f ms  %.1f G atomics/s  (%.2f lane-atomics/clk/CU @2.4GHz)\n",
           names[mode], ms, atomics / ms / 1e6, atomics / ms / 1e6 / 256 / 2.4);
  }
  for (int sm = 0; sm < 2; ++sm) {
    for (size_t mb : {4, 64, 512}) {
      const size_t n = mb * 1024 * 1024 / 4;
      float* buf; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
      const int gi = 16;
      auto launch = [&]() { hipLaunchKernelGGL(global_atomic_kernel, dim3(256 * 8), dim3(256), 0, 0, buf, n, gi, sm); };
      const float ms = time_ms(launch, 5);
      const double atomics = 256.0 * 8 * 256 * gi;
      printf("global atomic add f32 [%s, %zu MB buffer] %.3f ms  %.1f G atomics/s\n",
             sm ? "stride 4 dwords" : "coalesced      ", mb, ms, atomics / ms / 1e6);
      CK(hipFree(buf));
    }
  }
  for (size_t mb : {25, 54, 100, 403, 1024}) {
    const size_t n4 = mb * 1000 * 1000 / 16;
    float4* buf; CK(hipMalloc(&buf, n4 * 16)); CK(hipMemset(buf, 1, n4 * 16));
    for (int bpc : {4, 8}) {
      auto launch = [&]() { hipLaunchKernelGGL(read_kernel, dim3(256 * bpc), dim3(256), 0, 0, buf, n4, out); };
      const float ms = time_ms(launch, 10);
      printf("stream read %4zu MB, %d blocks/CU: %.2f us  %.0f GB/s\n", mb, bpc, ms * 1e3, n4 * 16 / ms / 1e6);
    }
    CK(hipFree(buf));
  }
  return 0;
}
