// LDS read-modify-write + VALU overlap microbenchmark (gfx950), 12 waves/CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP6(x) x x x x x x
#define VALU3 "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n"

// MODE bit0: VALU burst of 18; bit1: LDS RMW pair; STRIDE in bytes between lanes
template <int MODE, int STRIDE>
__global__ __launch_bounds__(768) void k(float* out, int iters) {
  extern __shared__ float4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-9f, b = a + 1e-9f, c = b + 1e-9f;
  float4* base = lds + wave * 512;
  for (int i = lane; i < 512; i += 64) base[i] = make_float4(0, 0, 0, 0);
  float4* cell = (float4*)((char*)base + lane * STRIDE);
  float4 V = make_float4(a, b, c, 1.f);
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    if (MODE & 1) asm volatile(REP6(VALU3) : "+v"(a), "+v"(b), "+v"(c));
    if (MODE & 2) {
      float4 t = cell[0];
      t.x += V.x * a; t.y += V.y * a; t.z += V.z * a; t.w += V.w * a;
      cell[0] = t;
      asm volatile("" ::: "memory");
      float4 u = cell[1];
      u.x += V.x * b; u.y += V.y * b; u.z += V.z * b; u.w += V.w * b;
      cell[1] = u;
      asm volatile("" ::: "memory");
    }
    if (MODE & 4) {  // both reads first, then both writes (needs disjoint cells)
      float4 t = cell[0], u = cell[1];
      t.x += V.x * a; t.y += V.y * a; t.z += V.z * a; t.w += V.w * a;
      u.x += V.x * b; u.y += V.y * b; u.z += V.z * b; u.w += V.w * b;
      cell[0] = t; cell[1] = u;
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  float4 r = base[lane];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + r.x + r.w;
}

template <int MODE, int STRIDE>
void run(const char* name, int threads) {
  const int blocks = 256, iters = 4000;
  float* out;
  (void)hipMalloc(&out, sizeof(float) * blocks * 1024);
  const size_t lds = 12 * 512 * 16 + 4096;
  (void)hipFuncSetAttribute((const void*)k<MODE, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<MODE, STRIDE><<<blocks, threads, lds>>>(out, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<MODE, STRIDE><<<blocks, threads, lds>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s stride %3d threads %4d: %.2f ns/iter\n", name, STRIDE, threads, ms * 1e6 / iters);
  (void)hipFree(out);
}

int main() {
  for (int threads : {256, 768}) {
    run<1, 16>("18 VALU", threads);
    run<2, 16>("RMW pair", threads);
    run<2, 32>("RMW pair", threads);
    run<2, 64>("RMW pair", threads);
    run<3, 16>("18 VALU + RMW pair", threads);
    run<3, 32>("18 VALU + RMW pair", threads);
    run<4, 32>("RMW pair, reads first", threads);
    run<5, 32>("18 VALU + RMW pair reads first", threads);
  }
  return 0;
}
