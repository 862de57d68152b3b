// Issue-cost microbenchmarks for gfx950: VALU / SALU / branch mixes at 3 waves/SIMD.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mb3 tools/microbench3.hip && /tmp/mb3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP10(x) x x x x x x x x x x
#define VALU4 "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
#define SALU3 "s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20\n s_or_b32 s22, s22, s21\n"
#define BR_NT "s_cmp_eq_u32 s23, 12345\n s_cbranch_scc1 1f\n"   // never taken (s23 = 0)
#define BR_T "s_cmp_eq_u32 s23, 0\n s_cbranch_scc1 2f\n s_nop 0\n 2:\n"  // always taken, skips 1 instr

template <int MODE>
__global__ __launch_bounds__(768) void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1e-9f, b = a + 1e-9f, c = b + 1e-9f, d = c + 1e-9f;
  asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 40 VALU
      asm volatile(REP10(VALU4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    } else if (MODE == 1) {  // 40 VALU + 30 SALU interleaved
      asm volatile(REP10(VALU4 SALU3) : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"s20", "s21", "s22", "scc");
    } else if (MODE == 2) {  // 40 VALU + 10 not-taken branches
      asm volatile(REP10(VALU4 BR_NT) "1:\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"scc");
    } else if (MODE == 3) {  // 40 VALU + 10 taken branches
      asm volatile(REP10(VALU4 BR_T) : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"scc");
    } else if (MODE == 4) {  // 40 VALU + 30 SALU + 10 NT branches
      asm volatile(REP10(VALU4 SALU3 BR_NT) "1:\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"s20", "s21", "s22", "scc");
    } else if (MODE == 5) {  // 20 v_pk_fma (same flops as 40 fma)
      asm volatile(REP10("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n")
                   : "+v"(*(double*)&a), "+v"(*(double*)&c));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int MODE>
void run(const char* name, int threads) {
  const int blocks = 256, iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * 1024);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  printf("%-40s threads %4d: %.1f us, %.1f counter-ticks/iter, %.2f ns/iter\n", name, threads, ms * 1e3,
         avg / iters, ms * 1e6 / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 768}) {
    run<0>("40 VALU", threads);
    run<5>("20 v_pk_fma", threads);
    run<1>("40 VALU + 30 SALU", threads);
    run<2>("40 VALU + 10 branch not taken", threads);
    run<3>("40 VALU + 10 branch taken", threads);
    run<4>("40 VALU + 30 SALU + 10 branch NT", threads);
  }
  return 0;
}
