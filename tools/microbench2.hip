// LDS primitive throughput on MI355X (per CU, lanes/clk) to choose the scatter
// accumulation scheme: fp32/int atomics (with/without return), plain RMW at
// b32/b64/b128, bpermute.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_LDS 8192
template <int OP>
__global__ void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* ldu = (unsigned*)lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < N_LDS; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  float acc = 0.f; unsigned uacc = 0;
  float4 v4 = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int it = 0; it < iters; ++it) {
    // address pattern: lane -> its own 16B slot, waves on disjoint 1KB rows, moving
    const int slot = ((wave * 64 + lane) + it * 64 * 5) & (N_LDS / 4 - 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int s = (slot + r * 256) & (N_LDS / 4 - 1);
      if (OP == 0) atomicAdd(&lds[s * 4], 1.0f);                       // ds_add_f32
      else if (OP == 1) acc += atomicAdd(&lds[s * 4], 1.0f);           // ds_add_rtn_f32
      else if (OP == 2) atomicAdd(&ldu[s * 4], 1u);                    // ds_add_u32
      else if (OP == 3) uacc += atomicAdd(&ldu[s * 4], 1u);            // ds_add_rtn_u32
      else if (OP == 4) uacc += atomicExch(&ldu[s * 4], (unsigned)lane); // ds_wrxchg_rtn
      else if (OP == 5) { float4 t = *(float4*)&lds[s * 4]; t.x += v4.x; t.y += v4.y; t.z += v4.z; t.w += v4.w; *(float4*)&lds[s * 4] = t; } // RMW b128
      else if (OP == 6) { float t = lds[s * 4]; lds[s * 4] = t + 1.0f; }  // RMW b32
      else if (OP == 7) { float2 t = *(float2*)&lds[s * 4]; t.x += 1.f; t.y += 2.f; *(float2*)&lds[s * 4] = t; } // RMW b64
      else if (OP == 8) { ldu[s * 4] = lane; uacc += ldu[s * 4]; }       // claim write+read b32
      else if (OP == 9) { uacc += __builtin_amdgcn_ds_bpermute(((lane * 7 + r) & 63) * 4, (int)uacc + lane); } // bpermute
      else if (OP == 10) atomicMax(&ldu[s * 4], (unsigned)(lane + it));  // ds_max_u32
      else if (OP == 11) { *(float4*)&lds[s * 4] = v4; }                 // write b128 only
      else if (OP == 12) { float4 t = *(float4*)&lds[s * 4]; acc += t.x + t.w; } // read b128 only
    }
  }
  __syncthreads();
  if (acc + (float)uacc == 12345.678f) out[0] = acc;
  if (tid == 0) out[blockIdx.x + 1] = lds[4];
}
template <int OP>
void run(const char* name, float* out) {
  const int iters = 100, blocks = 256 * 2, threads = 512;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), N_LDS * 4, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), N_LDS * 4, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double ops = (double)blocks * threads * iters * 8;
  printf("%-28s %.3f ms  %7.1f G lane-ops/s  %.2f lanes/clk/CU  (%.1f clk per wave-instr)\n", name, ms,
         ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4, 64.0 / (ops / ms / 1e6 / 256 / 2.4));
}
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  run<0>("ds_add_f32", out); run<1>("ds_add_rtn_f32", out); run<2>("ds_add_u32", out);
  run<3>("ds_add_rtn_u32", out); run<4>("ds_wrxchg_rtn_b32", out); run<10>("ds_max_u32", out);
  run<5>("RMW b128 (read+add+write)", out); run<7>("RMW b64", out); run<6>("RMW b32", out);
  run<8>("claim write+read b32", out); run<9>("ds_bpermute_b32", out);
  run<11>("write b128", out); run<12>("read b128", out);
  return 0;
}
