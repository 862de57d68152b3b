// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths and patterns the renderer's kernels use (MI355X_MICROARCH.md: FETCH_SIZE
// counts wide coalesced reads at half their size; other widths and WRITE_SIZE
// must be calibrated on a known byte count).  Every kernel moves exactly
// `bytes` (printed) through one pattern; tools/pmc_calib.sh runs the program
// under the two --pmc passes and prints counter * 1024 / bytes per pattern.
//   hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib.bin tools/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { if ((x) != hipSuccess) { printf("hip error %s:%d\n", __FILE__, __LINE__); exit(1); } } while (0)

// 16-byte coalesced loads (the forward's inputs); the sum keeps the loads alive
extern "C" __global__ void calib_read16(const v4* __restrict__ p, float* out, size_t n) {
  v4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += p[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}
// three 16-byte loads per lane, lanes 48 bytes apart (channels-last RGB)
extern "C" __global__ void calib_read16_s48(const v4* __restrict__ p, float* out, size_t n) {
  v4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 3; i += (size_t)gridDim.x * blockDim.x) {
    acc += p[3 * i]; acc += p[3 * i + 1]; acc += p[3 * i + 2];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}
extern "C" __global__ void calib_write16(v4* __restrict__ p, size_t n) {
  const v4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}
extern "C" __global__ void calib_write16_nt(v4* __restrict__ p, size_t n) {
  const v4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, p + i);
}
// three 16-byte stores per lane, lanes 48 bytes apart (gradients of the colours)
extern "C" __global__ void calib_write16_s48(v4* __restrict__ p, size_t n) {
  const v4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 3; i += (size_t)gridDim.x * blockDim.x) {
    p[3 * i] = v; p[3 * i + 1] = v; p[3 * i + 2] = v;
  }
}
extern "C" __global__ void calib_write16_s48_nt(v4* __restrict__ p, size_t n) {
  const v4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 3; i += (size_t)gridDim.x * blockDim.x) {
    __builtin_nontemporal_store(v, p + 3 * i); __builtin_nontemporal_store(v, p + 3 * i + 1);
    __builtin_nontemporal_store(v, p + 3 * i + 2);
  }
}
// scalar stores out[3 i + k] and w[i] (the 4-byte epilogues)
extern "C" __global__ void calib_write4_s12(float* __restrict__ p, size_t nfl) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nfl / 3; i += (size_t)gridDim.x * blockDim.x) {
    p[3 * i] = 1.f; p[3 * i + 1] = 2.f; p[3 * i + 2] = 3.f;
  }
}
extern "C" __global__ void calib_copy16(const v4* __restrict__ s, v4* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = s[i];
}

int main() {
  const size_t bytes = (size_t)768 << 20;  // 768 MiB: three times the Infinity Cache
  const size_t n = bytes / 16;
  v4 *a, *b;
  float* out;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
  const dim3 g(256 * 8), t(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_read16, g, t, 0, 0, a, out, n);
    hipLaunchKernelGGL(calib_read16_s48, g, t, 0, 0, b, out, n / 3 * 3);
    hipLaunchKernelGGL(calib_write16, g, t, 0, 0, a, n);
    hipLaunchKernelGGL(calib_write16_nt, g, t, 0, 0, b, n);
    hipLaunchKernelGGL(calib_write16_s48, g, t, 0, 0, a, n / 3 * 3);
    hipLaunchKernelGGL(calib_write16_s48_nt, g, t, 0, 0, b, n / 3 * 3);
    hipLaunchKernelGGL(calib_write4_s12, g, t, 0, 0, (float*)a, n * 4 / 3 * 3);
    hipLaunchKernelGGL(calib_copy16, g, t, 0, 0, a, b, n);
  }
  CHECK(hipDeviceSynchronize());
  printf("bytes_per_pattern %zu\n", bytes);
  return 0;
}
