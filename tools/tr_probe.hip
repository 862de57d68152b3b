// Semantics of ds_read_b64_tr_b16 (gfx950): LDS image [rows][64] of shorts with
// value = row * 64 + col; lane t of a 16-lane group points at row (t >> 2),
// columns 4 (t & 3) .. + 3 of its group's rows; prints what every lane gets.
//   hipcc --offload-arch=gfx950 -O2 -o tr_probe tools/tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, t = lane & 15, grp = lane >> 4;
  const int row = 8 * grp + (t >> 2), col = 4 * (t & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(lds + row * 64 + col));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
    printf("\n");
  }
  return 0;
}
