// Read-bandwidth microbenchmark (gfx950): how many bytes must a CU keep in
// flight to stream from HBM, and does the renderer's access pattern (each
// workgroup walks the rows of its own band in L layer planes) matter?
//   mode 0: linear -- workgroup g streams its contiguous 1/G slice
//   mode 1: banded -- tensor [L][B][H][W*4 floats]; workgroup (band, b) walks
//           rows of its band, per row 3 segments x L layers of 4 KiB, like the
//           stream kernel's tasks
// DEPTH = independent 16-byte loads per lane in flight (1 KiB per wave each).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int DEPTH>
__global__ __launch_bounds__(1024) void rd(const float4* __restrict__ src, float* out,
                                           long n4_per_wg, int mode, int L, int B, int H,
                                           int rows_per_band, long row4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  if (mode == 0) {
    const float4* p = src + (long)blockIdx.x * n4_per_wg;
    // each wave owns chunks of DEPTH KiB round-robin
    for (long c = (long)wave * DEPTH * 64; c + DEPTH * 64 <= n4_per_wg; c += (long)NW * DEPTH * 64) {
      float4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = p[c + d * 64 + lane];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
    }
  } else {
    const int nbands = H / rows_per_band;
    const int band = blockIdx.x % nbands, b = blockIdx.x / nbands;
    // items: (row in band, segment of 256 float4 = 4 KiB, layer); a wave takes items round-robin
    const int nseg = (int)(row4 / 256);
    const int nitem = rows_per_band * nseg * L;
    for (int it = wave; it < nitem; it += NW) {
      const int l = it % L, sg = (it / L) % nseg, r = it / (L * nseg);
      const float4* p = src + (((long)l * B + b) * H + (long)band * rows_per_band + r) * row4 + sg * 256;
      float4 v[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) v[d] = p[d * 64 + lane];
#pragma unroll
      for (int d = 0; d < 4; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
      if (DEPTH == 1) {  // dependent: next addresses wait for this data
        asm volatile("" : "+v"(acc.x));
      }
    }
  }
  if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int DEPTH>
void run(const float4* src, float* out, size_t bytes, int mode, int threads, int grid, int rows_per_band) {
  const int L = 4, B = 32, H = 256; const long row4 = 768;  // 768 px x 16 B
  const long n4 = bytes / 16, per = n4 / grid;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    rd<DEPTH><<<grid, threads>>>(src, out, per, mode, L, B, H, rows_per_band, row4);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d depth %d threads %4d grid %4d rows/band %2d: %7.1f us  %6.2f TB/s\n", mode, DEPTH, threads,
         grid, rows_per_band, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)4 * 32 * 256 * 768 * 16;  // 403 MB, cfg3's input size
  float4* src; float* out;
  (void)hipMalloc(&src, bytes + (1 << 20)); (void)hipMalloc(&out, 4096);
  (void)hipMemset(src, 0, bytes);
  for (int threads : {1024, 512}) {
    for (int grid : {256, 512, 1024}) {
      run<1>(src, out, bytes, 0, threads, grid, 0);
      run<2>(src, out, bytes, 0, threads, grid, 0);
      run<4>(src, out, bytes, 0, threads, grid, 0);
      run<8>(src, out, bytes, 0, threads, grid, 0);
    }
  }
  for (int rpb : {32, 16, 8}) {
    run<4>(src, out, bytes, 1, 1024, 32 * (256 / rpb), rpb);
    run<1>(src, out, bytes, 1, 1024, 32 * (256 / rpb), rpb);
  }
  return 0;
}
