// Read-bandwidth microbenchmark (gfx950) with the stream kernel's REAL access
// pattern: workgroup = (band of source rows, batch element); task = (source
// row, 256-pixel segment); item = one layer of a task = 1 KiB of disparities +
// 3 KiB of channels-last texture per wave (4 x dwordx4 per lane).  DEPTH items
// of loads are kept in flight per wave (register sets), `alu` dependent VALU
// rounds per item stand in for the projection (4 VALU each).
//   hipcc --offload-arch=gfx950 -O3 -o microbench6 microbench6.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct Set { float4 d, t0, t1, t2; };

template <int DEPTH>
__global__ __launch_bounds__(1024) void rd(const float* __restrict__ tex,
                                           const float* __restrict__ disp, float* out,
                                           int L, int B, int H, int W, int rows_per_band,
                                           int halo, int alu, int order, int b0, int touch,
                                           int stagger, int padpx) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int nbands = H / rows_per_band;
  // XCD-aware: contiguous runs of bands per XCD
  const unsigned nwg = gridDim.x, lin = blockIdx.x;
  const unsigned xcd = lin & 7u, q = nwg >> 3;
  const unsigned id = (nwg & 7u) ? lin : xcd * q + (lin >> 3);
  const int band = id % nbands, b = b0 + id / nbands;
  const int ya = max(0, band * rows_per_band - halo);
  const int yb = min(H, (band + 1) * rows_per_band + halo);
  const int nseg = W / 256, nrow = yb - ya;
  const int ntask = nrow * nseg;
  const int nitem_w = ((ntask - wave + NW - 1) / NW) * L;  // this wave's items
  auto issue = [&](Set& s, int j) {
    if (j >= nitem_w) j = 0;  // harmless re-read
    int t = wave + NW * (j / L);
    const int l = j % L;
    // stagger: every workgroup walks its tasks from another starting point, so
    // that the 256 workgroups do not read the same offset of their images at
    // the same time (addresses 2^k * 9 bytes apart: the same HBM channels?)
    if (stagger) t = (t + (b * 29 + band * 13) * stagger) % ntask;
    int r = t / nseg, sg = t % nseg;
    if (order) {  // rows a fifth of the band apart
      const int q5 = (nrow + 4) / 5;
      r = min((r % 5) * q5 + r / 5, nrow - 1);
    }
    const long px = ((long)l * 32 + b) * ((long)H * W + padpx) + (long)(ya + r) * W + sg * 256 + 4 * lane;
    const float4* pd = reinterpret_cast<const float4*>(disp + px);
    const float4* pt = reinterpret_cast<const float4*>(tex + 3 * px);
    if (touch < 0) {
      // contiguous variant: every texture load instruction covers 1 KiB (lane
      // k reads bytes 16 k .. of the segment's 3 KiB; the lane's own pixels
      // would then need a cross-lane exchange)
      const float4* ps = reinterpret_cast<const float4*>(tex + 3 * (px - 4 * lane));
      s.d = pd[0]; s.t0 = ps[lane]; s.t1 = ps[64 + lane]; s.t2 = ps[128 + lane];
    } else {
      s.d = pd[0]; s.t0 = pt[0]; s.t1 = pt[1]; s.t2 = pt[2];
    }
  };
  float4 acc = make_float4(0, 0, 0, 0);
  auto consume = [&](Set& s) {
    float4 v;
    v.x = s.d.x + s.t0.x + s.t1.x + s.t2.x;
    v.y = s.d.y + s.t0.y + s.t1.y + s.t2.y;
    v.z = s.d.z + s.t0.z + s.t1.z + s.t2.z;
    v.w = s.d.w + s.t0.w + s.t1.w + s.t2.w;
    for (int i = 0; i < alu; ++i) {
      acc.x = __fmaf_rn(acc.x, 1.0001f, v.x); acc.y = __fmaf_rn(acc.y, 1.0001f, v.y);
      acc.z = __fmaf_rn(acc.z, 1.0001f, v.z); acc.w = __fmaf_rn(acc.w, 1.0001f, v.w);
    }
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  };
  Set s[DEPTH];
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) issue(s[k], k);
  // touch: bring the lines of up to `touch` further items into L2 now (one
  // dword per lane straight into a junk LDS area: no registers)
  {
    float* junk = reinterpret_cast<float*>(smem) + threadIdx.x;
    for (int j = DEPTH; j < min(nitem_w, DEPTH + touch); ++j) {
      const int t = wave + NW * (j / L), l = j % L;
      int r = t / nseg, sg = t % nseg;
      if (order) {
        const int q5 = (nrow + 4) / 5;
        r = min((r % 5) * q5 + r / 5, nrow - 1);
      }
      const long px = (((long)l * 32 + b) * H + (ya + r)) * W + sg * 256 + 4 * lane;
      __builtin_amdgcn_global_load_lds(disp + px, junk - lane, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(tex + 3 * px, junk - lane, 4, 0, 0);
    }
  }
  for (int j0 = 0; j0 < nitem_w; j0 += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      consume(s[k]);
      issue(s[k], j0 + k + DEPTH);
    }
  }
  if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w + smem[0];
}

static const size_t npx_set = (size_t)4 * 32 * 256 * 768;
template <int DEPTH>
void run(const float* tex, const float* disp, float* out, int B, int threads, int rpb, int halo,
         int alu, int order, int lds, int touch = 0, int stagger = 0, int padpx = 0) {
  const int L = 4, H = 256, W = 768;
  const int grid = B * (H / rpb);
  const double bytes = (double)L * B * H * W * 16;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)rd<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float best = 1e9f;
  // rotate over the 32 batch elements of two input sets: no launch re-reads
  // what the 256 MiB Infinity Cache still holds
  const int nrot = 2 * 32 / B;
  for (int rep = 0; rep < (B == 32 ? 5 : 24); ++rep) {
    const int r = rep % nrot;
    const size_t set_px = npx_set + (size_t)4 * 32 * padpx;
    const float* tx = tex + (size_t)(r * B / 32) * set_px * 3;
    const float* dp = disp + (size_t)(r * B / 32) * set_px;
    (void)hipEventRecord(e0);
    rd<DEPTH><<<grid, threads, lds>>>(tx, dp, out, L, B, H, W, rpb, halo, alu, order, (r * B) % 32, touch, stagger, padpx);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 1 && ms < best) best = ms;
  }
  printf("B %2d depth %d touch %2d threads %4d rows/band %2d halo %d alu %3d order %d lds %3dK stagger %d pad %5d: %7.1f us  %5.2f TB/s (useful)\n",
         B, DEPTH, touch, threads, rpb, halo, alu, order, lds / 1024, stagger, padpx, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main() {
  const int L = 4, Bmax = 32, H = 256, W = 768;
  const size_t npx = (size_t)L * Bmax * H * W;
  // two input sets so that consecutive launches do not hit the Infinity Cache
  float *tex, *disp, *out;
  const size_t padmax = (size_t)L * Bmax * 8192;  // room for padded batch strides
  (void)hipMalloc(&tex, (npx + padmax) * 12 * 2 + (1 << 20)); (void)hipMalloc(&disp, (npx + padmax) * 4 * 2 + (1 << 20));
  (void)hipMalloc(&out, 4096);
  (void)hipMemset(tex, 0, npx * 12 * 2); (void)hipMemset(disp, 0, npx * 4 * 2);
  const int K = 1024;
  if (getenv("MB6_STAGGER")) {
    printf("--- staggered task order / padded batch stride, B=32\n");
    for (int rep = 0; rep < 2; ++rep)
      for (int alu : {0, 40, 75}) {
        run<2>(tex, disp, out, 32, 768, 32, 1, alu, 1, 100 * K, 0, 0, 0);
        run<2>(tex, disp, out, 32, 768, 32, 1, alu, 1, 100 * K, 0, 1, 0);
        run<2>(tex, disp, out, 32, 768, 32, 1, alu, 1, 100 * K, 0, 3, 0);
        run<2>(tex, disp, out, 32, 768, 32, 1, alu, 1, 100 * K, 0, 0, 1088);
        run<2>(tex, disp, out, 32, 768, 32, 1, alu, 1, 100 * K, 0, 0, 4160);
        run<2>(tex, disp, out, 32, 768, 32, 1, alu, 1, 100 * K, 0, 1, 4160);
      }
    printf("--- B=4\n");
    for (int alu : {0, 40}) {
      run<2>(tex, disp, out, 4, 768, 4, 1, alu, 0, 100 * K, 0, 0, 0);
      run<2>(tex, disp, out, 4, 768, 4, 1, alu, 0, 100 * K, 0, 1, 0);
      run<2>(tex, disp, out, 4, 768, 4, 1, alu, 0, 100 * K, 0, 0, 4160);
    }
    return 0;
  }
  if (getenv("MB6_CONTIG")) {
    printf("--- texture loads: lane-strided 48 B (touch 0, the kernel's) vs 1 KiB per instruction (touch -1)\n");
    for (int rep = 0; rep < 2; ++rep)
      for (int alu : {0, 40, 75})
        for (int touch : {0, -1}) {
          run<2>(tex, disp, out, 32, 768, 32, 1, alu, 1, 100 * K, touch);
          run<1>(tex, disp, out, 32, 1024, 32, 1, alu, 1, 100 * K, touch);
        }
    for (int touch : {0, -1}) run<2>(tex, disp, out, 4, 768, 4, 1, 40, 0, 100 * K, touch);
    return 0;
  }
  if (getenv("MB6_TOUCH")) {
    printf("--- touch prefetch, B=4, 4 rows per band + halo (256 WGs)\n");
    for (int alu : {0, 40, 75}) {
      for (int touch : {0, 2, 4, 8, 32}) {
        run<2>(tex, disp, out, 4, 768, 4, 1, alu, 0, 100 * K, touch);
      }
      run<1>(tex, disp, out, 4, 768, 4, 1, alu, 0, 100 * K, 32);
      run<2>(tex, disp, out, 4, 1024, 4, 1, alu, 0, 100 * K, 32);
    }
    printf("--- touch prefetch, B=32\n");
    for (int touch : {0, 2, 4, 8}) run<2>(tex, disp, out, 32, 768, 32, 1, 75, 1, 100 * K, touch);
    return 0;
  }

  printf("--- B=32, 32 source rows per band (+2 halo), 1 WG per CU (lds 100K)\n");
  for (int threads : {768, 1024}) {
    for (int alu : {0, 40, 75}) {
      run<1>(tex, disp, out, 32, threads, 32, 1, alu, 1, 100 * K);
      run<2>(tex, disp, out, 32, threads, 32, 1, alu, 1, 100 * K);
      run<3>(tex, disp, out, 32, threads, 32, 1, alu, 1, 100 * K);
      run<4>(tex, disp, out, 32, threads, 32, 1, alu, 1, 100 * K);
    }
  }
  printf("--- row order (0: consecutive rows)\n");
  run<2>(tex, disp, out, 32, 768, 32, 1, 40, 0, 100 * K);
  run<3>(tex, disp, out, 32, 768, 32, 1, 40, 0, 100 * K);
  printf("--- 2 WGs per CU (lds 70K), 16 source rows per band, 512 threads\n");
  for (int alu : {0, 40, 75}) {
    run<2>(tex, disp, out, 32, 512, 16, 1, alu, 1, 70 * K);
    run<3>(tex, disp, out, 32, 512, 16, 1, alu, 1, 70 * K);
    run<4>(tex, disp, out, 32, 512, 16, 1, alu, 1, 70 * K);
  }
  printf("--- 1280 threads worth: 2 WGs x 640\n");
  run<2>(tex, disp, out, 32, 640, 16, 1, 40, 1, 70 * K);
  run<3>(tex, disp, out, 32, 640, 16, 1, 40, 1, 70 * K);
  printf("--- B=4 (shard of 8): 4 source rows per band + halo 1 (256 WGs)\n");
  for (int threads : {768, 1024}) {
    for (int alu : {0, 40}) {
      run<2>(tex, disp, out, 4, threads, 4, 1, alu, 0, 100 * K);
      run<3>(tex, disp, out, 4, threads, 4, 1, alu, 0, 100 * K);
      run<4>(tex, disp, out, 4, threads, 4, 1, alu, 0, 100 * K);
    }
  }
  printf("--- B=4: 2 source rows per band + halo 1 (512 WGs, 2 per CU)\n");
  run<2>(tex, disp, out, 4, 512, 2, 1, 0, 0, 70 * K);
  run<4>(tex, disp, out, 4, 512, 2, 1, 0, 0, 70 * K);
  run<2>(tex, disp, out, 4, 512, 2, 1, 40, 0, 70 * K);
  run<4>(tex, disp, out, 4, 512, 2, 1, 40, 0, 70 * K);
  printf("--- B=4 no halo, 8 rows per band (128 WGs)\n");
  run<4>(tex, disp, out, 4, 1024, 8, 0, 0, 0, 100 * K);
  return 0;
}
