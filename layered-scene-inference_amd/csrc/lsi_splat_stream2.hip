// LSI_PATH_STREAM, compact instance: the configuration every documented
// command of the reference runs (ldi.py:71-182 with compose_layers=True, no
// mask input, rectified stereo with unit normaliser, channels-last textures
// or RGBD pixels, W % 4 == 0: rows of 256-pixel segments, the last one
// possibly partial).  Same algorithm and the same exact
// arithmetic as splat_stream_kernel (lsi_splat_stream.hip, which keeps every
// other case); what differs is everything around the pixel arithmetic:
//   * no prologue before the first loads: every wave finds the band's source
//     rows itself, takes its first task by wave index and has NSETS items of
//     loads in flight BEFORE the tile is cleared and the task table is filled
//     (small launches are bound by this start-up, not by bandwidth);
//   * the zbuffer weight is one fused multiply-add + exp2 (pre-scaled
//     constants), side weights come from one subtraction, the epilogue uses a
//     reciprocal: none of them feeds an index or a threshold decision;
//   * queue overflow (rare) is one out-of-line function instead of code
//     inlined at every push site: the kernel is ~10 KB of instructions instead
//     of ~55 KB (first execution of a code region is an instruction-cache miss
//     per workgroup);
//   * window slots and tile addresses of the merge are per-lane constants.
// Exactness contract: identical to lsi_splat_stream.hip (see its header and
// lsi_common.h): projected cells and every clamp decision bit-exact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/lsi_hip.h"
#include "lsi_common.h"
#include "lsi_splat_internal.h"

#pragma clang fp contract(off)

using namespace lsi;

namespace {

constexpr int SEG = 256;  // source pixels per task (64 lanes x 4)

struct __attribute__((aligned(16))) Task {
  int row0;        // target row of the task's top contribution, band-relative
  float wy0, wy1;  // row weights incl. border masks; both 0: nothing to do
  int win;         // window: (first cell + 32768) | number of cells << 16
};
struct __attribute__((aligned(8))) TaskX {
  int yx;      // source row | segment << 16
  float tmin;  // clamp threshold of the smaller row weight
};
// Ticket -> (row-segment unit, layers).  The first `nfull` tickets are whole
// units (all L layers: one merge per unit); the remaining `nsplit` units are
// handed out `lsub` layers per ticket (unit-minor order: concurrent tickets
// belong to different units), so that the waves of a short band all have work
// (a unit is L items on one wave, and a short band has fewer than two units
// per wave).
struct Unit { int u, l0, nl; };

struct S2Args {
  const float* tex;
  const float* disp;
  const float* M;
  float* out_img;
  float* out_wts;
  float* out_img_c;  // BOTH: the composed view next to the L per-layer views
  float* out_wts_c;
  // BOTH instance as the disparity pass (compute_trg_disp, ldi.py:147-180): the
  // per-layer tiles sum (d * w, -, -, w); the only output is out_disp
  float* out_disp;
  int B, H, Ht, Wt, L, nseg;
  int W;  // source width (W % 4 == 0; the last segment of a row may be partial)
  int tex_sl, tex_sb, tex_sy, disp_sl, disp_sb, disp_sy;  // element strides
  float s, max_disp, zA, zB, lbg;  // exp2(fma(clip(d), zA, zB)); L * bg weight
  float bg;                        // bg weight of one layer's canvas
  int R, wmax, qcap, cap, ilv;     // band rows, window cells, queue, table, row interleave
  int hf;                          // halo rows first (see BandOrder)
  int ep;                          // epilogue stores: 0 scalar, 1 16-byte, 2 16-byte write-through, 3 16-byte nt
  int nsplit, lsub;                // units handed out lsub layers per ticket
  float inv_gx, inv_nseg;
  long long* stamps;  // S2X_STAMPS build: [workgroup][wave][8] wall-clock ticks
  // probe launches (s2_adapt): {items on the folded routes B' / C, all items} of
  // the launch are added here, one atomic pair per workgroup; NULL otherwise
  unsigned* route_ctr;
};

#define S2_FENCE() asm volatile("" ::: "memory")
#define S2_RFL(x) __builtin_amdgcn_readfirstlane(x)

__device__ __forceinline__ int s2_div_small(int n, int d, float rcp) {
  int q = (int)((float)n * rcp);
  const int r = n - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// Ticket order of a band with `nsrc` source rows (see Unit above, and the row
// interleave: consecutive tickets go to source rows a fifth of the band apart,
// so that the units in flight merge into different tile rows; short bands keep
// the natural order).
// Halo rows first (hf): the band's first two and last two source rows are the
// ones its neighbours read as well (a band of R target rows reads 2R + 2 source
// rows at s = 0.5).  They are the first units of every band -- taken by wave
// index at launch, the one moment all workgroups are in step -- so that the two
// bands that share a row ask the XCD's L2 for it within the same microsecond
// and one of them is served from the other's fill instead of from HBM.
struct BandOrder {
  int nsrc, nrow_pad, q5, nunit, nsplit, nfull, lsub, ntask, rot, hf, nin;
  float inv_nsplit;
};
__host__ __device__ __forceinline__ int s2_rows_padded(int nsrc, int ilv, int hf) {
  const int edge = (hf && nsrc >= 6) ? 4 : 0;
  const int nin = nsrc - edge;
  return edge + (ilv ? (nin + 4) / 5 * 5 : nin);
}
// `wg`: a number that differs between workgroups (S2X_STAGGER builds: every
// workgroup walks its units from another starting point)
__device__ __forceinline__ BandOrder s2_band_order(const S2Args& a, int nsrc, int wg = 0) {
  BandOrder o;
  o.nsrc = nsrc;
  o.hf = (a.hf && nsrc >= 6) ? 1 : 0;
  o.nin = nsrc - 4 * o.hf;
  o.nrow_pad = s2_rows_padded(nsrc, a.ilv, a.hf);
  o.q5 = (o.nrow_pad - 4 * o.hf) / 5;
  o.nunit = o.nrow_pad * a.nseg;
  o.nsplit = min(a.nsplit, o.nunit);
  o.nfull = o.nunit - o.nsplit;
  o.lsub = max(a.lsub, 1);
  o.ntask = o.nfull + o.nsplit * ((a.L + o.lsub - 1) / o.lsub);
  o.inv_nsplit = __builtin_amdgcn_rcpf((float)max(o.nsplit, 1));
#ifdef S2X_STAGGER
  o.rot = o.nfull > 0 ? (int)((unsigned)wg % (unsigned)o.nfull) : 0;
#else
  o.rot = 0;
#endif
  return o;
}
__device__ __forceinline__ Unit s2_unit_of(const S2Args& a, const BandOrder& o, int tg) {
  Unit un;
  if (tg < o.nfull) {
    un.u = tg + o.rot;
    if (un.u >= o.nfull) un.u -= o.nfull;
    un.l0 = 0; un.nl = a.L;
    return un;
  }
  const int j = tg - o.nfull;
  const int part = s2_div_small(j, o.nsplit, o.inv_nsplit);
  un.u = o.nfull + (j - part * o.nsplit);
  un.l0 = part * o.lsub;
  un.nl = min(o.lsub, a.L - un.l0);
  return un;
}
// unit -> row of the band (may be a padding row >= nsrc) and segment
__device__ __forceinline__ int s2_unit_row(const S2Args& a, const BandOrder& o, int u,
                                           int& sg) {
  int yi = (int)(((float)u + 0.5f) * a.inv_nseg);
  sg = u - yi * a.nseg;
  int base = 0;
  if (o.hf) {
    // the first 4 * nseg units: rows 0, nsrc - 1, 1, nsrc - 2 of segment 0, then
    // of segment 1, ... (consecutive tickets merge into alternating tile rows)
    if (yi < 4) {
      if (a.hf == 2) return yi < 2 ? yi : o.nsrc - 4 + yi;  // (experiment: row-major)
      sg = u >> 2;
      const int hr = u & 3;
      return (hr & 1) ? o.nsrc - 1 - (hr >> 1) : (hr >> 1);
    }
    yi -= 4; base = 2;
  }
  int r = yi;
  if (a.ilv) {
    const int y5 = (int)(((float)yi + 0.5f) * 0.2f);
    r = (yi - 5 * y5) * o.q5 + y5;
  }
  return r < o.nin ? base + r : o.nsrc;  // (>= nsrc: a padding row)
}
// The unit a wave takes without a ticket (ticket == its wave index), as
// (row of the band, segment, first layer, layers); nl == 0: none.
struct Aim { int yr, sg, l0, nl; };
__device__ __forceinline__ Aim s2_own_unit(const S2Args& a, const BandOrder& o, int wave) {
  Aim r; r.yr = 0; r.sg = 0; r.l0 = 0; r.nl = 0;
  if (wave >= min(a.cap, o.ntask)) return r;
  const Unit un = s2_unit_of(a, o, wave);
  int sg;
  const int yr = s2_unit_row(a, o, un.u, sg);
  if (yr >= o.nsrc) return r;
  r.yr = yr; r.sg = sg; r.l0 = un.l0; r.nl = un.nl;
  return r;
}

__device__ __forceinline__ float s2_next_up(float t) {
  return __int_as_float(__float_as_int(t) + 1);
}
__device__ __forceinline__ float s2_next_down(float t) {
  return __int_as_float(__float_as_int(t) - 1);
}
// Smallest side weight w with fl(w * wy) > 1e-3f (sampling.py:218-222); see
// clamp_threshold in lsi_splat_stream.hip.
__device__ __forceinline__ float s2_clamp_threshold(float wy) {
  if (!(wy > 0.0f)) return __builtin_inff();
  float t = div_rn(1.0e-3f, wy);
  if (!(t < 4.0f)) return __builtin_inff();
  for (int k = 0; k < 8; ++k) {
    const float p = s2_next_down(t);
    if (p * wy > 1.0e-3f) t = p; else break;
  }
  for (int k = 0; k < 8; ++k) {
    if (!(t * wy > 1.0e-3f)) t = s2_next_up(t); else break;
  }
  return t;
}
__device__ __forceinline__ float s2_lane_below(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf,
                                                 0xf, true));
}
__device__ __forceinline__ float4 s2_fma4(float4 t, float4 v, float w) {
  t.x = __fmaf_rn(v.x, w, t.x); t.y = __fmaf_rn(v.y, w, t.y);
  t.z = __fmaf_rn(v.z, w, t.z); t.w = __fmaf_rn(v.w, w, t.w);
  return t;
}
__device__ __forceinline__ int s2_try1(unsigned a0) {
  int o;
  const int one = 1;
  asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(o) : "v"(a0), "v"(one) : "memory");
  return o;
}
__device__ __forceinline__ void s2_try2(unsigned a0, unsigned a1, int& o0, int& o1) {
  const int one = 1;
  asm volatile(
      "ds_wrxchg_rtn_b32 %0, %2, %4\n\tds_wrxchg_rtn_b32 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o0), "=&v"(o1)
      : "v"(a0), "v"(a1), "v"(one)
      : "memory");
}
__device__ __forceinline__ void s2_unlock(unsigned a0) {
  const int zero = 0;
  asm volatile("ds_write_b32 %0, %1" : : "v"(a0), "v"(zero) : "memory");
}

// Row locks (workgroup scope).
__device__ __forceinline__ void s2_lock_row(int* locks, int r, int lane) {
  if (lane == 0) {
    for (;;) {
      int expect = 0;
      if (__hip_atomic_compare_exchange_strong(&locks[r], &expect, 1,
                                               __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP))
        break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void s2_unlock_row(int* locks, int r, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0)
    __hip_atomic_store(&locks[r], 0, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The wave's queue of (tile cell, value) records -> tile.  CELL: under the
// cells' try-locks; else the caller holds the rows' locks.
template <bool CELL>
__device__ __forceinline__ void s2_apply_queue(float4* tile4, unsigned clk_addr,
                                               const float4* qv, const int* qc,
                                               int qn, int lane) {
  if (CELL) {
    for (int i0 = 0; i0 < qn; i0 += 64) {
      const int i = i0 + lane;
      bool pred = i < qn;
      const int tcell = pred ? qc[i] : 0;
      const float4 v = pred ? qv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      const unsigned la = clk_addr + (unsigned)tcell * 4u;
      while (__ballot(pred) != 0ull) {
        if (pred && s2_try1(la) == 0) {
          float4* e = tile4 + tcell;
          float4 t = *e;
          t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
          *e = t;
          S2_FENCE();
          s2_unlock(la);
          pred = false;
        }
      }
    }
    return;
  }
  float* extras = reinterpret_cast<float*>(tile4);
  for (int i = lane; i < qn; i += 64) {
    const float4 v = qv[i];
    float* e = extras + (long)qc[i] * 4;
    atomic_add_f32(e + 0, v.x);  // two records may name the same cell
    atomic_add_f32(e + 1, v.y);
    atomic_add_f32(e + 2, v.z);
    atomic_add_f32(e + 3, v.w);
  }
}

// Queue full in the middle of a task (rare): empty it into the tile now.  Out
// of line -- one copy instead of one per push site.  rowA / rowB: the tile
// rows the records can name (-1: none).  Everything is handed over as LDS byte
// offsets and used through address-space-3 pointers: generic pointers would
// turn the accesses into FLAT instructions, and flat fp32 atomics on the LDS
// aperture fault.
typedef __attribute__((address_space(3))) float LdsF;
typedef __attribute__((address_space(3))) int LdsI;
template <bool CELL>
__device__ __noinline__ void s2_flush_queue(unsigned tile_off, unsigned locks_off,
                                            unsigned clk_off, unsigned qv_off,
                                            unsigned qc_off, int qn, int lane,
                                            int rowA, int rowB) {
  LdsF* const tile = reinterpret_cast<LdsF*>(tile_off);
  LdsI* const locks = reinterpret_cast<LdsI*>(locks_off);
  LdsF* const qv = reinterpret_cast<LdsF*>(qv_off);
  LdsI* const qc = reinterpret_cast<LdsI*>(qc_off);
  if (CELL) {
    for (int i0 = 0; i0 < qn; i0 += 64) {
      const int i = i0 + lane;
      bool pred = i < qn;
      const int tcell = pred ? qc[i] : 0;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = pred ? qv[4 * i + k] : 0.0f;
      const unsigned la = clk_off + (unsigned)tcell * 4u;
      while (__ballot(pred) != 0ull) {
        if (pred && s2_try1(la) == 0) {
          LdsF* e = tile + 4 * tcell;
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] = e[k] + v[k];
          S2_FENCE();
          s2_unlock(la);
          pred = false;
        }
      }
    }
    return;
  }
  auto lock = [&](int r) {
    if (lane == 0) {
      for (;;) {
        int expect = 0;
        if (__hip_atomic_compare_exchange_strong(&locks[r], &expect, 1,
                                                 __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP))
          break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto unlock = [&](int r) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0)
      __hip_atomic_store(&locks[r], 0, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (rowA >= 0) lock(rowA);
  if (rowB >= 0) lock(rowB);
  for (int i = lane; i < qn; i += 64) {
    LdsF* e = tile + 4 * qc[i];
    // (two records may name the same cell)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __hip_atomic_fetch_add(e + k, qv[4 * i + k], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (rowB >= 0) unlock(rowB);
  if (rowA >= 0) unlock(rowA);
}

// Folded fields: several lanes of a wave name the same window cell and must add
// one after the other.  `cnt` is the wave's table of one byte per window cell
// (four cells per word, zero between uses): ONE returning integer LDS add per
// lane yields its rank among the lanes of its cell (a wave has 64 lanes: a byte
// never carries into its neighbour); the caller runs rounds k = 0, 1, ... in
// which the lanes of rank k add (distinct cells within a round), then clears
// the words it touched.  (Round 3 elected one lane per cell and round through a
// byte written and read back: two more LDS operations per round.)
__device__ __forceinline__ int s2_cell_rank(unsigned char* cnt, int cell, bool act) {
  if (!act) return -1;
  const unsigned sh = 8u * ((unsigned)cell & 3u);
  const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(cnt + (cell & ~3)), 1u << sh);
  return (int)((old >> sh) & 0xffu);
}
__device__ __forceinline__ void s2_cell_rank_reset(unsigned char* cnt, int cell, bool act) {
  if (act) *reinterpret_cast<unsigned*>(cnt + (cell & ~3)) = 0u;
}

// 16-byte output store by epilogue mode (S2Args.ep): 1 plain, 2 write-through
// (sc1), 3 non-temporal (what the launcher picks: see lsi_common.h)
__device__ __forceinline__ void s2_store4(float* p, float x, float y, float z, float w,
                                          int ep) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 v = {x, y, z, w};
  if (ep == 3) {
    __builtin_nontemporal_store(v, reinterpret_cast<v4*>(p));
  } else if (ep == 2) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  } else {
    *reinterpret_cast<v4*>(p) = v;
  }
}
struct Px { float4 d4, t0, t1, t2; };
typedef float s2_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 s2_ld_nt(const float* p) {
  const s2_f4 v = __builtin_nontemporal_load(reinterpret_cast<const s2_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// BOTH: lsi_splat_fwd_both -- the per-layer views AND the composed one from one
// sweep: one tile per layer in LDS, every item (one layer of a unit) merges its
// window into its layer's tile, the epilogue writes L + 1 views.
// (Tried in round 4 and dropped: layers outermost with two tiles -- the current
// layer's and the sum of the finished ones -- and a workgroup rendezvous between
// layers, which allows bands of 8 rows instead of 4: 257 us against 195 at
// config 3.  The rendezvous stops every wave's loads four times per band, and
// one-layer tickets cost more than their share: profiles/r04/both_layer_outer.txt.)
template <int NSETS, bool CELL, int MAXT, bool BOTH = false, bool PACK = false,
          bool RAGGED = false>
__global__ __launch_bounds__(MAXT) void splat_stream2_kernel(S2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = S2_RFL(tid >> 6);
  const int T = blockDim.x, NW = T >> 6;
  const int R = a.R, WMAX = a.wmax;
  const int Wt = a.Wt, Ht = a.Ht, H = a.H;
  // XCD-aware placement: workgroup i runs on XCD i % 8; every XCD gets a
  // contiguous run of bands (neighbouring bands share halo rows: same L2)
  int b, band;
  {
    const unsigned nwg = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7u, q = nwg >> 3, r8 = nwg & 7u;
    const unsigned base =
        xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
#ifdef S2X_ROUNDROBIN  // experiment: workgroup i keeps id i (XCD i % 8: an image's bands on all XCDs)
    const unsigned id = lin + 0 * base;
#else
    const unsigned id = base + (lin >> 3);
#endif
    b = s2_div_small((int)id, (int)gridDim.x, a.inv_gx);
    band = (int)id - b * (int)gridDim.x;
  }
  // band = target rows [row0, row0 + rows); it reads every source row that
  // touches them: floor(Y) in [row0 - 1, row0 + rows - 1]
  const int row0 = band * R;
  const int rows = min(R, Ht - row0);
  const int k_lo = row0 - 1, k_hi = row0 + rows - 1;

  // ---- LDS carve ---------------------------------------------------------
  const int WHS = ((WMAX / 2 + 15) & ~15) + 8;  // slots per window half
  const int WCELLS = 2 * WHS;
  float4* const rb_all = reinterpret_cast<float4*>(smem_raw);    // [NW][WCELLS]
  const int NT = BOTH ? a.L : 1;                                  // tiles
  float4* const tile4 = rb_all + NW * WCELLS;                     // [NT][R][Wt]
  Task* const task = reinterpret_cast<Task*>(tile4 + NT * R * Wt);  // [cap]
  TaskX* const taskx = reinterpret_cast<TaskX*>(task + a.cap);    // [cap]
  int* const ctl = reinterpret_cast<int*>(taskx + a.cap);         // [4]: ticket, table arrivals
  int* const locks = ctl + 4;                                     // [NT * R]
  const int Q = a.qcap;
  float4* const qv_all = reinterpret_cast<float4*>(ctl + ((4 + NT * R + 3) & ~3));
  int* const qc_all = reinterpret_cast<int*>(qv_all + NW * Q);
  unsigned char* const sc_all = reinterpret_cast<unsigned char*>(qc_all + NW * Q);
  int* const clk = reinterpret_cast<int*>(sc_all + ((NW * WMAX + 15) & ~15));  // CELL: [NT*R*Wt]
  const unsigned clk_addr = (unsigned)(uintptr_t)clk;
  float4* const rb = rb_all + wave * WCELLS;
  float4* const qv = qv_all + wave * Q;
  int* const qc = qc_all + wave * Q;
  unsigned char* const sc = sc_all + wave * WMAX;

#ifdef S2X_STAMPS
  long long* const stamp = a.stamps
      ? a.stamps + (((size_t)b * gridDim.x + band) * 16 + wave) * 8 : nullptr;
#define S2_STAMP(k) do { if (stamp && lane == 0) stamp[k] = (long long)wall_clock64(); } while (0)
#else
#define S2_STAMP(k)
#endif
  S2_STAMP(0);
  // rows 0 and 1 of M, one float per lane, asked for before anything else.
  // A vector load: its counter retires in order, so the first texture loads
  // below are issued behind it without waiting for it (scalar loads return out
  // of order: waiting for any kernel argument would wait for M as well).
  const float m_lane = a.M[16 * b + (lane & 7)];
  // ---- loader state ---------------------------------------------------------
  // (PACK: RGBD pixels, 16 floats per lane; the disparity pointer is unused)
  // Lanes past the end of a row (its last segment may be partial: W % 256 != 0)
  // read the row's last four pixels instead -- always inside the tensor -- and
  // are switched off where the item is computed.
  const float* const g_tex0 = a.tex + (long)b * a.tex_sb;
  const float* const g_disp0 = a.disp + (long)b * a.disp_sb;
  const int px_last = a.W - 4;
  const int px_own = RAGGED ? min(4 * lane, px_last) : 4 * lane;
  const float* const g_tex = g_tex0 + (PACK ? 4 : 3) * px_own;   // (harmless re-reads)
  const float* const g_disp = g_disp0 + px_own;
  const int tex_sl = a.tex_sl, disp_sl = a.disp_sl;
  const float* p_disp = g_disp;
  const float* p_tex = g_tex;
  int ld_left = 0, ld_done = 0, ld_slot = 0, ld_first = 0;
  auto aim = [&](int y, int sg, int l0) {
    const int px = RAGGED ? min(sg * SEG + 4 * lane, px_last) : sg * SEG + 4 * lane;
    p_disp = g_disp0 + (long)l0 * disp_sl + (long)y * a.disp_sy + px;
    p_tex = g_tex0 + (long)l0 * tex_sl + (long)y * a.tex_sy + (PACK ? 4 : 3) * px;
  };
#ifdef S2X_NT  // experiment: streamed inputs loaded with the non-temporal policy
#define S2_LD(p) s2_ld_nt(p)
#else
#define S2_LD(p) (*reinterpret_cast<const float4*>(p))
#endif
  auto load_layer = [&](Px& o) {
    if (PACK) {  // d4, t0, t1, t2 = the lane's pixels 0 .. 3 as (r, g, b, d)
      o.d4 = S2_LD(p_tex);
      o.t0 = S2_LD(p_tex + 4);
      o.t1 = S2_LD(p_tex + 8);
      o.t2 = S2_LD(p_tex + 12);
    } else {
      o.d4 = S2_LD(p_disp);
      o.t0 = S2_LD(p_tex);
      o.t1 = S2_LD(p_tex + 4);
      o.t2 = S2_LD(p_tex + 8);
      p_disp += disp_sl;
    }
    p_tex += tex_sl;
  };
  // ---- the wave's own first unit, aimed BEFORE the projection matrix is here --
  // M is one dependent scalar load from HBM away (~2 us cold).  For a
  // rectified pair with equal intrinsics Y(y) = (y + .5) s - .5: the wave aims
  // its first unit with that guess and has its loads in flight while M
  // arrives; a wrong guess (checked below against the exact row range) costs
  // one re-issue.
  Px set[NSETS];
  int tag[NSETS];
  int g_y = 0;
  Aim g_aim;
  {
    const float inv_s = __builtin_amdgcn_rcpf(a.s);
    const int ylo_g = (int)fminf(fmaxf(ceilf(((float)k_lo + 0.5f) * inv_s - 0.5f), 0.0f), (float)H);
    const int yhi_g = (int)fminf(fmaxf(ceilf(((float)k_hi + 1.5f) * inv_s - 0.5f), 0.0f), (float)H) - 1;
    const BandOrder bg = s2_band_order(a, max(0, yhi_g - ylo_g + 1), b * 29 + band * 13);
    g_aim = s2_own_unit(a, bg, wave);
    g_y = ylo_g + g_aim.yr;
    if (g_aim.nl) aim(g_y, g_aim.sg, g_aim.l0);
    // (always NSETS items of loads -- sets without a layer re-read the first
    // bytes: the number of loads in flight behind M is then known statically,
    // and waiting for M does not wait for them)
#pragma unroll
    for (int k = 0; k < NSETS; ++k) {
      if (k == g_aim.nl) { p_disp = g_disp; p_tex = g_tex; }
      load_layer(set[k]);
      if (k >= g_aim.nl) { p_disp = g_disp; p_tex = g_tex; }
    }
  }
  // ---- LDS init and the opening barrier: nothing here needs M ---------------
  {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ncl = BOTH ? NT * R * Wt : rows * Wt;
    for (int i = tid; i < NW * WCELLS + ncl; i += T) rb_all[i] = z4;  // windows + tiles
    for (int i = tid; i < 4 + NT * R; i += T) ctl[i] = (i == 0) ? NW : 0;  // tickets, arrivals, locks
    if (CELL)
      for (int i = tid; i < ncl; i += T) clk[i] = 0;
    for (int i = tid; i < (NW * WMAX + 3) / 4; i += T)  // the waves' cell-rank tables
      reinterpret_cast<unsigned*>(sc_all)[i] = 0u;
  }
  __syncthreads();
  float m[8];
  {
    float ml = m_lane;
    asm volatile("" : "+v"(ml));  // (M is not waited for before the barrier)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      m[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ml), k));
  }
  const float s = a.s;
  const float xmax = (float)Wt - 1.0f, ymax = (float)Ht - 1.0f;

  // Row-uniform target coordinate Y(y) = q1 * s - 0.5 (exact op order; the
  // normaliser is exactly 1 here)
  auto row_Y = [&](int y) {
    const float py = (float)y + 0.5f;
    return mrow(m, 1, 0.5f, py, 0.0f) * s - 0.5f;
  };

  // ---- source rows of the band (every wave on its own: no barrier) --------
  int y_lo = H, y_hi = -1;
  bool ranged = false;
  {
    const float YA = row_Y(0), YB = row_Y(H - 1);
    const bool ok = finite_f(YA) && finite_f(YB) && fabsf(YA) < 65536.0f &&
                    fabsf(YB) < 65536.0f &&
                    (H == 1 || (YB - YA) >= 0.0625f * (float)(H - 1));
    if (ok) {
      const float inv_s = __builtin_amdgcn_rcpf(s);  // estimates only
      const float inv_m5 = __builtin_amdgcn_rcpf(m[5]);
      auto lower = [&](float k, bool& good) {
        const float t = (k + 0.5f) * inv_s;
        const float py = (t - m[6]) * inv_m5;
        const int c = finite_f(py) ? (int)fminf(fmaxf(ceilf(py - 0.5f), 0.0f),
                                                (float)H)
                                   : 0;
        const int base = max(0, min(c - 32, H - 64));
        const int y = base + lane;
        const bool pass_ = y >= H || row_Y(y) >= k;
        const unsigned long long mask = __ballot(pass_);
        const int p = mask ? __builtin_ctzll(mask) : 64;
        good = good && mask == (p < 64 ? (~0ull << p) : 0ull) &&
               (p > 0 || base == 0) && (p < 64 || base + 64 >= H);
        return min(base + p, H);
      };
      bool good = true;
      const int lo = lower((float)k_lo, good);
      const int hi = lower((float)(k_hi + 1), good) - 1;
      if (good) { y_lo = lo; y_hi = hi; ranged = true; }
    }
  }
  if (!S2_RFL(ranged ? 1 : 0)) {  // maps too flat / decreasing: scan, per wave
    int lo = H, hi = -1;
    for (int y = lane; y < H; y += 64) {
      const float Y = row_Y(y);
      if (!finite_f(Y)) continue;
      const float y0 = floorf(Y);
      if (y0 >= (float)k_lo && y0 <= (float)k_hi) { lo = min(lo, y); hi = max(hi, y); }
    }
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o));
      hi = max(hi, __shfl_xor(hi, o));
    }
    y_lo = lo; y_hi = hi;
  }
  y_lo = S2_RFL(y_lo); y_hi = S2_RFL(y_hi);
  const int nsrc = (y_hi >= y_lo) ? (y_hi - y_lo + 1) : 0;
  const BandOrder bo = s2_band_order(a, nsrc, b * 29 + band * 13);
  const int ntask = bo.ntask;
  // the table holds a.cap tickets: one chunk unless the band has more source
  // rows than the planner assumed (it does not see the matrices)
  const int CAPT = a.cap;
  int chunk0 = 0, nchunk = min(CAPT, ntask);

  // One item of loads into dst; returns its tag: -1 (none), or task slot |
  // bit 20 (first layer of the task) | bit 21 (last).  Always exactly one
  // load per input array, so that every path has the same number in flight.
  // Before its first ticket a wave makes sure that every wave has filled its
  // share of the task table (ctl[1] counts them).
  int synced = 0;
  auto table_ready = [&]() {
    if (!synced) {
      if (lane == 0) {
        while (__hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE,
                                 __HIP_MEMORY_SCOPE_WORKGROUP) < NW)
          __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      S2_STAMP(3);
      synced = 1;
    }
  };
  auto issue = [&](Px& dst) -> int {
    if (S2_RFL(ld_left) == 0 && S2_RFL(ld_done) == 0) {
      table_ready();
      int t = 0;
      if (lane == 0) t = atomicAdd(&ctl[0], 1);
      const int sl = S2_RFL(t);
      if (sl >= nchunk) {
        ld_done = 1;
      } else {
        const Task ta = task[sl];
        // rows masked at the border add nothing: not even started
        if (S2_RFL((ta.wy0 != 0.0f || ta.wy1 != 0.0f) ? 1 : 0)) {
          const int yx = S2_RFL(taskx[sl].yx);
          const Unit un = s2_unit_of(a, bo, chunk0 + sl);
          aim(yx & 0xffff, yx >> 16, un.l0);
          ld_left = un.nl; ld_slot = sl; ld_first = 1 << 20;
        }
      }
    }
    int tag = -1;
    const bool have = S2_RFL(ld_left) != 0;
    if (!have) { p_disp = g_disp; p_tex = g_tex; }  // harmless re-read
    load_layer(dst);
    if (have) {
      ld_left = S2_RFL(ld_left) - 1;
      tag = ld_slot | ld_first | (ld_left == 0 ? (1 << 21) : 0);
      ld_first = 0;
    }
    return S2_RFL(tag);
  };

  // ---- the exact first unit: keep the loads in flight, or aim again ---------
  {
    const Aim r = s2_own_unit(a, bo, wave);
    const int r_y = y_lo + r.yr;
    const bool same = r.nl == g_aim.nl &&
                      (r.nl == 0 || (r_y == g_y && r.sg == g_aim.sg && r.l0 == g_aim.l0));
    if (!S2_RFL(same ? 1 : 0) && r.nl) {
      aim(r_y, r.sg, r.l0);
#pragma unroll
      for (int k = 0; k < NSETS; ++k) {
        if (k == r.nl) { p_disp = g_disp; p_tex = g_tex; }
        load_layer(set[k]);
        if (k >= r.nl) { p_disp = g_disp; p_tex = g_tex; }
      }
    }
    const int issued = min(NSETS, r.nl);
    ld_left = r.nl - issued; ld_slot = wave; ld_first = 0;
#pragma unroll
    for (int k = 0; k < NSETS; ++k)
      tag[k] = k < issued ? (wave | (k == 0 ? (1 << 20) : 0) |
                             (k == r.nl - 1 ? (1 << 21) : 0))
                          : -1;
  }
  S2_STAMP(1);
  // table entry of ticket tg (exact row geometry, clamp threshold, window)
  auto make_task = [&](const int tg, Task& ta, TaskX& tx) {
    ta.row0 = -1000000; ta.wy0 = 0.f; ta.wy1 = 0.f; ta.win = 0;
    tx.yx = 0; tx.tmin = __builtin_inff();
    int sg;
    const int yr = s2_unit_row(a, bo, s2_unit_of(a, bo, tg).u, sg);
    const int y = y_lo + yr;
    const float Y = row_Y(y);
    if (yr < bo.nsrc && finite_f(Y) && fabsf(Y) < 1.0e7f) {
      const Axis ay = splat_axis(Y, ymax);
      ta.row0 = (int)floorf(Y) - row0;
      ta.wy0 = ay.w0;
      ta.wy1 = ay.w1;
      tx.yx = y | (sg << 16);
      const float wymin =
          (ay.w0 == 0.f) ? ay.w1 : ((ay.w1 == 0.f) ? ay.w0 : fminf(ay.w0, ay.w1));
      tx.tmin = s2_clamp_threshold(wymin);
      // window hint: cells reachable for d in [0, max_disp] on the segment
      const int xs = sg * SEG;
      const float py = (float)y + 0.5f;
      auto x_of = [&](int xx, float dd) {
        return mrow(m, 0, (float)xx + 0.5f, py, dd) * s - 0.5f;
      };
      float lo = __builtin_inff(), hi = -__builtin_inff();
      if (m[0] > 0.0f) {
        const bool neg = m[3] < 0.0f;
        lo = x_of(xs, neg ? a.max_disp : 0.0f);
        hi = x_of(xs + SEG - 1, neg ? 0.0f : a.max_disp);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float X = x_of((c & 1) ? (xs + SEG - 1) : xs,
                               (c & 2) ? a.max_disp : 0.0f);
          lo = fminf(lo, X); hi = fmaxf(hi, X);
        }
      }
      if (finite_f(lo) && finite_f(hi) && fabsf(lo) < 1.0e7f &&
          fabsf(hi) < 1.0e7f) {
        // cells [wlo, wlo + wwin): not confined to the image (cells outside
        // are dummies the merge drops: the reference's border rule)
        const int c_lo = max((int)floorf(lo) - 1, -32000);
        const int c_hi = min((int)floorf(hi) + 4, 32000);
        const int wwin = max(0, min(WMAX, c_hi - c_lo + 1));
        ta.win = (c_lo + 32768) | (wwin << 16);
      }
    }
  };
  // tickets [c0 + first, c0 + n): one ticket per thread
  auto fill_table = [&](const int c0, const int first, const int n) {
    for (int sl = first + tid; sl < n; sl += T) {
      Task ta; TaskX tx;
      make_task(c0 + sl, ta, tx);
      task[sl] = ta;
      taskx[sl] = tx;
    }
  };
  // The wave's own table entry, its share of the other tickets, and "arrived":
  // the table is complete once every wave has (checked before a wave's first
  // ticket, by which time it normally is: no wave waits).
  {
    if (wave < nchunk) {
      Task ta; TaskX tx;
      make_task(wave, ta, tx);
      if (lane == 0) { task[wave] = ta; taskx[wave] = tx; }
    }
    fill_table(0, NW, nchunk);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0)
      __hip_atomic_fetch_add(&ctl[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  S2_STAMP(2);

  // ---- per-wave compute state ---------------------------------------------------
  const float m3 = m[3];
  const float zA = a.zA, zB = a.zB, max_disp = a.max_disp;
  int slot = 0, t_row0 = 0, t_wlo = 0, t_wwin = 0, has_max = 0, rmax_row = 0;
  int t_lay = 0;  // BOTH: first tile row of the item's layer (layer * R)
  int fast_ok = 0, fastA_ok = 0, win_ok = 0;
  unsigned wspan = 0u, wspanA = 0u;
  float tmin = 0.f, wy0 = 0.f, wy1 = 0.f, wymin = 0.f, wymax = 0.f, wlo_f = 0.f;
  int use_a = 0, use_b = 0;
  float qb[4] = {0.f, 0.f, 0.f, 0.f};
  bool lane_dead = false;   // this lane's pixels of the current unit are past the row end
  int qn = 0;
  // items of this wave: all in the low half, those on routes B' / C in the high
  // half (s2_adapt; a scalar register)
  unsigned n_items = 0u;
#ifdef S2X_STAMPS
  unsigned route_n[4] = {0u, 0u, 0u, 0u};  // items by route: A, B, B' (folded), C (general)
#define S2_ROUTE(k) do { route_n[k] += 1u; n_items = S2_RFL(n_items + (((k) >= 2) ? 0x10001u : 1u)); } while (0)
#else
#define S2_ROUTE(k) (n_items = S2_RFL(n_items + (((k) >= 2) ? 0x10001u : 1u)))
#endif
  // merge: the lane's window slots (cells lane, lane + 64, ...)
  const int mslot = (lane >> 1) + (lane & 1) * WHS;

  // (tile cell, value) records for corners the window cannot hold
  auto flush = [&]() {
    s2_flush_queue<CELL>((unsigned)(uintptr_t)tile4, (unsigned)(uintptr_t)locks,
                         clk_addr, (unsigned)(uintptr_t)qv, (unsigned)(uintptr_t)qc,
                         qn, lane, use_a ? t_lay + t_row0 : -1,
                         use_b ? t_lay + t_row0 + 1 : -1);
    qn = 0;
  };
  auto push = [&](bool pred, int tcell, float4 val) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const int n = __builtin_popcountll(mask);
    if (qn + n > Q) flush();  // full: empty it into the tile now (rare)
    const int rank = (int)__builtin_amdgcn_mbcnt_hi(
        (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (n > Q) {  // (a queue of fewer than 64 records: in rounds of Q lanes)
      for (int base = 0; base < n; base += Q) {
        if (pred && rank >= base && rank < base + Q) {
          qv[rank - base] = val; qc[rank - base] = tcell;
        }
        qn = min(Q, n - base);
        flush();
      }
      return;
    }
    if (pred) { qv[qn + rank] = val; qc[qn + rank] = tcell; }
    qn += n;
  };
  // exact 4-corner footprint of one pixel (sampling.py:193-222) for lanes whose
  // cells are not inside the window
  auto push_corners = [&](bool pred, float4 V, float x0, float gx, float fx) {
    const float x1 = x0 + 1.0f;
    const float x0s = fminf(fmaxf(x0, 0.0f), xmax);
    const float x1s = fminf(fmaxf(x1, 0.0f), xmax);
    const float wx[2] = {(x0 == x0s) ? gx : 0.0f, (x1 == x1s) ? fx : 0.0f};
    const int cx[2] = {(int)x0s, (int)x1s};
    const float wy[2] = {wy0, wy1};
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const int r = t_row0 + (k >> 1);
      const float c = ((k & 1) ? wx[1] : wx[0]) * ((k >> 1) ? wy[1] : wy[0]);
      const bool ok = pred && (c > 1.0e-3f) && r >= 0 && r < rows;
      push(ok, (t_lay + r) * Wt + ((k & 1) ? cx[1] : cx[0]),
           make_float4(V.x * c, V.y * c, V.z * c, V.w * c));
    }
  };

  auto item = [&](Px& cur, const int tg_) -> int {
    const bool live = tg_ >= 0;
    if (live && (tg_ & (1 << 20))) {  // ---- the item starts a task ----------
      slot = tg_ & 0xfffff;
      t_lay = BOTH ? s2_unit_of(a, bo, chunk0 + slot).l0 * R : 0;
      const Task ta = task[slot];
      const TaskX tx = taskx[slot];
      t_row0 = S2_RFL(ta.row0);
      const int t_win = S2_RFL(ta.win);
      t_wlo = (t_win & 0xffff) - 32768; t_wwin = t_win >> 16;
      const int yx = S2_RFL(tx.yx);
      const int y = yx & 0xffff, xs = (yx >> 16) * SEG;
      if (RAGGED) lane_dead = xs + 4 * lane >= a.W;
      tmin = tx.tmin;
      const float py = (float)y + 0.5f;
      const float pym01 = py * m[1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float px = (float)(xs + 4 * lane + i) + 0.5f;
        qb[i] = (px * m[0] + pym01) + m[2];
      }
      wy0 = ta.wy0; wy1 = ta.wy1;
      wymin = (wy0 == 0.f) ? wy1 : ((wy1 == 0.f) ? wy0 : fminf(wy0, wy1));
      wymax = fmaxf(wy0, wy1);
      wlo_f = (float)t_wlo;
      const int rmax = t_row0 + ((wy1 > wy0) ? 1 : 0);
      has_max = S2_RFL(((wymax != wymin) && rmax >= 0 && rmax < rows) ? 1 : 0);
      rmax_row = S2_RFL(rmax);
      wspan = (unsigned)max(t_wwin - 2, 0);
      win_ok = t_wwin >= 2 ? 1 : 0;
      wspanA = (unsigned)max(t_wwin - 4, 0);
      fast_ok = S2_RFL((win_ok && tmin <= 0.5f) ? 1 : 0);
      fastA_ok = S2_RFL((t_wwin >= 4 && tmin <= 0.5f) ? 1 : 0);
      use_a = S2_RFL((wy0 != 0.f && t_row0 >= 0 && t_row0 < rows) ? 1 : 0);
      use_b = S2_RFL((wy1 != 0.f && t_row0 + 1 >= 0 && t_row0 + 1 < rows) ? 1 : 0);
    }
    float x0v[4], w0v[4], w1v[4];
    float4 Vv[4];  // route A: the lane's 4 cell sums; else V of its 4 pixels
    int cl0 = 0, routeA = 0;
    if (live) {
      const float dvn[4] = {cur.d4.x, cur.d4.y, cur.d4.z, cur.d4.w};
      const float dvp[4] = {cur.d4.w, cur.t0.w, cur.t1.w, cur.t2.w};
      const float txn[12] = {cur.t0.x, cur.t0.y, cur.t0.z, cur.t0.w,
                             cur.t1.x, cur.t1.y, cur.t1.z, cur.t1.w,
                             cur.t2.x, cur.t2.y, cur.t2.z, cur.t2.w};
      const float txp[12] = {cur.d4.x, cur.d4.y, cur.d4.z, cur.t0.x,
                             cur.t0.y, cur.t0.z, cur.t1.x, cur.t1.y,
                             cur.t1.z, cur.t2.x, cur.t2.y, cur.t2.z};
      const float (&dv0)[4] = PACK ? dvp : dvn;
      const float (&tx0_)[12] = PACK ? txp : txn;
      float dv[4], tx_[12];
#pragma unroll
      for (int k = 0; k < 4; ++k) dv[k] = dv0[k];
#pragma unroll
      for (int k = 0; k < 12; ++k) tx_[k] = tx0_[k];
      if (RAGGED) {  // pixels past the row end: disparity 0 (weight 0), colour 0
#pragma unroll
        for (int k = 0; k < 4; ++k) dv[k] = lane_dead ? 0.0f : dv[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) tx_[k] = lane_dead ? 0.0f : tx_[k];
      }
      if (BOTH && a.out_disp) {  // the disparity pass: the "colour" is (d, 0, 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // (a pixel with zero weight or a non-finite disparity adds nothing:
          // never NaN * 0)
          tx_[3 * i] = (dv[i] > 0.0f && dv[i] < __builtin_inff()) ? dv[i] : 0.0f;
          tx_[3 * i + 1] = 0.f; tx_[3 * i + 2] = 0.f;
        }
      }
      float pwv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // q0 = ((px*m00 + py*m01) + m02) + d*m03, each op rounded; n' == 1
        const float q0 = qb[i] + dv[i] * m3;
        const float X = q0 * s - 0.5f;
        const float x0 = floorf(X);
        const float fx = X - x0;
        // (x0 + 1) - X == 1 - fx exactly for X >= 0; for X < 0 the left cell
        // is outside the image and its weight is never used
        const float gx = 1.0f - fx;
        // helpers.py:180-193 with D = d: exp((clip(d/max,0,1) - 0.5)*scale)
        const float dc = __builtin_amdgcn_fmed3f(dv[i], 0.0f, max_disp);
        const float e = __builtin_amdgcn_exp2f(__fmaf_rn(dc, zA, zB));
#ifdef S2X_NOEXP
        pwv[i] = dc;
        (void)e;
#else
        pwv[i] = dv[i] > 0.0f ? e : 0.0f;  // (NaN disparity -> weight 0)
#endif
        x0v[i] = x0; w0v[i] = gx; w1v[i] = fx;
      }
      // ---- route A: the lane's 4 pixels land in cells cl0 .. cl0 + 3 --------
      cl0 = (int)(x0v[0] - wlo_f);
      int dl[4];
      dl[0] = 0;
      unsigned long long regA =
          __ballot((unsigned)cl0 <= wspanA) &
          (__ballot(x0v[0] > s2_lane_below(x0v[0])) | 1ull);
#pragma unroll
      for (int i = 1; i < 4; ++i) {
        dl[i] = (int)(x0v[i] - x0v[0]);
        regA &= __ballot((unsigned)dl[i] <= 2u);
      }
      routeA = S2_RFL(((regA == ~0ull) && fastA_ok) ? 1 : 0);
#ifdef S2X_NOSUMS
      if (routeA) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          Vv[k] = make_float4(tx_[3 * k] * pwv[k] * w0v[k], tx_[3 * k + 1] * pwv[k] * w1v[k],
                              tx_[3 * k + 2] * pwv[k], pwv[k]);
      } else
#endif
      if (routeA) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Vv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float w0 = w0v[i], w1 = w1v[i];
          const float4 V = make_float4(tx_[3 * i] * pwv[i], tx_[3 * i + 1] * pwv[i],
                                       tx_[3 * i + 2] * pwv[i], pwv[i]);
          // clamp (at most the smaller side: tmin <= 0.5 on this route)
#ifdef S2X_NOCLAMP
          if (false) {
#else
          if (__ballot(!(fminf(w0, w1) >= tmin)) != 0ull) {
#endif
            const bool c0 = !(w0 >= tmin), c1 = !(w1 >= tmin);
            if (has_max) {
              const float kq = (c0 ? w0 : w1) * wymax;
              const int cellq = t_wlo + cl0 + dl[i] + (c0 ? 0 : 1);
              push((c0 || c1) && kq > 1.0e-3f && (unsigned)cellq < (unsigned)Wt,
                   (t_lay + rmax_row) * Wt + cellq,
                   make_float4(V.x * kq, V.y * kq, V.z * kq, V.w * kq));
            }
            if (c0) w0 = 0.0f;
            if (c1) w1 = 0.0f;
          }
          if (i == 0) {
            Vv[0] = make_float4(V.x * w0, V.y * w0, V.z * w0, V.w * w0);
            Vv[1] = make_float4(V.x * w1, V.y * w1, V.z * w1, V.w * w1);
          } else {
            const bool e0 = dl[i] == 0, e1 = dl[i] == 1, e2 = dl[i] == 2;
            const float a0 = e0 ? w0 : 0.0f;
            const float a1 = e1 ? w0 : (e0 ? w1 : 0.0f);
            const float a2 = e2 ? w0 : (e1 ? w1 : 0.0f);
            const float a3 = e2 ? w1 : 0.0f;
            Vv[0] = s2_fma4(Vv[0], V, a0);
            Vv[1] = s2_fma4(Vv[1], V, a1);
            Vv[2] = s2_fma4(Vv[2], V, a2);
            Vv[3] = s2_fma4(Vv[3], V, a3);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          Vv[i] = make_float4(tx_[3 * i] * pwv[i], tx_[3 * i + 1] * pwv[i],
                              tx_[3 * i + 2] * pwv[i], pwv[i]);
      }
      // pinned: the projection is not sunk below the loads, which then
      // overwrite dead registers
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("" : "+v"(Vv[i].x), "+v"(Vv[i].y), "+v"(Vv[i].z),
                          "+v"(Vv[i].w), "+v"(x0v[i]), "+v"(w0v[i]), "+v"(w1v[i]));
    }
    const int next_tag = issue(cur);  // the set takes the item NSETS ahead
    if (live) {
      if (routeA) {
        S2_ROUTE(0);
        const int par = cl0 & 1, hlf = cl0 >> 1;
        float4* ce = rb + hlf + par * WHS;              // cell cl0 (and cl0 + 2)
        float4* co = rb + hlf + par + (1 - par) * WHS;  // cell cl0 + 1 (and + 3)
        float4 t;
#ifndef S2X_NOWIN
        t = ce[0]; t.x += Vv[0].x; t.y += Vv[0].y; t.z += Vv[0].z; t.w += Vv[0].w; ce[0] = t;
        S2_FENCE();
        t = co[0]; t.x += Vv[1].x; t.y += Vv[1].y; t.z += Vv[1].z; t.w += Vv[1].w; co[0] = t;
        S2_FENCE();
        t = ce[1]; t.x += Vv[2].x; t.y += Vv[2].y; t.z += Vv[2].z; t.w += Vv[2].w; ce[1] = t;
        S2_FENCE();
        t = co[1]; t.x += Vv[3].x; t.y += Vv[3].y; t.z += Vv[3].z; t.w += Vv[3].w; co[1] = t;
#else
        t = ce[0]; t.x += Vv[0].x + Vv[1].x + Vv[2].x + Vv[3].x; t.y += Vv[0].y + Vv[1].y + Vv[2].y + Vv[3].y;
        t.z += Vv[0].z + Vv[1].z + Vv[2].z + Vv[3].z; t.w += Vv[0].w + Vv[1].w + Vv[2].w + Vv[3].w;
        if (t.x == 123.456f) ce[0] = t;
        (void)co;
#endif
        S2_FENCE();
      }
#ifndef S2X_ONLYA
      else {
        // every pixel inside the window (but not route A's pattern: folded or
        // steep disparity fields): per pixel, unrolled -- route B when floor(X)
        // increases strictly across the wave (distinct cells), else B': the
        // lanes of a cell elected one at a time through a byte table
        int clv[4];
        unsigned long long regular = ~0ull, inwin = ~0ull;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          clv[i] = (int)(x0v[i] - wlo_f);
          inwin &= __ballot((unsigned)clv[i] <= wspan);
          regular &= __ballot(x0v[i] > s2_lane_below(x0v[i])) | 1ull;
        }
        if (S2_RFL((inwin == ~0ull && fast_ok) ? 1 : 0)) {
          const bool fold = regular != ~0ull;
          S2_ROUTE(fold ? 2 : 1);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float w0 = w0v[i], w1 = w1v[i];
            const float4 V = Vv[i];
            // w0 + w1 = 1 and tmin <= 0.5: at most the smaller side is clamped
            if (__ballot(!(fminf(w0, w1) >= tmin)) != 0ull) {
              const bool c0 = !(w0 >= tmin), c1 = !(w1 >= tmin);
              if (has_max) {
                const float kq = (c0 ? w0 : w1) * wymax;
                const int cellq = t_wlo + clv[i] + (c0 ? 0 : 1);
                push((c0 || c1) && kq > 1.0e-3f && (unsigned)cellq < (unsigned)Wt,
                     (t_lay + rmax_row) * Wt + cellq,
                     make_float4(V.x * kq, V.y * kq, V.z * kq, V.w * kq));
              }
              if (c0) w0 = 0.0f;
              if (c1) w1 = 0.0f;
            }
            float4* cell = rb + (clv[i] >> 1) + (clv[i] & 1) * WHS;
            float4* cell1 = rb + ((clv[i] + 1) >> 1) + ((clv[i] + 1) & 1) * WHS;
            if (!fold) {
              *cell = s2_fma4(*cell, V, w0);
              S2_FENCE();
              *cell1 = s2_fma4(*cell1, V, w1);
              S2_FENCE();
            } else {
              // the lanes of a cell take turns: one returning LDS add gives every
              // lane its rank among them (s2_cell_rank), round k is rank k's
              const int rank = s2_cell_rank(sc, clv[i], true);
              for (int k = 0; __ballot(rank >= k) != 0ull; ++k) {
                if (rank == k) {
                  *cell = s2_fma4(*cell, V, w0);
                  S2_FENCE();
                  *cell1 = s2_fma4(*cell1, V, w1);
                }
                S2_FENCE();
              }
              s2_cell_rank_reset(sc, clv[i], true);
            }
          }
        } else {
        S2_ROUTE(3);
        // ---- per pixel (one copy of the code; the pixel's values are
        // selected).  In-window lanes whose cells are distinct add directly;
        // folded fields elect the lanes of a cell one at a time through a byte
        // table; lanes outside the window take the exact 4-corner path.
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
          auto pick = [&](float a0, float a1, float a2, float a3) {
            return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
          };
          const float x0 = pick(x0v[0], x0v[1], x0v[2], x0v[3]);
          float w0 = pick(w0v[0], w0v[1], w0v[2], w0v[3]);
          float w1 = pick(w1v[0], w1v[1], w1v[2], w1v[3]);
          const float4 V = make_float4(
              pick(Vv[0].x, Vv[1].x, Vv[2].x, Vv[3].x),
              pick(Vv[0].y, Vv[1].y, Vv[2].y, Vv[3].y),
              pick(Vv[0].z, Vv[1].z, Vv[2].z, Vv[3].z),
              pick(Vv[0].w, Vv[1].w, Vv[2].w, Vv[3].w));
          const int cl = (int)(x0 - wlo_f);
          const bool inw = ((unsigned)cl <= wspan) && win_ok;
          const unsigned long long inw_mask = __ballot(inw);
          const unsigned long long mono_ok =
              __ballot(x0 > s2_lane_below(x0)) | 1ull;
          // clamped sides: !(p > 1e-3) is also true for NaN weights
          const bool c0 = !(w0 * wymin > 1.0e-3f);
          const bool c1 = !(w1 * wymin > 1.0e-3f);
          if (has_max) {
            const float k0 = w0 * wymax, k1 = w1 * wymax;
            const int cm = (t_lay + rmax_row) * Wt + t_wlo + cl;
            push(inw && c0 && k0 > 1.0e-3f &&
                     (unsigned)(t_wlo + cl) < (unsigned)Wt, cm,
                 make_float4(V.x * k0, V.y * k0, V.z * k0, V.w * k0));
            push(inw && c1 && k1 > 1.0e-3f &&
                     (unsigned)(t_wlo + cl + 1) < (unsigned)Wt, cm + 1,
                 make_float4(V.x * k1, V.y * k1, V.z * k1, V.w * k1));
          }
          if (inw_mask != ~0ull)
            push_corners(!inw && V.w != 0.0f, V, x0, w0, w1);
          if (c0) w0 = 0.0f;
          if (c1) w1 = 0.0f;
          float4* cell = rb + (cl >> 1) + (cl & 1) * WHS;
          float4* cell1 = rb + ((cl + 1) >> 1) + ((cl + 1) & 1) * WHS;
          if (mono_ok == ~0ull) {
            if (inw) {
              *cell = s2_fma4(*cell, V, w0);
              S2_FENCE();
              *cell1 = s2_fma4(*cell1, V, w1);
            }
            S2_FENCE();
          } else if (inw_mask != 0ull) {
            const int rank = s2_cell_rank(sc, cl, inw);
            for (int k = 0; __ballot(rank >= k) != 0ull; ++k) {
              if (rank == k) {
                *cell = s2_fma4(*cell, V, w0);
                S2_FENCE();
                *cell1 = s2_fma4(*cell1, V, w1);
              }
              S2_FENCE();
            }
            s2_cell_rank_reset(sc, cl, inw);
          }
        }
        }
      }
#endif
      if (BOTH || (tg_ & (1 << 21))) {
        // ---- last layer done (BOTH: every layer): window -> two tile rows -----
#ifndef S2X_NOLOCK
        if (!CELL) {  // ascending order: no deadlock
          if (use_a) s2_lock_row(locks, t_lay + t_row0, lane);
          if (use_b) s2_lock_row(locks, t_lay + t_row0 + 1, lane);
        }
#endif
        if (qn != 0) {
          s2_apply_queue<CELL>(tile4, clk_addr, qv, qc, qn, lane);
          qn = 0;
        }
        float4* const trow = tile4 + (long)(t_lay + t_row0) * Wt + t_wlo + lane;
        float4* const wrow = rb + mslot;
        const int c_in = t_wlo + lane;
        if (CELL) {
          const unsigned lrow = clk_addr + (unsigned)((t_lay + t_row0) * Wt + t_wlo + lane) * 4u;
          for (int c = 0; c + lane < t_wwin; c += 64) {
            float4* wc = wrow + (c >> 1);
            const float4 v = *wc;
            *wc = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool inside = (unsigned)(c_in + c) < (unsigned)Wt;
            bool na = use_a && inside, nb = use_b && inside;
            const unsigned la = lrow + (unsigned)c * 4u, lb = la + (unsigned)Wt * 4u;
            while (__ballot(na || nb) != 0ull) {
              if (na && nb) {
                int oa, ob;
                s2_try2(la, lb, oa, ob);
                if (oa == 0) { trow[c] = s2_fma4(trow[c], v, wy0); }
                if (ob == 0) { trow[Wt + c] = s2_fma4(trow[Wt + c], v, wy1); }
                S2_FENCE();
                if (oa == 0) { s2_unlock(la); na = false; }
                if (ob == 0) { s2_unlock(lb); nb = false; }
              } else if (na) {
                if (s2_try1(la) == 0) {
                  trow[c] = s2_fma4(trow[c], v, wy0);
                  S2_FENCE();
                  s2_unlock(la);
                  na = false;
                }
              } else if (nb) {
                if (s2_try1(lb) == 0) {
                  trow[Wt + c] = s2_fma4(trow[Wt + c], v, wy1);
                  S2_FENCE();
                  s2_unlock(lb);
                  nb = false;
                }
              }
            }
          }
        } else {
          for (int c = 0; c + lane < t_wwin; c += 64) {
            float4* wc = wrow + (c >> 1);
            const float4 v = *wc;
            *wc = make_float4(0.f, 0.f, 0.f, 0.f);  // ready for the next task
            const bool inside = (unsigned)(c_in + c) < (unsigned)Wt;
#ifndef S2X_NOMERGE
            if (use_a && inside) trow[c] = s2_fma4(trow[c], v, wy0);
            if (use_b && inside) trow[Wt + c] = s2_fma4(trow[Wt + c], v, wy1);
#else
            if (inside && v.x == 123.456f) trow[c] = v;
#endif
          }
#ifndef S2X_NOLOCK
          if (use_b) s2_unlock_row(locks, t_lay + t_row0 + 1, lane);
          if (use_a) s2_unlock_row(locks, t_lay + t_row0, lane);
#endif
        }
      }
      if (BOTH) t_lay += R;  // the unit's next item is its next layer
    }
    return next_tag;
  };

  for (;;) {
    for (;;) {
      int any = ld_done == 0;
#pragma unroll
      for (int k = 0; k < NSETS; ++k) any |= tag[k] >= 0;
      if (!S2_RFL(any ? 1 : 0)) break;
#pragma unroll
      for (int k = 0; k < NSETS; ++k) tag[k] = item(set[k], tag[k]);
    }
    chunk0 += CAPT;
    if (chunk0 >= ntask) break;
    __syncthreads();  // (every wave is done with this chunk's table)
    nchunk = min(CAPT, ntask - chunk0);
    fill_table(chunk0, 0, nchunk);
    if (tid == 0) ctl[0] = 0;
    synced = 1;
    ld_done = 0;
    __syncthreads();
  }
  S2_STAMP(4);
  if (a.route_ctr) {  // (probe launches only: ctl[2], ctl[3] are free)
    if (lane == 0) {
      atomicAdd(reinterpret_cast<unsigned*>(&ctl[2]), n_items >> 16);
      atomicAdd(reinterpret_cast<unsigned*>(&ctl[3]), n_items & 0xffffu);
    }
  }
  __syncthreads();  // every window is merged: the tile is complete
  if (a.route_ctr && tid == 0) {
    atomicAdd(a.route_ctr, (unsigned)ctl[2]);
    atomicAdd(a.route_ctr + 1, (unsigned)ctl[3]);
  }
  S2_STAMP(5);
  // ---- epilogue: (tile + background) normalised, each output written once --
  // (a.ep: 0 scalar stores; 1 / 2 / 3 four cells per lane, plain / write-through
  // / non-temporal; 4 / 5 whole lines per store instruction, plain / non-temporal)
  const int ep_st = (a.ep == 3 || a.ep == 5) ? 3 : (a.ep == 2 ? 2 : 1);
  if (BOTH && a.out_disp) {
    // the disparity pass: every layer's splatted disparity normalised by its own
    // canvas weight, then the maximum over the layers (ldi.py:157-158, 170)
    const float bg = a.bg;
    const size_t P = (size_t)Ht * Wt;
    const size_t o0 = (size_t)b * P + (size_t)row0 * Wt;
    const int ncell = rows * Wt;
    auto dmax_of = [&](int i) {
      float dmax = 0.0f;
      for (int l = 0; l < a.L; ++l) {
        const float4 A = tile4[(size_t)l * R * Wt + i];
        const float dl = div_rn(A.x, safe_den(A.w + bg));
        dmax = l == 0 ? dl : fmaxf(dmax, dl);
      }
      return dmax;
    };
    if (a.ep) {
      for (int i = tid; i < (ncell >> 2); i += T)
        s2_store4(a.out_disp + o0 + 4 * (size_t)i, dmax_of(4 * i), dmax_of(4 * i + 1),
                  dmax_of(4 * i + 2), dmax_of(4 * i + 3), ep_st);
    } else {
      for (int i = tid; i < ncell; i += T) a.out_disp[o0 + i] = dmax_of(i);
    }
  } else if (BOTH && a.ep >= 4) {
    // per layer (ldi.py:157-163, 176-177) and composed (:167-174), whole lines
    // per store instruction as in the composed-only epilogue below: lane q of a
    // wave writes floats 4q .. 4q + 3 of every view's colour plane; they belong
    // to cells f / 3 and f / 3 + 1 of each layer's tile
    const float bg = a.bg;
    const size_t P = (size_t)Ht * Wt;
    const size_t o0 = (size_t)b * P + (size_t)row0 * Wt;
    const int ncell = rows * Wt;
    const int nq = (3 * ncell) >> 2;
    const size_t lstride = (size_t)R * Wt;
    for (int q = tid; q < nq; q += T) {
      const unsigned f = 4u * (unsigned)q;
      const unsigned c0 = __umulhi(f, 0xAAAAAAABu) >> 1;  // f / 3
      const unsigned o3 = f - 3u * c0;
      const int c1 = min((int)c0 + 1, ncell - 1);
      float4 CA = make_float4(0.f, 0.f, 0.f, 0.f), CB = CA;
      for (int l = 0; l < a.L; ++l) {
        const float4 A = tile4[l * lstride + c0];
        const float4 Bc = tile4[l * lstride + c1];
        const float wa = A.w + bg, wb = Bc.w + bg;
        const float ra = __builtin_amdgcn_rcpf(safe_den(wa));
        const float rb_ = __builtin_amdgcn_rcpf(safe_den(wb));
        const float ax = (A.x + bg) * ra, ay = (A.y + bg) * ra, az = (A.z + bg) * ra;
        const float bx = (Bc.x + bg) * rb_, by = (Bc.y + bg) * rb_, bz = (Bc.z + bg) * rb_;
        CA.x += A.x + bg; CA.y += A.y + bg; CA.z += A.z + bg; CA.w += wa;
        CB.x += Bc.x + bg; CB.y += Bc.y + bg; CB.z += Bc.z + bg; CB.w += wb;
        s2_store4(a.out_img + 3 * ((size_t)l * a.B * P + o0) + f,
                  o3 == 0 ? ax : (o3 == 1 ? ay : az), o3 == 0 ? ay : (o3 == 1 ? az : bx),
                  o3 == 0 ? az : (o3 == 1 ? bx : by), o3 == 0 ? bx : (o3 == 1 ? by : bz),
                  ep_st);
      }
      if (a.out_img_c) {
        const float ra = __builtin_amdgcn_rcpf(safe_den(CA.w));
        const float rb_ = __builtin_amdgcn_rcpf(safe_den(CB.w));
        const float ax = CA.x * ra, ay = CA.y * ra, az = CA.z * ra;
        const float bx = CB.x * rb_, by = CB.y * rb_, bz = CB.z * rb_;
        s2_store4(a.out_img_c + 3 * o0 + f,
                  o3 == 0 ? ax : (o3 == 1 ? ay : az), o3 == 0 ? ay : (o3 == 1 ? az : bx),
                  o3 == 0 ? az : (o3 == 1 ? bx : by), o3 == 0 ? bx : (o3 == 1 ? by : bz),
                  ep_st);
      }
    }
    const float* const tw = reinterpret_cast<const float*>(tile4) + 3;
    for (int q = tid; q < (ncell >> 2); q += T) {
      float cw[4] = {0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l < a.L; ++l) {
        const float* t = tw + 4 * (l * lstride + 4 * q);
        const float w0 = t[0] + bg, w1 = t[4] + bg, w2 = t[8] + bg, w3 = t[12] + bg;
        cw[0] += w0; cw[1] += w1; cw[2] += w2; cw[3] += w3;
        s2_store4(a.out_wts + (size_t)l * a.B * P + o0 + 4 * q, w0, w1, w2, w3, ep_st);
      }
      if (a.out_img_c)
        s2_store4(a.out_wts_c + o0 + 4 * q, cw[0], cw[1], cw[2], cw[3], ep_st);
    }
  } else if (BOTH && a.ep) {
    // per layer (ldi.py:157-163, 176-177) and composed (:167-174), four cells
    // per thread: 16-byte stores
    const float bg = a.bg;
    const size_t P = (size_t)Ht * Wt;
    const size_t o0 = (size_t)b * P + (size_t)row0 * Wt;
    const int nquad = (rows * Wt) >> 2;
    // (plain stores: a lane's 16 bytes are a third of the 48 it owns -- partial
    // lines per store instruction, which non-temporal stores do not merge:
    // measured 215 vs 174 us at config 3)
    const int ep = a.ep == 2 ? 2 : 1;
    for (int i = tid; i < nquad; i += T) {
      float C[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) C[k] = 0.0f;
      for (int l = 0; l < a.L; ++l) {
        float c[12], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 A = tile4[(size_t)l * R * Wt + 4 * i + k];
          const float Wsum = A.w + bg;
          const float rw = __builtin_amdgcn_rcpf(safe_den(Wsum));
          c[3 * k + 0] = (A.x + bg) * rw;
          c[3 * k + 1] = (A.y + bg) * rw;
          c[3 * k + 2] = (A.z + bg) * rw;
          w[k] = Wsum;
          C[4 * k + 0] += A.x + bg; C[4 * k + 1] += A.y + bg;
          C[4 * k + 2] += A.z + bg; C[4 * k + 3] += Wsum;
        }
        const size_t o = (size_t)l * a.B * P + o0 + 4 * (size_t)i;
        float* const pi = a.out_img + 3 * o;
        s2_store4(pi, c[0], c[1], c[2], c[3], ep);
        s2_store4(pi + 4, c[4], c[5], c[6], c[7], ep);
        s2_store4(pi + 8, c[8], c[9], c[10], c[11], ep);
        s2_store4(a.out_wts + o, w[0], w[1], w[2], w[3], ep);
      }
      if (a.out_img_c) {
        float c[12], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float rw = __builtin_amdgcn_rcpf(safe_den(C[4 * k + 3]));
          c[3 * k + 0] = C[4 * k + 0] * rw;
          c[3 * k + 1] = C[4 * k + 1] * rw;
          c[3 * k + 2] = C[4 * k + 2] * rw;
          w[k] = C[4 * k + 3];
        }
        float* const pi = a.out_img_c + 3 * (o0 + 4 * (size_t)i);
        s2_store4(pi, c[0], c[1], c[2], c[3], ep);
        s2_store4(pi + 4, c[4], c[5], c[6], c[7], ep);
        s2_store4(pi + 8, c[8], c[9], c[10], c[11], ep);
        s2_store4(a.out_wts_c + o0 + 4 * (size_t)i, w[0], w[1], w[2], w[3], ep);
      }
    }
  } else if (BOTH) {
    // per layer (ldi.py:157-163, 176-177) and composed (:167-174): the sum of
    // the layers' canvases, each with its own background
    const float bg = a.bg;
    const size_t P = (size_t)Ht * Wt;
    const size_t o0 = (size_t)b * P + (size_t)row0 * Wt;
    const int ncell = rows * Wt;
    for (int i = tid; i < ncell; i += T) {
      float4 C = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int l = 0; l < a.L; ++l) {
        const float4 A = tile4[(size_t)l * R * Wt + i];
        const float Wsum = A.w + bg;
        const float rw = __builtin_amdgcn_rcpf(safe_den(Wsum));
        const size_t o = (size_t)l * a.B * P + o0 + i;
        a.out_img[3 * o + 0] = (A.x + bg) * rw;
        a.out_img[3 * o + 1] = (A.y + bg) * rw;
        a.out_img[3 * o + 2] = (A.z + bg) * rw;
        a.out_wts[o] = Wsum;
        C.x += A.x + bg; C.y += A.y + bg; C.z += A.z + bg; C.w += Wsum;
      }
      if (a.out_img_c) {
        const float rw = __builtin_amdgcn_rcpf(safe_den(C.w));
        a.out_img_c[3 * (o0 + i) + 0] = C.x * rw;
        a.out_img_c[3 * (o0 + i) + 1] = C.y * rw;
        a.out_img_c[3 * (o0 + i) + 2] = C.z * rw;
        a.out_wts_c[o0 + i] = C.w;
      }
    }
  } else if (a.ep >= 4) {
    // Fully coalesced 16-byte stores: store instruction n of a wave writes ONE
    // contiguous kilobyte (whole 128-byte lines), lane l the floats 4 (64 n + l)
    // ... + 3 of the band's colour plane -- they belong to the cells f / 3 and
    // f / 3 + 1, read straight from the tile.  (With four cells per lane -- the
    // branch below -- a store instruction writes 16 bytes every 48: partial
    // lines, which non-temporal stores do not merge.)
    const float lbg = a.lbg;
    const size_t P = (size_t)Ht * Wt;
    float* const oi = a.out_img + ((size_t)b * P + (size_t)row0 * Wt) * 3;
    float* const ow = a.out_wts + (size_t)b * P + (size_t)row0 * Wt;
    const int ncell = rows * Wt;
    const int nq = (3 * ncell) >> 2;
    for (int q = tid; q < nq; q += T) {
      const unsigned f = 4u * (unsigned)q;
      const unsigned c0 = __umulhi(f, 0xAAAAAAABu) >> 1;  // f / 3
      const unsigned o = f - 3u * c0;
      const float4 A = tile4[c0];
      const float4 Bc = tile4[min((int)c0 + 1, ncell - 1)];
      const float ra = __builtin_amdgcn_rcpf(safe_den(A.w + lbg));
      const float rb_ = __builtin_amdgcn_rcpf(safe_den(Bc.w + lbg));
      const float ax = (A.x + lbg) * ra, ay = (A.y + lbg) * ra, az = (A.z + lbg) * ra;
      const float bx = (Bc.x + lbg) * rb_, by = (Bc.y + lbg) * rb_, bz = (Bc.z + lbg) * rb_;
      // o = 0: a.xyz b.x | o = 1: a.yz b.xy | o = 2: a.z b.xyz
      const float v0 = o == 0 ? ax : (o == 1 ? ay : az);
      const float v1 = o == 0 ? ay : (o == 1 ? az : bx);
      const float v2 = o == 0 ? az : (o == 1 ? bx : by);
      const float v3 = o == 0 ? bx : (o == 1 ? by : bz);
      s2_store4(oi + f, v0, v1, v2, v3, ep_st);
    }
    const float* const tw = reinterpret_cast<const float*>(tile4) + 3;
    for (int q = tid; q < (ncell >> 2); q += T) {
      const float* t = tw + 16 * q;
      s2_store4(ow + 4 * q, t[0] + lbg, t[4] + lbg, t[8] + lbg, t[12] + lbg, ep_st);
    }
  } else if (a.ep) {
    // four cells per thread: three 16-byte stores of colours and one of weights
    // (Wt % 4 == 0 and 16-byte aligned outputs: checked by the launcher)
    const float lbg = a.lbg;
    const size_t P = (size_t)Ht * Wt;
    float* const oi = a.out_img + ((size_t)b * P + (size_t)row0 * Wt) * 3;
    float* const ow = a.out_wts + (size_t)b * P + (size_t)row0 * Wt;
    const int nquad = (rows * Wt) >> 2;
    const int ep = a.ep;
    for (int i = tid; i < nquad; i += T) {
      float c[12], w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 A = tile4[4 * i + k];
        const float Wsum = A.w + lbg;
        const float rw = __builtin_amdgcn_rcpf(safe_den(Wsum));
        c[3 * k + 0] = (A.x + lbg) * rw;
        c[3 * k + 1] = (A.y + lbg) * rw;
        c[3 * k + 2] = (A.z + lbg) * rw;
        w[k] = Wsum;
      }
      float* const pi = oi + 12 * (size_t)i;
      s2_store4(pi, c[0], c[1], c[2], c[3], ep);
      s2_store4(pi + 4, c[4], c[5], c[6], c[7], ep);
      s2_store4(pi + 8, c[8], c[9], c[10], c[11], ep);
      s2_store4(ow + 4 * (size_t)i, w[0], w[1], w[2], w[3], ep);
    }
  } else {
    const float lbg = a.lbg;
    const size_t P = (size_t)Ht * Wt;
    float* const oi = a.out_img + ((size_t)b * P + (size_t)row0 * Wt) * 3;
    float* const ow = a.out_wts + (size_t)b * P + (size_t)row0 * Wt;
    const int ncell = rows * Wt;
    for (int i = tid; i < ncell; i += T) {
      const float4 A = tile4[i];
      const float Wsum = A.w + lbg;
      const float rw = __builtin_amdgcn_rcpf(safe_den(Wsum));
      oi[3 * i + 0] = (A.x + lbg) * rw;
      oi[3 * i + 1] = (A.y + lbg) * rw;
      oi[3 * i + 2] = (A.z + lbg) * rw;
      ow[i] = Wsum;
    }
  }
  S2_STAMP(6);
#ifdef S2X_STAMPS
  if (stamp && lane == 0)  // slot 7: the wave's items by route, 16 bits each
    stamp[7] = (long long)((unsigned long long)min(route_n[0], 65535u) |
                           ((unsigned long long)min(route_n[1], 65535u) << 16) |
                           ((unsigned long long)min(route_n[2], 65535u) << 32) |
                           ((unsigned long long)min(route_n[3], 65535u) << 48));
#endif
}

size_t s2_lds_bytes(int R, int Wt, int nw, int wmax, int cap, int qcap, int cell,
                    int nt = 1) {
  const int whs = ((wmax / 2 + 15) & ~15) + 8;
  return (size_t)nw * 2 * whs * 16 + (size_t)nt * R * Wt * 16 +
         (size_t)cap * (sizeof(Task) + sizeof(TaskX)) +
         (size_t)((4 + nt * R + 3) & ~3) * 4 + (size_t)nw * qcap * 20 +
         (size_t)((nw * wmax + 15) & ~15) + (cell ? (size_t)nt * R * Wt * 4 : 0) + 16;
}

struct S2Plan { int R, nw, cell, cap, qcap, ilv, hf, nsplit, lsub, nunit; size_t lds; double est; };

// Band height, waves per workgroup, how many row-segment units are handed out
// layer by layer.  Grounded in measurements (profiles/r03/): a launch streams
// at ~5.7 TB/s once every CU has a workgroup with >= 48 KB of loads in flight;
// what a launch loses is its start (first loads ~3 us after the workgroup
// starts), its tail (waves of a workgroup and workgroups of a launch finish
// apart) and whole rounds of workgroups: so one workgroup per CU (the LDS tile
// allows only one) in ONE round, as many waves as fit, the tallest band that
// still gives every CU a workgroup.
int s2_plan_search(const LsiSplatDesc* d, int wmax, int maxnw, bool both, S2Plan* out);

// The plan depends on the call's geometry only: remembered per geometry, so
// that an eager caller (one launch per Python call) does not pay the search
// again (~80 candidate plans) on every launch.
int s2_plan(const LsiSplatDesc* d, int wmax, int maxnw, bool both, S2Plan* out) {
  struct Key { int v[14]; };
  struct Entry { Key k; S2Plan p; };
  static std::mutex mu;
  static std::vector<Entry> memo;
  Key k;
  const int kv[14] = {d->L, d->B, d->H, d->W, d->Ht, d->Wt, wmax, maxnw, both ? 1 : 0,
                      d->tune_rows, d->tune_threads, d->reserved, 0,
                      d->tune_window & LSI_STREAM_FLAG_BITS};
  memcpy(k.v, kv, sizeof(kv));
  memcpy(&k.v[12], &d->trg_downsampling, sizeof(float));
  {
    std::lock_guard<std::mutex> g(mu);
    for (const Entry& e : memo)
      if (memcmp(e.k.v, k.v, sizeof(k.v)) == 0) { *out = e.p; return LSI_OK; }
  }
  const int rc = s2_plan_search(d, wmax, maxnw, both, out);
  if (rc == LSI_OK) {
    std::lock_guard<std::mutex> g(mu);
    if (memo.size() >= 64) memo.erase(memo.begin());
    memo.push_back(Entry{k, *out});
  }
  return rc;
}

int s2_plan_search(const LsiSplatDesc* d, int wmax, int maxnw, bool both, S2Plan* out) {
  const int nt = both ? d->L : 1;
  const int nseg = (d->W + SEG - 1) / SEG;
  static const char* cap_env = getenv("LSI_STREAM_LDS_CAP");
  const size_t lds_cap = cap_env ? (size_t)atol(cap_env) : 160 * 1024;
  static const char* ns_env = getenv("LSI_S2_NSPLIT");  // experiments
  static const char* ls_env = getenv("LSI_S2_LSUB");
  const int force_cell = (d->reserved >> 18) & 3;
  static const char* hf_env = getenv("LSI_S2_HALOFIRST");
  const int hf = hf_env ? atoi(hf_env) : 1;
  const double NCU = 256.0, CHIP_GBPS = 5700.0, CU_GBPS = 32.0;
  S2Plan best; best.est = -1.0; best.nw = 0;
  // Band heights: powers of two; tune_rows asks for any one height.  (Bands
  // need not divide the image: the last one is shorter.)
  for (int Rp = 1; Rp <= 64; Rp *= 2) {
    const int R = d->tune_rows > 0 ? d->tune_rows : Rp;
    if (d->tune_rows > 0 && Rp > 1) break;
    if (d->tune_rows <= 0 && R > 1 && R / 2 >= d->Ht) break;
    const long nwg = (long)((d->Ht + R - 1) / R) * d->B;
    const int srows = (int)ceilf((float)(R + 1) / d->trg_downsampling);
    const int ilv = srows >= 15 ? 1 : 0;
    // (the table holds either order: halo rows first or not is decided per plan)
    const int nrow = max(s2_rows_padded(srows, ilv, hf), s2_rows_padded(srows, ilv, 0));
    const int nunit = nrow * nseg;
    // (both outputs: every item merges, and a merge under cell locks costs more
    // LDS operations than one under two row locks: 204 vs 184 us at config 3)
    int cell = (R <= 8 && nunit <= 40 && !both) ? 1 : 0;
    if (force_cell == 1) cell = 0;
    if (force_cell == 2 && !both) cell = 1;  // (both outputs: row locks only)
    for (int c = maxnw; c >= 2; --c) {
      if (d->tune_threads > 0 && c != (d->tune_threads + 63) / 64) continue;
      // A unit is all L layers of a row segment on one wave (one merge per
      // unit).  Units are handed out whole up to 6 layers; deeper LDIs in
      // parts of at most 4 layers, so that a wave's last unit stays short.
      // (Measured at L = 4: halves or single layers cost more in merges and
      // task switches than they win in balance -- cfg3 86 -> 87..154 us, a
      // 4-view shard 22 -> 24..31 us.)  reserved bits 12-15 (tests,
      // experiments): layers per ticket for every unit.
      int lsub = d->L <= 6 ? d->L : (d->L + (d->L + 3) / 4 - 1) / ((d->L + 3) / 4);
      const int sub_override = (d->reserved >> 12) & 0xf;
      if (sub_override) lsub = sub_override < d->L ? sub_override : d->L;
      if (ls_env) lsub = atoi(ls_env);
      if (lsub < 1) lsub = 1;
      if (both) lsub = d->L;  // (a unit's items merge one by one anyway)
      int nsplit = lsub < d->L ? nunit : 0;
      if (ns_env) nsplit = atoi(ns_env);
      if (nsplit > nunit) nsplit = nunit;
      const int tickets = nunit - nsplit + nsplit * ((d->L + lsub - 1) / lsub);
      if (d->tune_threads <= 0 && c > tickets && c > 2) continue;
      const int cap = (tickets + 15) / 16 * 16;
      int q = 64;
      while (q >= 16 && s2_lds_bytes(R, d->Wt, c, wmax, cap, q, cell, nt) > lds_cap) q /= 2;
      if (q < 16) continue;
      const size_t lds = s2_lds_bytes(R, d->Wt, c, wmax, cap, q, cell, nt);
      long k = (long)(160 * 1024 / lds);          // co-resident workgroups per CU
      if (k > maxnw / c) k = maxnw / c;
      if (k < 1) k = 1;
      const double slots = NCU * (double)k;
      const double rounds = ceil((double)nwg / slots);
      const double conc = fmin((double)nwg, slots);   // workgroups running together
      // microseconds: streaming share of the chip (a CU alone cannot pull more
      // than ~CU_GBPS), or the waves' own latency chains (2 items in flight)
      const double bytes = (double)srows * nseg * d->L * 4096.0;
      const double gbps = fmin(CU_GBPS / (double)k, CHIP_GBPS / conc);
      const double t_stream = bytes / (gbps * 1e3);
      const double t_lat = ceil((double)nunit * d->L / c) * 1.1;
      const double t_wg = 3.0 + fmax(t_stream, t_lat) + 0.5 +
                          (double)R * d->Wt / (c * 64.0) * 0.02;
      const double est = rounds * t_wg + (nwg < (long)NCU ? 0.0 : 0.0);
      if (best.nw == 0 || est < best.est - 1e-9 ||
          (est <= best.est * 1.0001 && R > best.R)) {
        best.est = est; best.R = R; best.nw = c; best.cell = cell;
        best.cap = cap; best.qcap = q; best.lds = lds; best.ilv = ilv;
        // halo rows first only when the 4 * nseg halo units are the waves' own
        // first units (taken by wave index, all workgroups in step): with more
        // of them (config 5: 24 for 12 waves) the second round is no longer
        // aligned between neighbours and every one of them merges into the
        // band's first or last tile row (config 5: 92.4 us without, 94.3 with)
        best.hf = (hf && 4 * nseg <= c) ? hf : 0;
        best.nsplit = nsplit; best.lsub = lsub; best.nunit = nunit;
      }
    }
  }
  if (best.nw == 0) return LSI_EINVAL;
  *out = best;
  static const bool verbose = getenv("LSI_STREAM_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "lsi stream2 plan: R=%d waves=%d %s-locks table=%d queue=%d "
            "interleave=%d split=%dx%d est=%.1f us lds=%zu\n", best.R, best.nw,
            best.cell ? "cell" : "row", best.cap, best.qcap, best.ilv,
            best.nsplit, best.lsub, best.est, best.lds);
  return LSI_OK;
}

#ifndef LSI_S2_NSETS
#define LSI_S2_NSETS 2
#endif
#ifndef LSI_S2_MAXT
#define LSI_S2_MAXT 768
#endif

// 16 waves x one register set (<= 128 VGPRs) instead of 12 x two
// (lsi_stream2_launch).  Measured, same box, 12 x 2 -> 16 x 1
// (profiles/r04/ab_wide.txt):
//  * bands with fewer than two units per wave are all start and tail -- a
//    wave's second item in flight buys nothing when it has one or two units in
//    all, four more waves take a quarter of the units off the others: 4-view
//    shard of config 3 (18 units per band) 22.6 -> 20.9 us, config 2 18.4 -> 16.6;
//  * smooth disparity fields at larger launches are bandwidth-bound and tie or
//    lose a little (16 items in flight per CU instead of 24): config 3 81.2 ->
//    81.5, 81.4 -> 81.4, 79.9 -> 80.7 us on three boxes, 16-view shard 48.1 ->
//    47.6, 8-view shard 30.6 -> 31.1;
//  * folded / i.i.d. fields (routes B' / C: the item loop is bound by vector and
//    LDS work, which more waves hide) gain 12 - 15 %: config 3 140.7 -> 119.3 /
//    152.5 -> 130.7 us, its shards 80.3 -> 70.5, 52.6 -> 46.7, 38.0 -> 34.7;
//  * where 16 waves do not fit next to the tile (config 5: the plan stays at
//    12) the one-register-set build loses: 88.4 -> 93.6 us.
// The planner cannot see the field: by itself it takes 16 x 1 for the small
// bands only; tune_threads > 768 asks for it anywhere (bench.py --threads 1024,
// forward_splat(threads=1024): the choice for rough fields), tune_threads <= 768
// for 12 x 2; LSI_S2_WIDE=0/1 forces one for experiments.
bool s2_choose(const LsiSplatDesc* d, int wmax, bool both, S2Plan* plan, bool* wide) {
  static const char* env = getenv("LSI_S2_WIDE");
  S2Plan narrow, wideplan;
  const int rc_n = d->tune_threads > LSI_S2_MAXT
                       ? LSI_EINVAL : s2_plan(d, wmax, LSI_S2_MAXT / 64, both, &narrow);
  // (tune_threads > 768 names the build, not a wave count: as many of the 16
  // waves as fit next to the tile)
  LsiSplatDesc dw = *d;
  if (dw.tune_threads > LSI_S2_MAXT) dw.tune_threads = 0;
  const int rc_w = (d->tune_threads > 0 && d->tune_threads <= LSI_S2_MAXT)
                       ? LSI_EINVAL : s2_plan(&dw, wmax, 1024 / 64, both, &wideplan);
  bool w = rc_w == LSI_OK &&
           (rc_n != LSI_OK || (wideplan.nw > LSI_S2_MAXT / 64 &&
                               narrow.nunit < 2 * (LSI_S2_MAXT / 64)));
  if (env && d->tune_threads <= 0) w = atoi(env) != 0 ? rc_w == LSI_OK : rc_n != LSI_OK;
  if (!w && rc_n != LSI_OK) return false;
  *plan = w ? wideplan : narrow;
  *wide = w;
  return true;
}

// ---- which build for a field the planner cannot see -----------------------------
// Smooth disparity fields (a trained network's, the benchmark's) run best on 12
// waves x two register sets; folded / noisy fields (an untrained network's) 12 -
// 15 % faster on 16 x one.  The kernel knows which it is: it counts the items
// that took the folded routes B' / C.  The CALLER may own a small record per call
// geometry (LsiSplatDesc.adapt -> LsiStreamAdapt, include/lsi_hip.h; NULL: no
// adaptation, the planner's choice): on PROBE launches (the first calls, then two
// of every 64) the kernel adds its counts to the record's device counter, a
// 12-byte asynchronous copy brings them and the probe's sequence number to the
// record's pinned host memory, and the NEXT call that sees the number there (a
// plain read of host memory: no runtime call, legal also while a stream is being
// captured) takes the decision -- 16 x 1 when more than FOLD_SHARE of the items
// were folded.  The decision of a call never depends on that call's own data, no
// host synchronisation is added, and a launch that is being captured into a HIP
// graph takes the decision standing at that moment and probes nothing (the
// graph has it baked in).  The library itself keeps NO state: everything lives
// in the caller's record (round 5 kept a table here -- global mutable state,
// which SURVEY 8(b) rules out, and a race between two threads probing the same
// geometry); calls that share a record must not overlap, which is the owner's
// business (lsi/geometry/ldi.py: one record per device, stream and geometry,
// used under a lock).  tune_threads, LSI_S2_WIDE and LSI_S2_ADAPT=0 switch the
// mechanism off.
constexpr double S2_FOLD_SHARE = 0.5;
typedef LsiStreamAdapt S2Adapt;
// the pending probe's counts if they have arrived
bool s2_adapt_poll(S2Adapt* e) {
  if (!e->pending) return false;
  volatile uint32_t* h = e->ctr_host;
  if (h[2] != e->seq) return false;
  __sync_synchronize();
  const unsigned fold = h[0], all = h[1];
  if (all > 0u) e->state = ((double)fold > S2_FOLD_SHARE * (double)all) ? 2 : 1;
  e->pending = 0;
  return true;
}

// Returns the caller's record (NULL: mechanism off); *want_wide the standing
// decision, *probe whether this launch counts.
S2Adapt* s2_adapt_begin(const LsiSplatDesc* d, hipStream_t stream, bool* want_wide,
                        bool* probe) {
  *want_wide = false; *probe = false;
  S2Adapt* e = d->adapt;
  if (!e || !e->ctr_dev || !e->ctr_host) return nullptr;
  static const char* off = getenv("LSI_S2_ADAPT");
  static const char* forced = getenv("LSI_S2_WIDE");
  if ((off && off[0] == '0') || forced || d->tune_threads != 0 || d->tune_rows != 0) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  const bool capturing = cap != hipStreamCaptureStatusNone;
  s2_adapt_poll(e);
  *want_wide = e->state == 2;
  if (!capturing && !e->pending && (e->state == 0 || (e->calls & 63u) < 2u)) *probe = true;
  e->calls += 1u;
  return e;
}

}  // namespace

// The standing decision of a caller-owned record (after looking for a pending
// probe's counts): 0 none yet, 1 twelve waves x two register sets, 2 sixteen x
// one; -1 for NULL.
extern "C" int lsi_stream_adapt_state(LsiStreamAdapt* a) {
  if (!a || !a->ctr_host) return -1;
  s2_adapt_poll(a);
  return a->state;
}

// Whether the compact instance renders this call (else: splat_stream_kernel).
bool lsi_stream2_applies(const SplatArgs& a, bool simple, int layout) {
  const LsiSplatDesc* d = &a.d;
  static const char* off = getenv("LSI_STREAM2");
  if (off && off[0] == '0') return false;
  if (d->reserved & 0x40000000) return false;  // tests: force the general kernel
  if (((d->reserved >> 16) & 3) == 2) return false;  // exchange bands asked for
  if (!simple || (layout != 0 && layout != 2)) return false;
  const bool both = a.out_img_c != nullptr;  // lsi_splat_fwd_both: per-layer + composed
  const unsigned mode = d->flags & (LSI_COMPOSE | LSI_HAS_MASK | LSI_WANT_DISP | LSI_DETERMINISTIC);
  // per-layer outputs alone (compose_layers=False): the tile-per-layer instance
  // without its composed view, for small launches (a training-size LDI: 31 us
  // against the general kernel's 62; at config 3 the general kernel is ahead)
  const bool layers_only = !both && mode == 0u && (long)d->B * d->L <= 16;
  if (!layers_only && mode != (both ? 0u : (unsigned)LSI_COMPOSE)) return false;
  if ((both || layers_only) && d->L > 15) return false;
  if (both && !a.out_wts_c) return false;
  if (d->W % 4 != 0 || d->W < 4 || d->H > 65535 || (d->W + SEG - 1) / SEG > 32767) return false;
  if (layout == 0 && (d->tex_sx != 3 || d->tex_sc != 1 || d->disp_sx != 1)) return false;
  return true;
}

int lsi_stream2_launch(const SplatArgs& a, int wmax, hipStream_t stream,
                       bool disp_pass) {
  const LsiSplatDesc* d = &a.d;
  S2Plan plan;
  // (one tile per layer: both outputs, the disparity pass, per-layer outputs alone)
  const bool both = disp_pass || a.out_img_c != nullptr || !(d->flags & LSI_COMPOSE);
  // Two builds of the kernel: 12 waves with two register sets (two items in
  // flight per wave), or 16 waves with one (<= 128 VGPRs); see s2_choose().
  bool wide = false;
  if (!s2_choose(d, wmax, both, &plan, &wide)) return LSI_EINVAL;
  // the field decides between the two builds where the planner alone would
  // take 12 x 2 (s2_adapt_begin)
  S2Adapt* adapt = nullptr;
  bool probe = false;
#ifndef S2X_STAMPS
  if (!both && !wide) {
    bool want_wide = false;
    adapt = s2_adapt_begin(d, stream, &want_wide, &probe);
    if (adapt && want_wide) {
      S2Plan wp;
      if (s2_plan(d, wmax, 1024 / 64, both, &wp) == LSI_OK && wp.nw > LSI_S2_MAXT / 64) {
        plan = wp; wide = true;
      }
    }
  }
#endif
  S2Args k;
  k.tex = a.tex; k.disp = a.disp; k.M = a.M;
  k.out_img = a.out_img; k.out_wts = a.out_wts;
  k.out_img_c = a.out_img_c; k.out_wts_c = a.out_wts_c;
  k.out_disp = disp_pass ? a.out_disp : nullptr;
  k.B = d->B; k.H = d->H; k.Ht = d->Ht; k.Wt = d->Wt; k.L = d->L;
  k.nseg = (d->W + SEG - 1) / SEG;
  k.W = d->W;
  k.tex_sl = (int)d->tex_sl; k.tex_sb = (int)d->tex_sb; k.tex_sy = (int)d->tex_sy;
  k.disp_sl = (int)d->disp_sl; k.disp_sb = (int)d->disp_sb; k.disp_sy = (int)d->disp_sy;
  k.s = d->trg_downsampling; k.max_disp = d->max_disp;
  // exp((clip(d/max,0,1) - 0.5)*scale) = exp2(clip(d,0,max)*zA + zB)
  const double l2e = 1.4426950408889634;
  k.zA = (float)((double)d->zbuf_scale * l2e / (double)d->max_disp);
  k.zB = (float)(-0.5 * (double)d->zbuf_scale * l2e);
  k.lbg = (float)d->L * d->bg_wt;
  k.bg = d->bg_wt;
  k.R = plan.R; k.wmax = wmax; k.qcap = plan.qcap; k.cap = plan.cap;
  k.ilv = plan.ilv;
  k.hf = plan.hf;
  {
    static const char* ep_env = getenv("LSI_S2_EPILOGUE");
    // 5: whole 128-byte lines per store instruction, non-temporal.  Measured at
    // config 3 on one box (profiles/r04/ab_epilogue.txt): scalar stores 89.6 us,
    // 16-byte plain 90.5, 16-byte non-temporal (16 of every 48 bytes per
    // instruction) 90.4, whole lines plain 90.7, whole lines non-temporal 82.0:
    // the rendered views no longer linger dirty in the L2s until the launch ends.
    const int ep = ep_env ? atoi(ep_env) : 5;
    // (four cells per thread, 16-byte stores: rows of whole quads, aligned outputs)
    bool al = d->Wt % 4 == 0;
    for (const float* p : {k.out_img, k.out_wts, k.out_img_c, k.out_wts_c, k.out_disp})
      al = al && ((uintptr_t)p & 15) == 0;
    k.ep = al ? ep : 0;
  }
  k.nsplit = plan.nsplit;
  k.lsub = plan.lsub;
  const int nbands = (d->Ht + plan.R - 1) / plan.R;
  k.inv_gx = 1.0f / (float)nbands;
  k.inv_nseg = 1.0f / (float)k.nseg;
  k.stamps = nullptr;
  k.route_ctr = nullptr;
  if (adapt && probe) {
    adapt->seq += 1u;
    if (adapt->seq == 0u) adapt->seq = 1u;
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(adapt->ctr_dev), 0, 2, stream) == hipSuccess &&
        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(adapt->ctr_dev + 2), (int)adapt->seq, 1,
                          stream) == hipSuccess)
      k.route_ctr = adapt->ctr_dev;
    else
      (void)hipGetLastError();
  }
#ifdef S2X_STAMPS
  // (the last bytes of the workspace)
  if ((d->reserved & 4) && a.canvas &&
      a.ws_bytes >= (size_t)nbands * d->B * 16 * 8 * 8)
    k.stamps = reinterpret_cast<long long*>(
        reinterpret_cast<char*>(a.canvas) + a.ws_bytes - (size_t)nbands * d->B * 16 * 8 * 8);
#endif
  const bool pack = (d->flags & LSI_PACKED_RGBD) != 0;  // (verified by the caller)
  const bool ragged = d->W % SEG != 0;  // the last segment of a row is partial
  // <NSETS, cell locks, MAXT, both outputs, RGBD pixels, partial last segment>
  // (row locks with both outputs: s2_plan never picks cell locks there)
#define S2_FN(C, B, P, R)                                                             \
  (wide ? (const void*)splat_stream2_kernel<1, C, 1024, B, P, R>                         \
        : (const void*)splat_stream2_kernel<LSI_S2_NSETS, C, LSI_S2_MAXT, B, P, R>)
#define S2_PICK(C, B) (pack ? (ragged ? S2_FN(C, B, true, true) : S2_FN(C, B, true, false)) \
                            : (ragged ? S2_FN(C, B, false, true) : S2_FN(C, B, false, false)))
  const void* fn = both ? S2_PICK(false, true)
                        : (plan.cell ? S2_PICK(true, false) : S2_PICK(false, false));
#undef S2_PICK
#undef S2_FN
  if (lsi_ensure_dynamic_lds(fn, plan.lds) != LSI_OK) return LSI_ELAUNCH;
  void* kargs[1] = {&k};
  if (hipLaunchKernel(fn, dim3(nbands, d->B), dim3(plan.nw * 64), kargs, plan.lds,
                      stream) != hipSuccess)
    return LSI_ELAUNCH;
  if (k.route_ctr) {   // the counts travel to the host behind the launch; read by a later call
    if (hipMemcpyAsync(const_cast<uint32_t*>(adapt->ctr_host), adapt->ctr_dev, 12,
                       hipMemcpyDeviceToHost, stream) == hipSuccess)
      adapt->pending = 1;
    else
      (void)hipGetLastError();
  }
  return hipGetLastError() == hipSuccess ? LSI_OK : LSI_ELAUNCH;
}
