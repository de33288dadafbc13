// afx_kernels_grid.hip — occupancy-grid acceleration of the reference's training loop as gfx950 kernels
// (nerf/run_nerf_acc.py:196-198,284-287; nerf/nerf_helpers_acc.py:10-31,65-78), a device ray sampler
// (nerf/nerf_helpers.py:137-150) and the counter-based generator both use in "perf mode".
//
// The reference delegates the grid and the march to nerfacc 0.3.x (CUDA, neither vendored nor pinned: absent here), so
// what is implemented is that package's PUBLISHED algorithm, restated in oracle/angio_oracle.py for the tests — parity
// unpinned at that third-party boundary.  All of this is HBM/latency-bound integer and fp32 work: one thread per ray or
// per cell, coalesced reads, no MFMA.  The MLP evaluations between the kernels (occupancy of jittered cell points,
// alpha of the candidate samples) are afx_mlp_infer launches on the points these kernels emit.
//
// Float arithmetic that decides an INDEX (cell of a point, number of steps of a ray) is written with explicit
// round-to-nearest single operations in the order of the torch expressions it restates, so that indices are bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afx {

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011): counter-based, so "perf mode" randomness needs no state buffer and no ordering:
// value i of stream (seed, offset) is a pure function.  u in [0,1): top 24 bits.
// ---------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform [0,1) number `i` of stream (seed, stream id)
__host__ __device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t stream, uint64_t i) {
  uint32_t o[4];
  philox4x32_10((uint32_t)(i >> 2), (uint32_t)(i >> 34), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return (float)(o[i & 3] >> 8) * (1.0f / 16777216.0f);
}

struct GridDesc {
  float lo[3], hi[3];      // roi_aabb
  int32_t res[3];
};

// cell of a world point: u = (p - lo) / (hi - lo); inside <=> 0 <= u < 1 on every axis; ijk = min(floor(u * res), res - 1)
__device__ __forceinline__ bool grid_cell(const GridDesc& g, float px, float py, float pz, int64_t& idx) {
  const float p[3] = {px, py, pz};
  int ijk[3];
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float u = __fdiv_rn(__fsub_rn(p[a], g.lo[a]), __fsub_rn(g.hi[a], g.lo[a]));
    inside = inside && (u >= 0.f) && (u < 1.f);
    int c = (int)floorf(__fmul_rn(u, (float)g.res[a]));
    c = c < 0 ? 0 : c;
    ijk[a] = c > g.res[a] - 1 ? g.res[a] - 1 : c;
  }
  idx = ((int64_t)ijk[0] * g.res[1] + ijk[1]) * g.res[2] + ijk[2];
  return inside;
}

// --- OccupancyGrid._update, step 1: one jittered world point per selected cell -----------------------------------
// cell_idx == nullptr: all cells (cell i).  jitter [n,3] in [0,1) (parity mode) or nullptr: Philox(seed, stream 3*?).
__global__ void k_grid_points(const int32_t* cell_idx, int64_t n, const float* jitter, uint64_t seed, uint64_t stream, GridDesc g, float* pts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cell_idx ? cell_idx[i] : i;
  const int k = (int)(c % g.res[2]);
  const int j = (int)((c / g.res[2]) % g.res[1]);
  const int ii = (int)(c / ((int64_t)g.res[2] * g.res[1]));
  const int ijk[3] = {ii, j, k};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float u = jitter ? jitter[3 * i + a] : philox_uniform(seed, stream, 3 * (uint64_t)i + a);
    // x = (coords + u) / resolution; x * (hi - lo) + lo
    const float x = __fdiv_rn(__fadd_rn((float)ijk[a], u), (float)g.res[a]);
    pts[3 * i + a] = __fadd_rn(__fmul_rn(x, __fsub_rn(g.hi[a], g.lo[a])), g.lo[a]);
  }
}

// --- step 2: occs[c] = max(occs[c] * decay, occ).  Two passes so that a cell drawn twice gets max(old * decay, occ_a, occ_b)
// whatever the order: (i) every selected cell drops to its decayed value, formed from a snapshot of the values BEFORE the
// update (idempotent under duplicates); (ii) an integer atomicMax on the bit patterns of the non-negative floats folds
// the new occupancies in - order-independent, so the result is deterministic.
__global__ void k_grid_decay(float* occs, const float* occs_before, const int32_t* cell_idx, int64_t n, float decay) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cell_idx ? cell_idx[i] : i;
  occs[c] = fmaxf(__fmul_rn(occs_before[c], decay), 0.f);
}
__global__ void k_grid_ema(float* occs, const int32_t* cell_idx, const float* occ_new, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cell_idx ? cell_idx[i] : i;
  const float v = occ_new[i];
  if (v > 0.f) atomicMax((unsigned int*)(occs + c), __float_as_uint(v));
}

// --- step 3: binary = occs > min(mean(occs), occ_thre); also the packed bitfield the march reads ------------------------
// fixed-order two-level sum (deterministic): partial[b] = sum of block b's slice in a tree of fixed shape
__global__ void __launch_bounds__(256) k_grid_sum(const float* occs, int64_t n, double* partial) {
  __shared__ double red[256];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  double s = 0.0;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) s += (double)occs[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void k_grid_binarize(const float* occs, int64_t n, const double* partial, int n_partial, float occ_thre, uint8_t* binary, uint32_t* bits) {
  double tot = 0.0;
  for (int i = 0; i < n_partial; ++i) tot += partial[i];
  const float mean = (float)(tot / (double)n);
  const float thr = mean < occ_thre ? mean : occ_thre;
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // one 32-cell word per thread
  if (w * 32 >= n) return;
  uint32_t word = 0;
  for (int b = 0; b < 32; ++b) {
    const int64_t c = w * 32 + b;
    if (c >= n) break;
    const bool on = occs[c] > thr;
    binary[c] = on ? 1 : 0;
    word |= (on ? 1u : 0u) << b;
  }
  bits[w] = word;
}

// bitfield of a byte mask that did not come out of k_grid_binarize (a restored grid): bits[w] bit b = binary[32 w + b] != 0
__global__ void k_grid_pack(const uint8_t* binary, int64_t n, uint32_t* bits) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 32 >= n) return;
  uint32_t word = 0;
  for (int b = 0; b < 32; ++b) {
    const int64_t c = w * 32 + b;
    if (c >= n) break;
    word |= (binary[c] ? 1u : 0u) << b;
  }
  bits[w] = word;
}

// --- ray marching (nerfacc.ray_marching, fixed-step lattice, AABB contraction) ----------------------------------------
struct MarchArgs {
  const float* org; const float* dir;      // [R,3]
  int64_t n_rays;
  int32_t has_aabb; float aabb[6];        // scene_aabb
  int32_t has_near, has_far; float near_plane, far_plane;
  float dt;
  const uint32_t* bits;                    // grid bitfield or nullptr
  GridDesc g;
};
__device__ __forceinline__ void march_range(const MarchArgs& a, int64_t r, float o[3], float d[3], float& tmin, int& n_steps) {
#pragma unroll
  for (int q = 0; q < 3; ++q) { o[q] = a.org[3 * r + q]; d[q] = a.dir[3 * r + q]; }
  float t_min = 0.f, t_max = 1e10f;
  if (a.has_aabb) {
    // slab test; rays that miss get t_min = t_max = 1e10
    float lo = -INFINITY, hi = INFINITY;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float inv = __fdiv_rn(1.0f, d[q] == 0.f ? 1e-12f : d[q]);
      const float t0 = __fmul_rn(__fsub_rn(a.aabb[q], o[q]), inv), t1 = __fmul_rn(__fsub_rn(a.aabb[3 + q], o[q]), inv);
      lo = fmaxf(lo, fminf(t0, t1));
      hi = fminf(hi, fmaxf(t0, t1));
    }
    const bool miss = hi < fmaxf(lo, 0.f);
    t_min = miss ? 1e10f : fmaxf(lo, 0.f);
    t_max = miss ? 1e10f : hi;
  }
  if (a.has_near) t_min = fmaxf(t_min, a.near_plane);
  if (a.has_far) t_max = fminf(t_max, a.far_plane);
  float ns = ceilf(__fdiv_rn(__fsub_rn(t_max, t_min), a.dt));
  if (!(ns > 0.f)) ns = 0.f;
  n_steps = t_min >= 1e10f ? 0 : (ns > 2.0e9f ? 2000000000 : (int)ns);
  // nerfacc marches `while (t_mid < far)`: a step belongs to the ray when its MID-POINT lies before t_max (not its start), so the
  // ceil() above is corrected by the step(s) at the end whose mid-point falls outside / inside - same fp32 expressions as the loops below
  auto mid_of = [&](int k) { const float ts = __fadd_rn(t_min, __fmul_rn((float)k, a.dt)); return __fmul_rn(__fadd_rn(ts, __fadd_rn(ts, a.dt)), 0.5f); };
  if (n_steps < 2000000000) {
    while (n_steps > 0 && !(mid_of(n_steps - 1) < t_max)) --n_steps;
    for (int i = 0; i < 2 && t_min < 1e10f && mid_of(n_steps) < t_max; ++i) ++n_steps;
  }
  tmin = t_min;
}
__device__ __forceinline__ bool march_keep(const MarchArgs& a, const float o[3], const float d[3], float ts, float te) {
  if (!a.bits) return true;
  const float m = __fmul_rn(__fadd_rn(ts, te), 0.5f);
  int64_t idx;
  if (!grid_cell(a.g, __fadd_rn(o[0], __fmul_rn(d[0], m)), __fadd_rn(o[1], __fmul_rn(d[1], m)), __fadd_rn(o[2], __fmul_rn(d[2], m)), idx)) return false;
  return (a.bits[idx >> 5] >> (idx & 31)) & 1u;
}
// The five per-ray kernels below run ONE WAVEFRONT PER RAY (blocks of 256 threads = 4 rays): the 64 lanes take 64 consecutive steps /
// samples of the ray at a time, compaction offsets come from a ballot + prefix population count, so the writes of a chunk are
// contiguous and the arithmetic per step is exactly the one-thread-per-ray version's (same expressions, same order where order matters).
// (One thread per ray left 5 625 rays x 300 dependent steps on 88 wavefronts: 0.66 ms of a 2.1 ms training iteration.)
__device__ __forceinline__ int lane_prefix(uint64_t ballot) {      // kept lanes below this one
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}
// pass 1: kept steps per ray.  pass 2 (offsets = exclusive scan of the counts): packed (ray_indices, t_starts, t_ends[, mid-points]).
__global__ void __launch_bounds__(256) k_march_count(const MarchArgs a, int32_t* counts) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= a.n_rays) return;
  float o[3], d[3], tmin;
  int ns;
  march_range(a, r, o, d, tmin, ns);
  int c = 0;
  for (int k0 = 0; k0 < ns; k0 += 64) {
    const int k = k0 + lane;
    bool keep = false;
    if (k < ns) {
      const float ts = __fadd_rn(tmin, __fmul_rn((float)k, a.dt)), te = __fadd_rn(ts, a.dt);
      keep = march_keep(a, o, d, ts, te);
    }
    c += __popcll(__ballot(keep));
  }
  if (lane == 0) counts[r] = c;
}
__global__ void __launch_bounds__(256) k_march_write(const MarchArgs a, const int64_t* offsets, int32_t* ray_indices, float* t_starts, float* t_ends, float* mid_pts) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= a.n_rays) return;
  float o[3], d[3], tmin;
  int ns;
  march_range(a, r, o, d, tmin, ns);
  int64_t base = offsets[r];
  for (int k0 = 0; k0 < ns; k0 += 64) {
    const int k = k0 + lane;
    bool keep = false;
    float ts = 0.f, te = 0.f;
    if (k < ns) {
      ts = __fadd_rn(tmin, __fmul_rn((float)k, a.dt));
      te = __fadd_rn(ts, a.dt);
      keep = march_keep(a, o, d, ts, te);
    }
    const uint64_t bal = __ballot(keep);
    if (keep) {
      const int64_t w = base + lane_prefix(bal);
      ray_indices[w] = (int32_t)r; t_starts[w] = ts; t_ends[w] = te;
      if (mid_pts) {
        // positions = o + d * (t_s + t_e) / 2.0   (alpha_fn, nerf_helpers_acc.py:13-15)
        const float sm = __fadd_rn(ts, te);
#pragma unroll
        for (int q = 0; q < 3; ++q) mid_pts[3 * w + q] = __fadd_rn(o[q], __fdiv_rn(__fmul_rn(d[q], sm), 2.0f));
      }
    }
    base += __popcll(bal);
  }
}
// alpha of the candidates from the raw MLP output: 1 - exp(-sigmoid(raw) * (t_e - t_s))  (alpha_fn, nerf_helpers_acc.py:19-23),
// then nerfacc's render_visibility per ray over its packed segment: steps with alpha < alpha_thre are skipped WITHOUT
// attenuating T; the ray stops once T < early_stop_eps.  keep[i] in {0,1}; counts[r] = kept steps of ray r.
// The 64 lanes form the alphas of 64 samples at once (the transcendental part); the transmittance product then runs over them IN ORDER
// (wave-uniform loop of lane broadcasts: the same multiplications in the same order as a sequential march, so the kept set is the same bit for bit).
__global__ void __launch_bounds__(256) k_march_visibility(const float* raw, int is_alpha, const float* t_starts, const float* t_ends, const int64_t* offsets, int64_t n_rays,
                                                          float early_stop_eps, float alpha_thre, uint8_t* keep, int32_t* counts) {
  __shared__ float sf[4][64];
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  const int64_t i0 = offsets[r], i1 = offsets[r + 1];
  float T = 1.f;
  int c = 0;
  for (int64_t b = i0; b < i1; b += 64) {
    const int64_t i = b + lane;
    float alpha = 0.f;
    if (i < i1) {
      alpha = raw[i];           // is_alpha: the caller's alpha_fn output
      if (!is_alpha) {
        const float sg = 1.f / (1.f + expf(-alpha));
        alpha = 1.f - expf(-__fmul_rn(sg, __fsub_rn(t_ends[i], t_starts[i])));
      }
    }
    const bool thick = i < i1 && !(alpha < alpha_thre);
    // The transmittance in front of each of the 64 samples, IN ORDER and branch-free.  A thin (or absent) sample multiplies by exactly 1, so lane j's
    // p = ((T f_0) f_1) ... f_(j-1) is the sequential march's value bit for bit: every lane runs the same 64 multiplications on factors that come
    // back from LDS as broadcast reads (all issued up front), with f_k replaced by 1 in the lanes <= k.  Sample j is kept iff it is thick and p has
    // not fallen below early_stop_eps (T never grows, so the products behind the stop - which a sequential march does not form - cannot bring a
    // sample back).  Replaces a loop with two branches and a ds_bpermute per sample, on which the kernel was latency-bound.
    bool mine = false;
    if (!(T < early_stop_eps)) {                   // (wave-uniform)
      const float f = thick ? 1.f - alpha : 1.f;
      const int w = threadIdx.x >> 6;
      sf[w][lane] = f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations complete in order; the fence is for the compiler)
      float fk[64];
#pragma unroll
      for (int k = 0; k < 64; ++k) fk[k] = sf[w][k];
      float p = T;
#pragma unroll
      for (int k = 0; k < 64; ++k) p = __fmul_rn(p, k < lane ? fk[k] : 1.f);
      mine = thick && !(p < early_stop_eps);
      T = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, __fmul_rn(p, f)), 63));
      asm volatile("" ::: "memory");               // (the next chunk overwrites the factors behind these reads)
    }
    if (i < i1) keep[i] = mine ? 1 : 0;
    c += __popcll(__ballot(mine));
  }
  if (lane == 0) counts[r] = c;
}
// counts[R] -> offsets[R+1] (exclusive prefix sums, int64), optionally the offsets of the group-aligned copy (ceil(count / 32) groups per
// ray) and the two totals in one place for the host's single read.  One block: R is a batch of rays (5 625 in the reference), and the
// torch sequence this replaces (zeros, cumsum = 2 rocprim launches, for the groups another 4) was a quarter of the launches of the
// dispatch-bound grid iteration.
// `mailbox` (host-mapped, fine-grained memory; afx_march_train_step_mse): the two totals and then `tag`, behind a system-scope fence - the host
// polls the tag instead of queueing a copy and a stream synchronisation behind the kernel.
__global__ void __launch_bounds__(1024) k_ray_offsets(const int32_t* counts, int64_t n_rays, int64_t* offsets, int64_t* group_offsets, int64_t* totals,
                                                       volatile int64_t* mailbox = nullptr, int64_t tag = 0) {
  __shared__ int64_t ws_[16], wg_[16];      // the 16 waves' totals, then their exclusive prefix
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t per = (n_rays + 1023) / 1024, r0 = t * per, r1 = r0 + per < n_rays ? r0 + per : n_rays;
  int64_t s = 0, g = 0;
  for (int64_t r = r0; r < r1; ++r) { const int64_t c = counts[r]; s += c; g += (c + 31) >> 5; }
  // inclusive scan inside the wave (6 shuffle steps), the 16 wave totals by one wave, two barriers in all
  int64_t is = s, ig = g;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t as = __shfl_up(is, d), ag = __shfl_up(ig, d);
    if (lane >= d) { is += as; ig += ag; }
  }
  if (lane == 63) { ws_[wave] = is; wg_[wave] = ig; }
  __syncthreads();
  if (wave == 0) {
    int64_t a = lane < 16 ? ws_[lane] : 0, b = lane < 16 ? wg_[lane] : 0;
    const int64_t a0 = a, b0 = b;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const int64_t as = __shfl_up(a, d), ag = __shfl_up(b, d);
      if (lane >= d) { a += as; b += ag; }
    }
    if (lane < 16) { ws_[lane] = a - a0; wg_[lane] = b - b0; }      // exclusive
  }
  __syncthreads();
  const int64_t tot_s = ws_[15] + __shfl(is, 63), tot_g = wg_[15] + __shfl(ig, 63);      // (valid in wave 15)
  s = ws_[wave] + is - s; g = wg_[wave] + ig - g;             // exclusive prefix of this thread's rays
  for (int64_t r = r0; r < r1; ++r) {
    const int64_t c = counts[r];
    offsets[r] = s;
    if (group_offsets) group_offsets[r] = g;
    s += c; g += (c + 31) >> 5;
  }
  if (t == 1023) {
    offsets[n_rays] = tot_s;
    if (group_offsets) group_offsets[n_rays] = tot_g;
    if (totals) { totals[0] = tot_s; totals[1] = tot_g; }
    if (mailbox) {
      mailbox[0] = tot_s; mailbox[1] = tot_g;
      __threadfence_system();
      mailbox[2] = tag;
    }
  }
}

__global__ void __launch_bounds__(256) k_march_compact(const uint8_t* keep, const int64_t* offsets_in, const int64_t* offsets_out, int64_t n_rays,
                                                       const float* ts_in, const float* te_in, int32_t* ri_out, float* ts_out, float* te_out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  int64_t w = offsets_out[r];
  const int64_t i1 = offsets_in[r + 1];
  for (int64_t b = offsets_in[r]; b < i1; b += 64) {
    const int64_t i = b + lane;
    const bool k = i < i1 && keep[i] != 0;
    const uint64_t bal = __ballot(k);
    if (k) {
      const int64_t o = w + lane_prefix(bal);
      ri_out[o] = (int32_t)r; ts_out[o] = ts_in[i]; te_out[o] = te_in[i];
    }
    w += __popcll(bal);
  }
}

// Group-aligned copy of a packed, ray-sorted sample list for the fused packed training step: ray r's samples start at padded index
// 32 goff[r] (goff = exclusive scan of ceil(count / 32)); the rest of its last 32-sample group is dead padding (ts = te = 0); group_ray[g] = r.
__global__ void __launch_bounds__(256) k_pack_groups(const int64_t* offsets, const int64_t* goff, int64_t n_rays, const float* ts_in, const float* te_in,
                                                     float* ts_pad, float* te_pad, int32_t* group_ray) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  const int64_t i0 = offsets[r], cnt = offsets[r + 1] - i0, g0 = goff[r], g1 = goff[r + 1];
  for (int64_t k = lane; k < (g1 - g0) * 32; k += 64) {
    ts_pad[g0 * 32 + k] = k < cnt ? ts_in[i0 + k] : 0.f;
    te_pad[g0 * 32 + k] = k < cnt ? te_in[i0 + k] : 0.f;
  }
  for (int64_t g = g0 + lane; g < g1; g += 64) group_ray[g] = (int32_t)r;
}

// --- device ray sampler (sample_pixel_rays, nerf/nerf_helpers.py:137-150): weighted sampling WITHOUT replacement of k of n
// rays.  Efraimidis-Spirakis keys: key_i = u_i^(1/w_i) (log form: log(u_i)/w_i), the k largest keys are a weighted sample
// without replacement; u from Philox (perf mode) or supplied.  The top-k selection is the radix select below.
// (blockIdx.y = batch b of afx_sample_batches: Philox stream `stream + b`, keys row b; a plain call has one row)
__global__ void k_sample_keys(const float* weights, int64_t n, const float* u_in, uint64_t seed, uint64_t stream, float* keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w = weights ? weights[i] : 1.f;
  float u = u_in ? u_in[i] : philox_uniform(seed, stream + blockIdx.y, (uint64_t)i);
  u = fmaxf(u, 5.9604645e-08f);
  keys[(int64_t)blockIdx.y * n + i] = w > 0.f ? logf(u) / w : -INFINITY;
}
// --- top-k by radix select (the sampler's selection step; a full device sort of the 900 000 keys was 13 % of the reference's
// training iteration).  Keys map to uint32 monotonically; three histogram passes (11 + 11 + 10 bits, each over the keys that
// match the prefix found so far) locate the k-th largest key T exactly; the selected set {key > T} plus the first `need_eq`
// keys equal to T (lowest index first) is then written in ASCENDING INDEX order by a counted, two-level compaction - no
// atomics on the output, so the result is deterministic.  Byte/integer work, HBM-bound: 5 reads of the key array.
// Every kernel takes blockIdx.y as a batch index (afx_sample_batches: B independent selections per launch - at 900 000 keys a selection is
// 11 launches of a few microseconds each, i.e. launch latency; B of them share the 11 launches): row b of keys [B][n], hist [B][SEL_BINS],
// state [B], counts [B][nb], output [B][k].
__device__ __forceinline__ uint32_t key_bits(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
struct SelState { uint32_t prefix, remaining, pad0, pad1; };
constexpr int SEL_BINS = 2048;
constexpr int SEL_PER_BLOCK = 1024;      // elements per block of the compaction kernels (256 threads x 4)

__global__ void __launch_bounds__(256) k_sel_init(SelState* st, uint32_t k, uint32_t* hist) {
  st += blockIdx.y; hist += (size_t)blockIdx.y * SEL_BINS;
  if (threadIdx.x == 0) { st->prefix = 0; st->remaining = k; }
  for (int i = threadIdx.x; i < SEL_BINS; i += 256) hist[i] = 0;
}
__global__ void __launch_bounds__(256) k_sel_hist(const float* keys, int64_t n, const SelState* st, uint32_t himask, int shift,
                                                  uint32_t binmask, uint32_t* hist) {
  __shared__ uint32_t h[SEL_BINS];
  keys += (int64_t)blockIdx.y * n; st += blockIdx.y; hist += (size_t)blockIdx.y * SEL_BINS;
  for (int i = threadIdx.x; i < SEL_BINS; i += 256) h[i] = 0;
  __syncthreads();
  const uint32_t prefix = st->prefix;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t u = key_bits(keys[i]);
    if ((u & himask) == prefix) atomicAdd(&h[(u >> shift) & binmask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_BINS; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}
// one block: the bin (from the top) in which the remaining rank falls; extends the prefix, leaves the histogram zeroed
__global__ void __launch_bounds__(256) k_sel_scan(uint32_t* hist, SelState* st, int shift) {
  __shared__ uint32_t part[256];
  __shared__ uint32_t above[256];      // keys in the groups above group t
  st += blockIdx.y; hist += (size_t)blockIdx.y * SEL_BINS;
  const int t = threadIdx.x;           // group t = bins [8 (255 - t), 8 (255 - t) + 7], i.e. t = 0 is the top group
  const uint32_t r = st->remaining;    // read by every thread BEFORE the barriers; one thread rewrites it behind them
  uint32_t c[8], s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { c[j] = hist[8 * (255 - t) + (7 - j)]; s += c[j]; }     // c[0] = the group's top bin
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int g = 0; g < 256; ++g) { above[g] = run; run += part[g]; }
  }
  __syncthreads();
  uint32_t a = above[t];
  if (a < r && r <= a + s) {           // exactly one group holds the r-th largest candidate
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (a < r && r <= a + c[j]) {
        st->prefix |= (uint32_t)(8 * (255 - t) + (7 - j)) << shift;
        st->remaining = r - a;          // rank inside the bin (>= 1)
      }
      a += c[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) hist[8 * t + j] = 0;
}
// per block of 1024 keys: how many are above the threshold / equal to it
__global__ void __launch_bounds__(256) k_sel_count(const float* keys, int64_t n, const SelState* st, uint32_t* cnt_gt, uint32_t* cnt_eq) {
  __shared__ uint32_t sg[256], se[256];
  keys += (int64_t)blockIdx.y * n; st += blockIdx.y; cnt_gt += (size_t)blockIdx.y * gridDim.x; cnt_eq += (size_t)blockIdx.y * gridDim.x;
  const uint32_t T = st->prefix;
  const int64_t base = (int64_t)blockIdx.x * SEL_PER_BLOCK + 4 * threadIdx.x;
  uint32_t g = 0, e = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) { const uint32_t u = key_bits(keys[base + j]); g += u > T; e += u == T; }
  sg[threadIdx.x] = g; se[threadIdx.x] = e;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if ((int)threadIdx.x < sft) { sg[threadIdx.x] += sg[threadIdx.x + sft]; se[threadIdx.x] += se[threadIdx.x + sft]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { cnt_gt[blockIdx.x] = sg[0]; cnt_eq[blockIdx.x] = se[0]; }
}
// one block: exclusive prefix sums of the per-block counts, in place
__global__ void __launch_bounds__(1024) k_sel_offsets(uint32_t* cnt_gt, uint32_t* cnt_eq, int64_t nb) {
  __shared__ uint32_t pg[1024], pe[1024];
  cnt_gt += (size_t)blockIdx.y * nb; cnt_eq += (size_t)blockIdx.y * nb;
  const int t = threadIdx.x;
  const int64_t per = (nb + 1023) / 1024, b0 = t * per, b1 = b0 + per < nb ? b0 + per : nb;
  uint32_t g = 0, e = 0;
  for (int64_t b = b0; b < b1; ++b) { g += cnt_gt[b]; e += cnt_eq[b]; }
  pg[t] = g; pe[t] = e;
  __syncthreads();
  if (t == 0) {
    uint32_t rg = 0, re = 0;
    for (int i = 0; i < 1024; ++i) { const uint32_t a = pg[i], c = pe[i]; pg[i] = rg; pe[i] = re; rg += a; re += c; }
  }
  __syncthreads();
  g = pg[t]; e = pe[t];
  for (int64_t b = b0; b < b1; ++b) { const uint32_t a = cnt_gt[b], c = cnt_eq[b]; cnt_gt[b] = g; cnt_eq[b] = e; g += a; e += c; }
}
// output position of a selected key = (# keys above T before it) + min(# keys equal to T before it, need_eq)
__global__ void __launch_bounds__(256) k_sel_write(const float* keys, int64_t n, const SelState* st, const uint32_t* off_gt,
                                                   const uint32_t* off_eq, int64_t k, int64_t* out_idx) {
  __shared__ uint32_t sg[256], se[256];
  keys += (int64_t)blockIdx.y * n; st += blockIdx.y; off_gt += (size_t)blockIdx.y * gridDim.x; off_eq += (size_t)blockIdx.y * gridDim.x;
  out_idx += (int64_t)blockIdx.y * k;
  const uint32_t T = st->prefix, need_eq = st->remaining;
  const int t = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * SEL_PER_BLOCK + 4 * t;
  uint32_t u[4], g = 0, e = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u[j] = base + j < n ? key_bits(keys[base + j]) : 0u;
    if (base + j < n) { g += u[j] > T; e += u[j] == T; }
  }
  sg[t] = g; se[t] = e;
  __syncthreads();
  if (t == 0) {      // 256-entry exclusive scan (tiny next to the key reads)
    uint32_t rg = off_gt[blockIdx.x], re = off_eq[blockIdx.x];
    for (int i = 0; i < 256; ++i) { const uint32_t a = sg[i], c = se[i]; sg[i] = rg; se[i] = re; rg += a; re += c; }
  }
  __syncthreads();
  g = sg[t]; e = se[t];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j >= n) break;
    const bool gt = u[j] > T, eq = u[j] == T;
    if (gt || (eq && e < need_eq)) {
      const int64_t pos = (int64_t)g + (e < need_eq ? e : need_eq);
      if (pos < k) out_idx[pos] = base + j;
    }
    g += gt; e += eq;
  }
}

// gather the sampled rays of a device-resident ray table (o, d, pixel) by index
__global__ void k_gather_rays(const float* org, const float* dir, const float* pix, const int64_t* idx, int64_t k, float* o_out, float* d_out, float* p_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int64_t s = idx[i];
#pragma unroll
  for (int q = 0; q < 3; ++q) { o_out[3 * i + q] = org[3 * s + q]; d_out[3 * i + q] = dir[3 * s + q]; }
  if (pix && p_out) p_out[i] = pix[s];
}
__global__ void k_philox_fill(uint64_t seed, uint64_t stream, int64_t n, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = philox_uniform(seed, stream, (uint64_t)i);
}

}  // namespace afx
