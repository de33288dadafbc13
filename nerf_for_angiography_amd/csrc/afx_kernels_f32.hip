// afx_kernels_f32.hip — exact-fp32 kernels of the hot path for gfx950 (MI355X).
//
// One workgroup = 4 wavefronts (one per SIMD, up to 512 VGPR/AGPR each) = one tile of 128
// ray-samples.  A wavefront owns 32 sample columns; the activations of ALL `width` features
// of those 32 samples live in its registers for the whole MLP, in the C/D layout of
// v_mfma_f32_32x32x2_f32 (column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).
// A layer is H_out^T[width x 32] = W[width x width] . H_in^T[width x 32]: the weight matrix
// is the MFMA A operand, streamed through LDS in "slabs" (one 32-row output tile each), and
// the previous layer's accumulators are fed straight back as the B operand — no LDS round
// trip, no lane movement: the k-order is permuted instead, and afx_prepare_weights() lays the
// slabs out in exactly that permuted order (k_lo = 32t' + R(j'), k_hi = k_lo + 4,
// R(j) = (j&3) + 8*(j>>2)).  Replaces CPPN.forward (model/CPPN.py:166-205) and the autograd
// graph behind it, plus everything around it on the hot path (see include/afx.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include "afx_internal.h"

using namespace afx;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// LDS-DMA copy of `bytes` (multiple of 4096) by the 4 waves of the workgroup: each wave
// instruction moves 64 lanes x 16 B = 1 KiB, destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds_copy(const char* g, char* l, uint32_t bytes, int wave, int lane, uint32_t nwaves = 4) {
  for (uint32_t off = (uint32_t)wave * 1024u; off < bytes; off += nwaves * 1024u)
    __builtin_amdgcn_global_load_lds(GPTR(g + off + lane * 16), LPTR(l + off), 16, 0, 0);
}

__device__ __forceinline__ int rowperm(int j) { return (j & 3) + 8 * (j >> 2); }

// Power-of-two scale of the weight-gradient contraction (f16 mode, afx_kernels_bf16.hip): the B operand is (g_n Ls) H[n] in f16, Ls = 2^-e with
// gmax = m 2^e, m in [0.5, 1), so that g_n Ls <= 1; terms more than 2^24 below the largest one flush, which no sum notices.
__device__ __forceinline__ int wgrad_scale_exp(const uint32_t* gmax_bits) {
  const float gm = __builtin_bit_cast(float, *gmax_bits);
  if (!(gm > 0.f) || !(gm < 3.0e38f)) return 0;
  int e;
  (void)frexpf(gm, &e);
  return e;
}

// One element of CPPN.pos_enc's output (model/CPPN.py:207-234) for input point (px,py,pz).
// aux (LDS): BARF [freq(3L) | weight(3L)], FOURIER [coef(3L)].
// sin(v) (want_cos = false) or cos(v) for the encodings: three-constant Cody-Waite reduction by pi/2 with FMAs and the degree-7 /
// degree-8 minimax polynomials on [-pi/4, pi/4] (cephes sinf/cosf) - a dozen instructions, no branches.  libm's sinf/cosf, inlined
// 32 times per lane (each with its large-argument path and scratch array) or called out of line (registers spilled around every
// call), made the encoding the dominant cost of a tile's prologue.  Accuracy: <= 1e-7 absolute up to |v| = 1e5 (measured against
// float64), degrading gracefully beyond as k = rint(v 2/pi) loses its last bits; fp32 arguments above 2^23 carry less than one
// radian of information in any implementation.  Arguments here: 2^k pi x (BARF) and 2 pi x coef (fourier), |x| of a few hundred.
__device__ __forceinline__ float enc_sincos(float v, bool want_cos) {
  const float kf = rintf(v * 0.636619772367581343f);          // 2 / pi
  float r = fmaf(kf, -1.57079637050628662109375f, v);         // pi/2 = C1 + C2 + C3
  r = fmaf(kf, 4.37113900018624283e-8f, r);
  r = fmaf(kf, 1.71512449441083685e-15f, r);
  const int q = ((int)kf + (want_cos ? 1 : 0)) & 3;
  const float z = r * r;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.f));
  const float res = (q & 1) ? cp : sp;
  return (q & 2) ? -res : res;
}

__device__ __forceinline__ float enc_value(int k, float px, float py, float pz, const float* aux,
                                           int enc, int n_freq, int k0) {
  if (k < 3) return k == 0 ? px : (k == 1 ? py : pz);
  if (k >= k0) return 0.f;
  int m = k - 3;
  const int nb = 3 * n_freq;
  const bool is_cos = m >= nb;
  if (is_cos) m -= nb;
  const int c = m % 3;
  const float x = c == 0 ? px : (c == 1 ? py : pz);
  if (enc == 1) {  // barf: w * sin|cos(freq * x)
    const float v = __fmul_rn(aux[m], x);
    const float s = enc_sincos(v, is_cos);
    return __fmul_rn(aux[nb + m], s);
  }
  // fourier: sin|cos(((2*pi) * x) * coef)
  const float v = __fmul_rn(__fmul_rn(6.283185307179586f, x), aux[m]);
  return enc_sincos(v, is_cos);
}

// d(enc)/d(coef) / (2 pi) of the fourier encoding (fourier_pos_enc, model/CPPN.py:320-327), as 2*nb extra "input columns":
// column m < nb: d enc[3+m] / d coef_m = 2 pi x_c cos(v);  column nb+m: d enc[3+nb+m] / d coef_m = -2 pi x_c sin(v)  (c = m % 3).
// The factor 2 pi is applied by k_reduce_coef (keeps the f16 stash of x cos(v) inside f16's range for |x| < 65 504).
__device__ __forceinline__ float enc_dcoef(int k, float px, float py, float pz, const float* aux, int n_freq) {
  const int nb = 3 * n_freq;
  if (k >= 2 * nb) return 0.f;
  const bool second = k >= nb;
  const int m = second ? k - nb : k;
  const int c = m % 3;
  const float x = c == 0 ? px : (c == 1 ? py : pz);
  const float v = __fmul_rn(__fmul_rn(6.283185307179586f, x), aux[m]);
  return second ? -x * enc_sincos(v, false) : x * enc_sincos(v, true);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Hidden-layer activation other than ReLU (CPPN act_func 'tanh' / 'sine', model/CPPN.py:53-60,278-300), forward-only kernels:
// act 1 = tanh(z), act 2 = sin(w z) with w = w0 on the first layer and 1 behind it.  (ReLU is the integer fast path at the call sites.)
__device__ __forceinline__ float act_value(float z, int act, float w) {
  return act == 1 ? tanhf(z) : enc_sincos(__fmul_rn(w, z), false);
}
// d act / dz for the backward chain of the exact-fp32 kernel: tanh' = 1 - h^2 (h = the activation's value), sin(w z)' = w cos(w z)
__device__ __forceinline__ float act_slope(float z, float h, int act, float w) {
  return act == 1 ? 1.f - h * h : w * enc_sincos(__fmul_rn(w, z), true);
}

// Per-lane description of the sample a lane's column holds.
struct Sample {
  float px, py, pz;   // query point
  float dt;           // distance weight of the sample (0 for padding)
  int ray;            // ray index (rays mode)
  int s;              // sample index on the ray
  bool live;          // contributes to outputs
};

__device__ __forceinline__ void load_ray(const ChainArgs& a, int r, float& ox, float& oy, float& oz,
                                         float& dx, float& dy, float& dz) {
  if (a.poses == nullptr) {
    ox = a.org[3 * (int64_t)r + 0]; oy = a.org[3 * (int64_t)r + 1]; oz = a.org[3 * (int64_t)r + 2];
    dx = a.dir[3 * (int64_t)r + 0]; dy = a.dir[3 * (int64_t)r + 1]; dz = a.dir[3 * (int64_t)r + 2];
    return;
  }
  // get_ray_values (phantomdata/helpers.py:156-175): float64, then cast to fp32 as
  // sample_pixel_rays does (nerf/nerf_helpers.py:147-148).
  const int64_t id = a.ray_ids ? (int64_t)a.ray_ids[r] : a.ray_id0 + r;
  const int64_t hw = (int64_t)a.width * a.height;
  const int64_t proj = id / hw;
  const int pix = (int)(id - proj * hw);
  const int jj = pix / a.width, ii = pix - jj * a.width;
  const double* M = a.poses + 12 * proj;
  const double c0 = ((double)ii - a.width * 0.5) / a.focal;
  const double c1 = -((double)jj - a.height * 0.5) / a.focal;
  const double c2 = -1.0;
  double dv[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
    dv[q] = __dadd_rn(__dadd_rn(__dmul_rn(c0, M[4 * q + 0]), __dmul_rn(c1, M[4 * q + 1])), __dmul_rn(c2, M[4 * q + 2]));
  ox = (float)M[3]; oy = (float)M[7]; oz = (float)M[11];
  dx = (float)dv[0]; dy = (float)dv[1]; dz = (float)dv[2];
}

// Counter-based uniform number (Philox4x32-10; the same function as afx_kernels_grid.hip's philox_uniform - kept here
// because this file is also compiled into the chain-kernel translation units)
__device__ __forceinline__ float chain_philox(uint64_t seed, uint64_t stream, uint64_t i) {
  uint32_t c0 = (uint32_t)(i >> 2), c1 = (uint32_t)(i >> 34), c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const uint32_t o[4] = {c0, c1, c2, c3};
  return (float)(o[i & 3] >> 8) * (1.0f / 16777216.0f);
}
// stratified depth i of AFX_DEPTH_STRATIFIED: z = near (1 - t) + far t, t = i / (S - 1); mids = .5 (z[1:] + z[:-1]);
// lower = [z0, mids], upper = [mids, z_last]; z' = lower + (upper - lower) u
__device__ __forceinline__ float strat_z(const ChainArgs& a, int i) {
  const int S = a.n_samples;
  auto zl = [&](int k) {
    const float t = __fdiv_rn((float)k, (float)(S - 1));
    return __fadd_rn(__fmul_rn(a.t_near, __fsub_rn(1.f, t)), __fmul_rn(a.t_far, t));
  };
  const float zi = zl(i);
  const float lower = i > 0 ? 0.5f * __fadd_rn(zi, zl(i - 1)) : zi;
  const float upper = i + 1 < S ? 0.5f * __fadd_rn(zl(i + 1), zi) : zi;
  const float u = chain_philox(a.jitter_seed, a.jitter_stream, (uint64_t)i);
  return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u));
}

__device__ __forceinline__ Sample make_sample(const ChainArgs& a, int64_t n) {
  Sample sp;
  sp.px = sp.py = sp.pz = 0.f; sp.dt = 0.f; sp.ray = 0; sp.s = 0; sp.live = false;
  if (n >= a.n_total) return sp;
  if (a.mode == 0) {
    sp.px = a.pts[3 * n + 0]; sp.py = a.pts[3 * n + 1]; sp.pz = a.pts[3 * n + 2];
    sp.live = true;
    return sp;
  }
  if (a.depth_mode == 4) {
    // packed group-aligned samples of the grid march: interval [ts, te) of ray group_ray[n / 32], evaluated at its mid-point exactly as
    // the reference forms its positions, o + d (t_s + t_e) / 2 (nerf/run_nerf_acc.py:290-292); padding slots (te <= ts) are dead
    const int r = a.group_ray[n >> 5];
    const float ts = a.z[n], te = a.te[n];
    sp.ray = r; sp.s = (int)n; sp.live = te > ts;
    float ox, oy, oz, dx, dy, dz;
    load_ray(a, r, ox, oy, oz, dx, dy, dz);
    const float q = __fadd_rn(ts, te);
    sp.px = __fadd_rn(ox, __fmul_rn(dx, q) * 0.5f);
    sp.py = __fadd_rn(oy, __fmul_rn(dy, q) * 0.5f);
    sp.pz = __fadd_rn(oz, __fmul_rn(dz, q) * 0.5f);
    sp.dt = sp.live ? __fsub_rn(te, ts) : 0.f;
    return sp;
  }
  const int r = (int)(n / a.s_pad);
  int s = (int)(n - (int64_t)r * a.s_pad);
  sp.ray = r;
  sp.live = s < a.n_samples;
  if (!sp.live) s = a.n_samples - 1;
  sp.s = s;
  float ox, oy, oz, dx, dy, dz;
  load_ray(a, r, ox, oy, oz, dx, dy, dz);
  if (a.depth_mode == 0) {
    // dense no-grid march of nerf_helpers_acc.py:27 and mid-point evaluation :13-15
    const float ts = __fadd_rn(a.t_near, __fmul_rn((float)s, a.t_step));
    const float te = __fadd_rn(ts, a.t_step);
    const float q = __fadd_rn(ts, te);
    sp.px = __fadd_rn(ox, __fmul_rn(dx, q) * 0.5f);
    sp.py = __fadd_rn(oy, __fmul_rn(dy, q) * 0.5f);
    sp.pz = __fadd_rn(oz, __fmul_rn(dz, q) * 0.5f);
    sp.dt = sp.live ? __fsub_rn(te, ts) : 0.f;
  } else if (a.depth_mode == 3) {
    // randomize_depth (nerf/nerf_helpers.py:13-22) of z = linspace(near, far, S) in the kernel: one jitter vector per
    // call, u_i = Philox(jitter_seed, jitter_stream)[i]; then the render_volume_density convention
    const float zs = strat_z(a, s);
    sp.px = __fadd_rn(ox, __fmul_rn(dx, zs));
    sp.py = __fadd_rn(oy, __fmul_rn(dy, zs));
    sp.pz = __fadd_rn(oz, __fmul_rn(dz, zs));
    const float dist = s + 1 < a.n_samples ? __fsub_rn(strat_z(a, s + 1), zs) : 1e10f;
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    sp.dt = sp.live ? __fmul_rn(dist, nrm) : 0.f;
  } else {
    const float* zr = a.depth_mode == 2 ? a.z + (int64_t)r * a.n_samples : a.z;
    const float zs = zr[s];
    sp.px = __fadd_rn(ox, __fmul_rn(dx, zs));
    sp.py = __fadd_rn(oy, __fmul_rn(dy, zs));
    sp.pz = __fadd_rn(oz, __fmul_rn(dz, zs));
    // render_volume_density, nerf_helpers.py:60-65: last distance 1e10, scaled by ||d||
    const float dist = s + 1 < a.n_samples ? __fsub_rn(zr[s + 1], zs) : 1e10f;
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    sp.dt = sp.live ? __fmul_rn(dist, nrm) : 0.f;
  }
  return sp;
}

// Ray parameter t of a (rays-mode) sample, p = o + t d as make_sample formed it, and the ray's direction d.
__device__ __forceinline__ void ray_param(const ChainArgs& a, const Sample& sp, float& t, float& dx, float& dy, float& dz) {
  float ox, oy, oz;
  load_ray(a, sp.ray, ox, oy, oz, dx, dy, dz);
  if (a.depth_mode == 4) {
    t = __fadd_rn(a.z[sp.s], a.te[sp.s]) * 0.5f;      // (packed: sp.s is the sample's index in the padded list)
  } else if (a.depth_mode == 0) {
    const float ts = __fadd_rn(a.t_near, __fmul_rn((float)sp.s, a.t_step));
    t = __fadd_rn(ts, __fadd_rn(ts, a.t_step)) * 0.5f;
  } else if (a.depth_mode == 3) {
    t = strat_z(a, sp.s);
  } else {
    const float* zr = a.depth_mode == 2 ? a.z + (int64_t)sp.ray * a.n_samples : a.z;
    t = zr[sp.s];
  }
}

// ---------------------------------------------------------------------------------------
// Fused chain kernel.  BWD=false: inference / forward.  BWD=true: recompute the forward,
// then the input-gradient chain, stashing H_l and dZ_l of the tile for the weight-gradient
// contraction over samples (k_wgrad_f32).
// tanh / sine models (a.act != 0) train here too: where ReLU keeps one mask bit per activation in LDS, the forward half writes the
// activation's slope d act / dz into the SLOT OF dZ_l (same lane, same address the lane later stores dZ_l to), and the backward half reads
// it back just before it overwrites it - no extra memory, and the weight-gradient kernel sees the same two stashes.
// ---------------------------------------------------------------------------------------
template <int F, bool BWD>
__global__ void __launch_bounds__(256, 1) k_chain_f32(const ChainArgs a) {
  constexpr int NT = F / 32;          // 32-row tiles per layer
  constexpr int MW = (NT + 1) / 2;    // mask words per layer per lane (16 bits per tile)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, hh = lane >> 5;
  const int N = a.n_hidden;
  if (a.tile0 + (int)blockIdx.x >= a.tile1) return;

  float* sm = (float*)lds;
  for (uint32_t i = tid * 4; i < a.small_floats; i += 1024) *(f32x4*)(sm + i) = *(const f32x4*)(a.small + i);
  char* slot0 = lds + a.small_bytes_pad;
  uint32_t* mk = (uint32_t*)(slot0 + 2 * (size_t)a.slot_bytes);   // [(l*MW + w)*256 + tid]
  const float* bias_perm = sm;                    // [((l*2 + hh)*NT + t)*16 + j]
  const float* wout_perm = sm + (N + 1) * F;      // [(hh*NT + t)*16 + j]
  const float* aux = sm + (N + 2) * F + 4;

  const int steps_per_tile = 1 + N * NT * (BWD ? 2 : 1);
  int seq = 0;          // index (within the tile's slab sequence) of the step about to start
  uint32_t par = 0;     // ring slot holding that step's slab
  bool has_next = false;

  // Every step: wait for this step's slab, barrier (also retires all readers of the other
  // slot), start the LDS-DMA of the next slab into the other slot, return this step's slab.
  auto step_begin = [&]() -> const char* {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int ni = seq + 1;
    bool doload = true;
    if (ni == steps_per_tile) { ni = 0; doload = has_next; }
    if (doload) {
      const char* src;
      uint32_t bytes = a.slabh_bytes;
      if (ni == 0) { src = a.stream_fwd; bytes = a.slab0_bytes; }
      else if (ni <= N * NT) src = a.stream_fwd + a.slab0_bytes + (size_t)(ni - 1) * a.slabh_bytes;
      else src = a.stream_bwd + (size_t)(ni - N * NT - 1) * a.slabh_bytes;
      glds_copy(src, slot0 + (par ^ 1u) * (size_t)a.slot_bytes, bytes, wave, lane);
    }
    const char* cur = slot0 + par * (size_t)a.slot_bytes;
    par ^= 1u;
    seq = ni;
    return cur;
  };

  glds_copy(a.stream_fwd, slot0, a.slab0_bytes, wave, lane);

  for (int tile = a.tile0 + blockIdx.x; tile < a.tile1; tile += gridDim.x) {
    has_next = tile + (int)gridDim.x < a.tile1;
    const int64_t n = (int64_t)tile * TILE + wave * GROUP + col;
    const int64_t m = (int64_t)(tile - a.tile0) * TILE + wave * GROUP + col;   // stash row
    const Sample sp = make_sample(a, n);

    f32x16 h[NT];
    // ---------------- layer 0: K0 encoded inputs, natural k order, k = 2q + (lane>>5)
    {
      const float* s0 = (const float*)step_begin();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4* bp = (const f32x4*)(bias_perm + ((0 * 2 + hh) * NT + t) * 16);
        const f32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        h[t] = (f32x16){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3],
                        b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
      }
      for (int q = 0; q < a.nq; ++q) {
        const float e = enc_value(2 * q + hh, sp.px, sp.py, sp.pz, aux, a.enc, a.n_freq, a.k0);
        if (BWD) a.stash_e[m * (2 * a.nq) + 2 * q + hh] = e;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          h[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(s0[(q * NT + t) * 64 + lane], e, h[t], 0, 0, 0);
      }
      uint32_t mw[MW];
#pragma unroll
      for (int w = 0; w < MW; ++w) mw[w] = 0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (a.act != 0) {      // tanh / sine
          f32x16 sl0;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float z = h[t][j];
            h[t][j] = act_value(z, a.act, a.act_w0);
            if (BWD) sl0[j] = act_slope(z, h[t][j], a.act, a.act_w0);
          }
          if (BWD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const size_t at = ((size_t)0 * a.stash_rows + m) * F + 32 * t + 8 * q + 4 * hh;
              *(f32x4*)(a.stash_h + at) = (f32x4){h[t][4 * q], h[t][4 * q + 1], h[t][4 * q + 2], h[t][4 * q + 3]};
              *(f32x4*)(a.stash_dz + at) = (f32x4){sl0[4 * q], sl0[4 * q + 1], sl0[4 * q + 2], sl0[4 * q + 3]};
            }
          }
          continue;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          // integer VALU only (see afx_kernels_bf16.hip): max_i32(bits(x),0) = bits(relu(x)); min_u32(.,1) = [x > 0]
          const int ri = max(__float_as_int(h[t][j]), 0);
          h[t][j] = __int_as_float(ri);
          if (BWD) mw[t >> 1] |= min((uint32_t)ri, 1u) << (16 * (t & 1) + j);
        }
        if (BWD) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(f32x4*)(a.stash_h + ((size_t)0 * a.stash_rows + m) * F + 32 * t + 8 * q + 4 * hh) =
                (f32x4){h[t][4 * q], h[t][4 * q + 1], h[t][4 * q + 2], h[t][4 * q + 3]};
        }
      }
      if (BWD) {
#pragma unroll
        for (int w = 0; w < MW; ++w) mk[(0 * MW + w) * 256 + tid] = mw[w];
      }
    }
    // ---------------- hidden layers 1..N
    for (int l = 1; l <= N; ++l) {
      f32x16 hn[NT];
      uint32_t mw[MW];
#pragma unroll
      for (int w = 0; w < MW; ++w) mw[w] = 0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4* sl = (const f32x4*)step_begin();
        const f32x4* bp = (const f32x4*)(bias_perm + ((l * 2 + hh) * NT + t) * 16);
        const f32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        f32x16 acc = (f32x16){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3],
                              b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
#pragma unroll
        for (int u = 0; u < NT * 4; ++u) {
          const f32x4 a4 = sl[u * 64 + lane];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], h[u >> 2][4 * (u & 3) + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], h[u >> 2][4 * (u & 3) + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], h[u >> 2][4 * (u & 3) + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], h[u >> 2][4 * (u & 3) + 3], acc, 0, 0, 0);
        }
        if (a.act != 0) {
          f32x16 sl1;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            hn[t][j] = act_value(acc[j], a.act, 1.f);
            if (BWD) sl1[j] = act_slope(acc[j], hn[t][j], a.act, 1.f);
          }
          if (BWD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const size_t at = ((size_t)l * a.stash_rows + m) * F + 32 * t + 8 * q + 4 * hh;
              *(f32x4*)(a.stash_h + at) = (f32x4){hn[t][4 * q], hn[t][4 * q + 1], hn[t][4 * q + 2], hn[t][4 * q + 3]};
              *(f32x4*)(a.stash_dz + at) = (f32x4){sl1[4 * q], sl1[4 * q + 1], sl1[4 * q + 2], sl1[4 * q + 3]};
            }
          }
          continue;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int ri = max(__float_as_int(acc[j]), 0);
          hn[t][j] = __int_as_float(ri);
          if (BWD) mw[t >> 1] |= min((uint32_t)ri, 1u) << (16 * (t & 1) + j);
        }
        if (BWD) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(f32x4*)(a.stash_h + ((size_t)l * a.stash_rows + m) * F + 32 * t + 8 * q + 4 * hh) =
                (f32x4){hn[t][4 * q], hn[t][4 * q + 1], hn[t][4 * q + 2], hn[t][4 * q + 3]};
        }
      }
      if (BWD) {
#pragma unroll
        for (int w = 0; w < MW; ++w) mk[(l * MW + w) * 256 + tid] = mw[w];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) h[t] = hn[t];
    }
    // ---------------- output layer (width -> 1) on the VALU, exact fp32
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4* wp = (const f32x4*)(wout_perm + (hh * NT + t) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w4 = wp[q];
        dot = fmaf(h[t][4 * q + 0], w4[0], dot);
        dot = fmaf(h[t][4 * q + 1], w4[1], dot);
        dot = fmaf(h[t][4 * q + 2], w4[2], dot);
        dot = fmaf(h[t][4 * q + 3], w4[3], dot);
      }
    }
    dot += __shfl_xor(dot, 32);
    const float raw = dot + sm[(N + 2) * F];

    float g = 0.f;
    if (a.mode == 0) {
      if (!BWD) {
        if (hh == 0 && sp.live) a.out[n] = a.apply_sigmoid ? sigmoidf_(raw) : raw;
        continue;
      }
      g = sp.live ? a.dod[n] : 0.f;      // points mode: dL/draw is supplied per point
    } else {
      // ---------------- Beer-Lambert: optical depth of this wave's 32 samples of one ray
      const float sig = sigmoidf_(raw);
      const float tau = sp.live ? __fmul_rn(sig, sp.dt) : 0.f;
      if (hh == 0 && sp.live) {
        if (a.sigma) a.sigma[(int64_t)sp.ray * a.n_samples + sp.s] = sig;
        if (a.tau) a.tau[(int64_t)sp.ray * a.n_samples + sp.s] = tau;
      }
      if (!BWD) {
        float od = tau;
#pragma unroll
        for (int sh = 16; sh >= 1; sh >>= 1) od += __shfl_xor(od, sh);
        if (lane == 0 && n < a.n_total) {
          const int gpr = a.s_pad / GROUP;
          a.od_part[(int64_t)sp.ray * gpr + (int)((n - (int64_t)sp.ray * a.s_pad) / GROUP)] = od;
        }
        continue;
      }
      if (sp.live) g = a.dod[sp.ray] * sp.dt * (sig * (1.f - sig));
    }
    if (BWD) {
      // ---------------- backward: dL/draw, then the input-gradient chain
      if (hh == 0) a.graw[m] = g;
      f32x16 dz[NT];
      // tanh / sine: the slopes of layer l, as this lane left them in dZ_l's slot (every forward store has completed: step_begin waits
      // for vmcnt(0) at each of the steps since)
      auto slopes = [&](int l, int t, int q) { return *(const f32x4*)(a.stash_dz + ((size_t)l * a.stash_rows + m) * F + 32 * t + 8 * q + 4 * hh); };
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint32_t bits = a.act != 0 ? 0u : mk[(N * MW + (t >> 1)) * 256 + tid] >> (16 * (t & 1));
        const f32x4* wp = (const f32x4*)(wout_perm + (hh * NT + t) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w4 = wp[q];
          if (a.act != 0) {
            const f32x4 s4 = slopes(N, t, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[t][4 * q + e] = w4[e] * g * s4[e];
            continue;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            dz[t][4 * q + e] = __int_as_float(__float_as_int(w4[e] * g) & (((int)(bits << (31 - (4 * q + e)))) >> 31));
        }
      }
      for (int l = N; l >= 1; --l) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(f32x4*)(a.stash_dz + ((size_t)l * a.stash_rows + m) * F + 32 * t + 8 * q + 4 * hh) =
                (f32x4){dz[t][4 * q], dz[t][4 * q + 1], dz[t][4 * q + 2], dz[t][4 * q + 3]};
        f32x16 dn[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4* sl = (const f32x4*)step_begin();
          f32x16 acc = {0.f};
#pragma unroll
          for (int u = 0; u < NT * 4; ++u) {
            const f32x4 a4 = sl[u * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], dz[u >> 2][4 * (u & 3) + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], dz[u >> 2][4 * (u & 3) + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], dz[u >> 2][4 * (u & 3) + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], dz[u >> 2][4 * (u & 3) + 3], acc, 0, 0, 0);
          }
          if (a.act != 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 s4 = slopes(l - 1, t, q);
#pragma unroll
              for (int e = 0; e < 4; ++e) dn[t][4 * q + e] = acc[4 * q + e] * s4[e];
            }
            continue;
          }
          const uint32_t bits = mk[((l - 1) * MW + (t >> 1)) * 256 + tid] >> (16 * (t & 1));
#pragma unroll
          for (int j = 0; j < 16; ++j) dn[t][j] = __int_as_float(__float_as_int(acc[j]) & (((int)(bits << (31 - j))) >> 31));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) dz[t] = dn[t];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(a.stash_dz + ((size_t)0 * a.stash_rows + m) * F + 32 * t + 8 * q + 4 * hh) =
              (f32x4){dz[t][4 * q], dz[t][4 * q + 1], dz[t][4 * q + 2], dz[t][4 * q + 3]};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------
// Weight gradients: dW_l[o][i] = sum_n dZ_l[n][o] * H_{l-1}[n][i]  (layer 0: B = encoded inputs).
// Contraction over samples, split over blockIdx.x; each split writes a partial that
// k_reduce sums in fixed order (deterministic).  grid = (n_splits, N+1), block = 512.
// 32-sample chunks of both stashes are staged in LDS by LDS-DMA (double-buffered); the MFMA
// operands (A[i][k] = dZ[n+k][o0+i], B[k][j] = H[n+k][i0+j], k = lane>>5) are conflict-free
// ds_read_b32 (consecutive lanes -> consecutive features).
// ---------------------------------------------------------------------------------------
template <int F>
__global__ void __launch_bounds__(512) k_wgrad_f32(const WgradArgs a) {
  constexpr int NT = F / 32;
  constexpr int TPW = (NT * NT + 7) / 8;
  constexpr int KB = 32;                      // samples per stage
  constexpr int STAGE = 2 * KB * F;           // floats per stage: A chunk then B chunk
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float* lds = (float*)lds_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, hh = lane >> 5;
  const int layer = blockIdx.y, split = blockIdx.x;
  const float* A = a.stash_dz + (size_t)layer * a.stride_rows * F;
  const float* B;
  int ldb, ncols;
  if (layer == 0) { B = a.stash_e; ldb = a.k0pad; ncols = a.k0pad; }
  else { B = a.stash_h + (size_t)(layer - 1) * a.stride_rows * F; ldb = F; ncols = F; }
  const int ntb = (ncols + 31) / 32;
  int64_t r0 = (int64_t)split * a.rows_per_split;
  int64_t r1 = r0 + a.rows_per_split;
  if (r1 > a.rows) r1 = a.rows;
  const int nst = r1 > r0 ? (int)((r1 - r0) / KB) : 0;

  f32x16 acc[TPW];
  int to[TPW], ti[TPW];
  bool ok[TPW];
#pragma unroll
  for (int mm = 0; mm < TPW; ++mm) {
    acc[mm] = (f32x16){0.f};
    const int idx = wave + 8 * mm;
    ok[mm] = idx < NT * ntb;
    to[mm] = ok[mm] ? idx / ntb : 0;
    ti[mm] = ok[mm] ? idx % ntb : 0;
  }
  auto stage_load = [&](int st, int buf) {
    float* dA = lds + buf * STAGE;
    float* dB = dA + KB * F;
    const char* gA = (const char*)(A + (r0 + (int64_t)st * KB) * F);
    for (int off = wave * 1024; off < KB * F * 4; off += 8192)
      __builtin_amdgcn_global_load_lds(GPTR(gA + off + lane * 16), LPTR((char*)dA + off), 16, 0, 0);
    if (layer != 0) {
      const char* gB = (const char*)(B + (r0 + (int64_t)st * KB) * F);
      for (int off = wave * 1024; off < KB * F * 4; off += 8192)
        __builtin_amdgcn_global_load_lds(GPTR(gB + off + lane * 16), LPTR((char*)dB + off), 16, 0, 0);
    } else {
      const float* gB = B + (r0 + (int64_t)st * KB) * ldb;
      for (int i = tid; i < KB * ldb; i += 512) dB[i] = gB[i];
    }
  };
  if (nst > 0) stage_load(0, 0);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < nst) stage_load(st + 1, (st + 1) & 1);
    const float* sA = lds + (st & 1) * STAGE;
    const float* sB = sA + KB * F;
#pragma unroll 4
    for (int k = 0; k < KB; k += 2) {
#pragma unroll
      for (int mm = 0; mm < TPW; ++mm) {
        if (ok[mm]) {
          const float av = sA[(k + hh) * F + 32 * to[mm] + col];
          const int cb = 32 * ti[mm] + col;
          const float bv = cb < ncols ? sB[(k + hh) * ldb + cb] : 0.f;
          acc[mm] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mm], 0, 0, 0);
        }
      }
    }
  }
  float* P = a.partial + ((size_t)layer * a.n_splits + split) * F * F;
#pragma unroll
  for (int mm = 0; mm < TPW; ++mm) {
    if (!ok[mm]) continue;
    const int cb = 32 * ti[mm] + col;
    if (cb >= ncols) continue;
#pragma unroll
    for (int j = 0; j < 16; ++j) P[(size_t)(32 * to[mm] + rowperm(j) + 4 * hh) * ncols + cb] = acc[mm][j];
  }
}

// Column sums over samples: bias gradients (y = 0..N: sum_n dZ_y[n][f]) and the output layer
// (y = N+1: sum_n graw[n] * H_N[n][f], plus sum_n graw[n] in slot F).  grid = (n_splits, N+2),
// block = F threads.
template <int F>
__global__ void __launch_bounds__(1024) k_colsum_f32(const WgradArgs a) {
  constexpr int G = 1024 / F;                 // row groups per block
  __shared__ float red[1024 + 32];
  const int f = threadIdx.x % F, gq = threadIdx.x / F, y = blockIdx.y, split = blockIdx.x;
  int64_t r0 = (int64_t)split * a.rows_per_split;
  int64_t r1 = r0 + a.rows_per_split;
  if (r1 > a.rows) r1 = a.rows;
  float s = 0.f, sg = 0.f;
  if (y <= a.n_hidden) {
    const float* X = a.stash_dz + (size_t)y * a.stride_rows * F;
#pragma unroll 8
    for (int64_t r = r0 + gq; r < r1; r += G) s += X[r * F + f];
  } else {
    const float* X = a.stash_h + (size_t)a.n_hidden * a.stride_rows * F;
#pragma unroll 8
    for (int64_t r = r0 + gq; r < r1; r += G) {
      const float g = a.graw[r];
      s = fmaf(g, X[r * F + f], s);
      sg += g;
    }
  }
  red[threadIdx.x] = s;
  if (f == 0) red[1024 + gq] = sg;
  __syncthreads();
  if (gq == 0) {
    float t = 0.f, tg = 0.f;
    for (int q = 0; q < G; ++q) { t += red[q * F + f]; if (f == 0) tg += red[1024 + q]; }
    float* P = a.partial2 + ((size_t)y * a.n_splits + split) * (F + 4);
    P[f] = t;
    if (y == a.n_hidden + 1 && f == 0) P[F] = tg;
  }
}

// grad += sum over splits (fixed order).  grid = (ceil(F*F/256), N+1) for weights.
// (bodies as device functions: k_reduce_all - one launch for the three reductions of the 8-bit-stash path - calls them by block range)
template <int F>
__device__ __forceinline__ void reduce_w_body(const ReduceArgs& a, int bx, int layer, int tx) {
  if (a.hidden_only && layer == 0 && !a.layer0_mfma) return;
  const int ncols = layer == 0 ? a.k0pad : F;
  const int ncr = layer == 0 ? a.k0 : F;
  const int e = bx * 256 + tx;
  if (e >= F * ncols) return;
  const int row = e / ncols, c = e % ncols;
  if (c >= ncr) return;
  float s = 0.f;
#pragma unroll 8      // (independent loads in flight: the loop is latency-bound at the reference's batch sizes)
  for (int sp = 0; sp < a.n_splits; ++sp) s += a.partial[((size_t)layer * a.n_splits + sp) * F * F + e];
  // flat layout: W0[F,k0] b0[F] then (W_l[F,F] b_l[F])*, Wout[F] bout
  size_t off = layer == 0 ? 0 : (size_t)F * a.k0 + F + (size_t)(layer - 1) * (F * F + F);
  if (a.gmax) s *= ldexpf(1.f, wgrad_scale_exp(a.gmax) - a.scale_shift);       // f16 mode (hidden layers only reach here): undo Ls, exact
  a.grad[off + (size_t)row * ncr + c] += s;
}
template <int F>
__global__ void k_reduce_w(const ReduceArgs a) { reduce_w_body<F>(a, blockIdx.x, blockIdx.y, threadIdx.x); }

// biases + output layer.  grid = (1, N+2), block = F.
template <int F>
__device__ __forceinline__ void reduce_b_body(const ReduceArgs& a, int y, int f) {
  if (a.hidden_only && ((y == 0 && !a.layer0_mfma) || y == a.n_hidden + 1)) return;
  float s = 0.f, sg = 0.f;
#pragma unroll 8
  for (int sp = 0; sp < a.n_splits; ++sp) {
    const float* P = a.partial2 + ((size_t)y * a.n_splits + sp) * (F + 4);
    s += P[f];
    if (y == a.n_hidden + 1 && f == 0) sg += P[F];
  }
  const size_t hidden0 = (size_t)F * a.k0 + F;
  if (a.gmax) s *= ldexpf(1.f, wgrad_scale_exp(a.gmax) - a.scale_shift);
  if (y == 0) a.grad[(size_t)F * a.k0 + f] += s;
  else if (y <= a.n_hidden) a.grad[hidden0 + (size_t)(y - 1) * (F * F + F) + (size_t)F * F + f] += s;
  else {
    const size_t wout = hidden0 + (size_t)a.n_hidden * (F * F + F);
    a.grad[wout + f] += s;
    if (f == 0) a.grad[wout + F] += sg;
  }
}
template <int F>
__global__ void k_reduce_b(const ReduceArgs a) { reduce_b_body<F>(a, blockIdx.y, threadIdx.x); }

#ifndef AFX_TEMPLATES_ONLY      // plain kernels below: defined once, in the host translation unit
// ---------------------------------------------------------------------------------------
// Weight re-tiling (afx_prepare_weights), fp32.
// ---------------------------------------------------------------------------------------
struct PrepArgs {
  const float* params;
  const float* enc_aux;
  char* prepared;
  int32_t F, n_hidden, k0, nq, enc, n_freq, weights;   // weights = 0: only the `small` section (bf16 modes)
  uint32_t small_off, slab0_off, fwd_off, bwd_off, slab0_bytes, slabh_bytes, small_floats;
};

__device__ __forceinline__ void prepare_f32_body(const PrepArgs& p, int64_t gid, int64_t gsz) {      // (grid-stride over gsz threads)
  const int F = p.F, NT = F / 32, N = p.n_hidden;
  const size_t hidden0 = (size_t)F * p.k0 + F;
  const size_t stride = (size_t)F * F + F;
  // small: bias_perm[(l*2+h)*NT+t][16] | wout_perm[(h*NT+t)][16] | bout,0,0,0 | aux
  float* sm = (float*)(p.prepared + p.small_off);
  for (int64_t i = gid; i < p.small_floats; i += gsz) {
    float v = 0.f;
    const int64_t nb = (int64_t)(N + 1) * F;
    if (i < nb) {
      const int l = (int)(i / F), r = (int)(i % F);
      const int h = r / (NT * 16), t = (r / 16) % NT, j = r % 16;
      const int row = 32 * t + rowperm(j) + 4 * h;
      const size_t boff = l == 0 ? (size_t)F * p.k0 : hidden0 + (size_t)(l - 1) * stride + (size_t)F * F;
      v = p.params[boff + row];
    } else if (i < nb + F) {
      const int r = (int)(i - nb);
      const int h = r / (NT * 16), t = (r / 16) % NT, j = r % 16;
      v = p.params[hidden0 + (size_t)N * stride + 32 * t + rowperm(j) + 4 * h];
    } else if (i == nb + F) {
      v = p.params[hidden0 + (size_t)N * stride + F];
    } else if (i >= nb + F + 4) {
      const int64_t k = i - (nb + F + 4);
      const int naux = p.enc == 1 ? 6 * p.n_freq : (p.enc == 2 ? 3 * p.n_freq : 0);
      if (k < naux && p.enc_aux) v = p.enc_aux[k];
    }
    sm[i] = v;
  }
  if (!p.weights) return;
  // slab0: [q][t][lane] = W0[32t + (lane&31)][2q + (lane>>5)]
  float* s0 = (float*)(p.prepared + p.slab0_off);
  for (int64_t i = gid; i < p.slab0_bytes / 4; i += gsz) {
    const int lane = (int)(i & 63);
    const int64_t qt = i >> 6;
    const int t = (int)(qt % NT), q = (int)(qt / NT);
    const int k = 2 * q + (lane >> 5);
    float v = 0.f;
    if (q < p.nq && k < p.k0) v = p.params[(size_t)(32 * t + (lane & 31)) * p.k0 + k];
    s0[i] = v;
  }
  // hidden slabs: element ((u*64 + lane)*4 + c), u = t'*4 + q: j' = 4q + c,
  //   fwd (layer l, tile t):  W_l[32t + (lane&31)][32t' + R(j') + 4(lane>>5)]
  //   bwd (layer l, tile t):  W_l[32t' + R(j') + 4(lane>>5)][32t + (lane&31)]
  const int64_t per_slab = p.slabh_bytes / 4;
  const int64_t nslab = (int64_t)N * NT;
  float* fw = (float*)(p.prepared + p.fwd_off);
  float* bw = (float*)(p.prepared + p.bwd_off);
  for (int64_t i = gid; i < 2 * nslab * per_slab; i += gsz) {
    const bool isb = i >= nslab * per_slab;
    const int64_t ii = isb ? i - nslab * per_slab : i;
    const int64_t slab = ii / per_slab;
    const int e = (int)(ii % per_slab);
    const int c = e & 3, lane = (e >> 2) & 63, u = e >> 8;
    const int tp = u >> 2, q = u & 3, jp = 4 * q + c;
    const int t = (int)(slab % NT);
    const int lidx = (int)(slab / NT);
    const int l = isb ? N - lidx : 1 + lidx;
    const float* W = p.params + hidden0 + (size_t)(l - 1) * stride;
    const int r = 32 * t + (lane & 31);
    const int k = 32 * tp + rowperm(jp) + 4 * (lane >> 5);
    (isb ? bw : fw)[ii] = isb ? W[(size_t)k * F + r] : W[(size_t)r * F + k];
  }
}
__global__ void k_prepare_f32(const PrepArgs p) { prepare_f32_body(p, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x); }

// ---------------------------------------------------------------------------------------
// Small per-ray kernels.
// ---------------------------------------------------------------------------------------
// pixel[r] = exp(-sum_g od_part[r][g])   (== prod_i exp(-sigma_i dt_i), nerf_helpers_acc.py:53-58)
__global__ void k_finish_fwd(const float* od_part, int groups, int64_t n_rays, float* pixel) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float od = 0.f;
  for (int g = 0; g < groups; ++g) od += od_part[r * groups + g];
  pixel[r] = expf(-od);
}

// split training step (rays that straddle tiles): pixel = exp(-sum of the ray's group partials) and, for
// L = inv_n sum_r (pixel_r - target_r)^2, dL/d(optical depth) = -pixel * 2 (pixel - target) inv_n  (nerf/run_nerf_acc.py:296-298,306)
__global__ void k_finish_mse(const float* od_part, int groups, int64_t n_rays, const float* target, float inv_n, float* pixel, float* dod) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float od = 0.f;
  for (int g = 0; g < groups; ++g) od += od_part[r * groups + g];
  const float T = expf(-od);
  pixel[r] = T;
  dod[r] = -T * (2.f * (T - target[r]) * inv_n);
}

// the same for packed samples: ray r owns the groups [goff[r], goff[r+1]) (none: pixel = 1, the empty product of scatter_mul into ones,
// nerf/nerf_helpers_acc.py:58)
__global__ void k_finish_mse_packed(const float* od_part, const int64_t* goff, int64_t n_rays, const float* target, float inv_n, float* pixel, float* dod) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float od = 0.f;
  for (int64_t g = goff[r]; g < goff[r + 1]; ++g) od += od_part[g];
  const float T = expf(-od);
  pixel[r] = T;
  dod[r] = -T * (2.f * (T - target[r]) * inv_n);
}

// Hierarchical step with coarse re-use: per ray, merge the S coarse depths and the NF new ones (coarse first on ties, as k_fine_depths), form the step
// lengths of the merged list (last: 1e10) x ||d|| (render_volume_density, nerf/nerf_helpers.py:60-65), the optical depth, pixel = exp(-od), the MSE
// gradient, and the finished dL/draw = dod dt sigma (1 - sigma) of every sample of both sets, written to the samples' stash rows
// (gA[(r - ray0) spadA + i], gB[(r - ray0) spadB + k]).  One thread per ray, two walks over the S + NF depths.
__global__ void k_hier_composite(const ChainArgs a, int64_t ray0, int64_t n_rays, const float* zf, int NF, const float* sig_c, const float* sig_f,
                                 const float* target, float inv_n, float* pixel, float* gA, int spadA, float* gB, int spadB, float* z_all) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays) return;
  const int64_t r = ray0 + i;
  const int S = a.n_samples;
  float ox, oy, oz, dx, dy, dz;
  load_ray(a, (int)r, ox, oy, oz, dx, dy, dz);
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float* zc = a.depth_mode == 2 ? a.z + r * S : a.z;
  const float* zn = zf + r * NF;
  const float* sc = sig_c + r * S;
  const float* sf = sig_f + r * NF;
  float dod = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    int ic = 0, kf = 0;
    float od = 0.f, pz = 0.f, psig = 0.f;
    int pset = 0, pidx = 0;
    for (int p = 0; p <= S + NF; ++p) {
      float z = 0.f, sg = 0.f;
      int set = 0, idx = 0;
      if (p < S + NF) {
        if (kf >= NF || (ic < S && zc[ic] <= zn[kf])) { z = zc[ic]; sg = sc[ic]; set = 0; idx = ic++; }
        else { z = zn[kf]; sg = sf[kf]; set = 1; idx = kf++; }
        if (pass == 1 && z_all) z_all[r * (S + NF) + p] = z;
      }
      if (p > 0) {      // the element before this one now knows its step: to this element, or 1e10 behind the last
        const float dist = p < S + NF ? __fsub_rn(z, pz) : 1e10f;
        const float dt = __fmul_rn(dist, nrm);
        if (pass == 0) od += __fmul_rn(psig, dt);
        else {
          const float gv = dod * dt * (psig * (1.f - psig));
          if (pset == 0) gA[i * spadA + pidx] = gv; else gB[i * spadB + pidx] = gv;
        }
      }
      pz = z; psig = sg; pset = set; pidx = idx;
    }
    if (pass == 0) {
      const float T = expf(-od);
      pixel[r] = T;
      dod = -T * (2.f * (T - target[r]) * inv_n);
    }
  }
}

// dL/d(optical depth) = -pixel * dL/dpixel
__global__ void k_finish_bwd(const float* pixel, const float* dpix, int64_t n_rays, float* dod) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  dod[r] = -pixel[r] * dpix[r];
}

// render_volume_density, C == 1 branch (nerf/nerf_helpers.py:59-123), one thread per ray.
__global__ void k_composite_dense(const float* raw, const float* dirs, const float* z, int z_per_ray,
                                  int64_t n_rays, int S, float* rgb_map, float* depth_map, float* weights,
                                  float* entropy, float* sigma_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float* zr = z_per_ray ? z + r * S : z;
  const float* rw = raw + r * S;
  float T = 1.f, dep = 0.f, ssum = 0.f;
  for (int s = 0; s < S; ++s) {
    const float dist = s + 1 < S ? __fsub_rn(zr[s + 1], zr[s]) : 1e10f;
    const float nd = __fmul_rn(dist, nrm);
    const float sg = sigmoidf_(rw[s]);
    const float al = expf(-__fmul_rn(sg, nd));
    if (sigma_out) sigma_out[r * S + s] = sg;
    if (weights) weights[r * S + s] = __fmul_rn(__fadd_rn(__fsub_rn(1.f, al), 1e-10f), T);
    T = __fmul_rn(T, al);
    dep = __fadd_rn(dep, __fmul_rn(al, zr[s]));
    ssum = __fadd_rn(ssum, sg);
  }
  rgb_map[r] = T;
  if (depth_map) depth_map[r] = dep;
  if (entropy) {
    float ent = 0.f;
    const float den = __fadd_rn(ssum, 1e-10f);
    for (int s = 0; s < S; ++s) {
      const float dn = sigmoidf_(rw[s]) / den;
      ent = __fadd_rn(ent, __fmul_rn(dn, logf(__fadd_rn(dn, 1e-10f))));
    }
    entropy[r] = (1.f - T) > 0.4f ? -ent : 0.f;
  }
}

__global__ void k_composite_dense_bwd(const float* raw, const float* dirs, const float* z, int z_per_ray,
                                      int64_t n_rays, int S, const float* rgb_map, const float* d_rgb,
                                      float* d_raw) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays * S) return;
  const int64_t r = i / S;
  const int s = (int)(i - r * S);
  const float dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float* zr = z_per_ray ? z + r * S : z;
  const float dist = s + 1 < S ? __fsub_rn(zr[s + 1], zr[s]) : 1e10f;
  const float sg = sigmoidf_(raw[i]);
  const float gT = rgb_map[r] * d_rgb[r];
  d_raw[i] = gT == 0.f ? 0.f : -gT * (dist * nrm) * (sg * (1.f - sg));
}

// acc_render_volume_density (nerf/nerf_helpers_acc.py:45-63) for packed, ray-sorted samples.
__device__ __forceinline__ int64_t lower_bound_i32(const int32_t* a, int64_t n, int32_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}
__global__ void k_composite_packed(const float* pred, const int32_t* ri, const float* ts, const float* te,
                                   int64_t n, int64_t n_rays, float* rgb_map) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float T = 1.f;
  for (int64_t i = lower_bound_i32(ri, n, (int32_t)r); i < n && ri[i] == (int32_t)r; ++i)
    T = __fmul_rn(T, expf(-__fmul_rn(sigmoidf_(pred[i]), __fsub_rn(te[i], ts[i]))));
  rgb_map[r] = T;
}
__global__ void k_composite_packed_bwd(const float* pred, const int32_t* ri, const float* ts, const float* te,
                                       int64_t n, const float* rgb_map, const float* d_rgb, float* d_pred) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = ri[i];
  const float sg = sigmoidf_(pred[i]);
  d_pred[i] = -(rgb_map[r] * d_rgb[r]) * __fsub_rn(te[i], ts[i]) * (sg * (1.f - sg));
}

// Ground-truth X-ray projector over a voxel volume — ray_tracing, phantomdata/helpers.py:192-224.
// mu(p) by trilinear interpolation on a regular grid with a constant fill value outside (scipy
// RegularGridInterpolator(method='linear', bounds_error=False, fill_value)), float64 coordinates as upstream;
// 'ct': img = prod_s exp(-mu * dz_s * ||d||) with dz_last = 1e10; otherwise img = prod_s exp(-mu).
// One thread per ray: neighbouring threads are neighbouring pixels, so the 8 voxel reads per sample of a wave
// fall in the same few cache lines (the kernel is L2/HBM-bound: 32 B of volume per sample, no reuse in registers).
struct VolArgs {
  const float* vol;
  int32_t nx, ny, nz;
  double x0, y0, z0, dx, dy, dz;
  float fill;
  int32_t type_ct;
};

__device__ __forceinline__ bool vol_axis(double p, double a0, double da, int n, int& i, double& t) {
  const double a1 = a0 + da * (n - 1);
  if (!(p >= a0 && p <= a1)) return false;
  double u = (p - a0) / da;
  i = (int)u;
  if (i > n - 2) i = n - 2;
  if (i < 0) i = 0;
  t = u - i;
  return true;
}

__global__ void k_project_volume(const ChainArgs a, const VolArgs v) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_total) return;          // n_total = n_rays here
  double ox, oy, oz, dx, dy, dz;
  if (a.poses == nullptr) {
    ox = a.org[3 * r]; oy = a.org[3 * r + 1]; oz = a.org[3 * r + 2];
    dx = a.dir[3 * r]; dy = a.dir[3 * r + 1]; dz = a.dir[3 * r + 2];
  } else {
    const int64_t id = a.ray_ids ? (int64_t)a.ray_ids[r] : a.ray_id0 + r;
    const int64_t hw = (int64_t)a.width * a.height;
    const int64_t proj = id / hw;
    const int pix = (int)(id - proj * hw);
    const int jj = pix / a.width, ii = pix - jj * a.width;
    const double* M = a.poses + 12 * proj;
    const double c0 = ((double)ii - a.width * 0.5) / a.focal, c1 = -((double)jj - a.height * 0.5) / a.focal;
    dx = __dadd_rn(__dadd_rn(__dmul_rn(c0, M[0]), __dmul_rn(c1, M[1])), -M[2]);
    dy = __dadd_rn(__dadd_rn(__dmul_rn(c0, M[4]), __dmul_rn(c1, M[5])), -M[6]);
    dz = __dadd_rn(__dadd_rn(__dmul_rn(c0, M[8]), __dmul_rn(c1, M[9])), -M[10]);
    ox = M[3]; oy = M[7]; oz = M[11];
  }
  const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
  const int S = a.n_samples;
  const size_t sy = (size_t)v.nz, sx = (size_t)v.ny * v.nz;
  double prod = 1.0;
  for (int s = 0; s < S; ++s) {
    const double zs = (double)a.z[s];
    const double px = ox + dx * zs, py = oy + dy * zs, pz = oz + dz * zs;
    int ix, iy, iz;
    double tx, ty, tz;
    double mu = v.fill;
    if (vol_axis(px, v.x0, v.dx, v.nx, ix, tx) && vol_axis(py, v.y0, v.dy, v.ny, iy, ty) && vol_axis(pz, v.z0, v.dz, v.nz, iz, tz)) {
      const float* b = v.vol + ix * sx + iy * sy + iz;
      const double c00 = b[0] * (1 - tz) + b[1] * tz, c01 = b[sy] * (1 - tz) + b[sy + 1] * tz;
      const double c10 = b[sx] * (1 - tz) + b[sx + 1] * tz, c11 = b[sx + sy] * (1 - tz) + b[sx + sy + 1] * tz;
      mu = (c00 * (1 - ty) + c01 * ty) * (1 - tx) + (c10 * (1 - ty) + c11 * ty) * tx;
    }
    if (v.type_ct) {
      const double dist = s + 1 < S ? (double)__fsub_rn(a.z[s + 1], a.z[s]) : (double)1e10f;
      prod *= exp(-mu * (dist * nrm));
    } else prod *= exp(-mu);
  }
  a.pixel[r] = (float)prod;
}

// sample_pdf + merge of fine_sampling (nerf/nerf_helpers.py:178-222); one thread per ray.
#define AFX_MAX_COARSE 512
#define AFX_MAX_FINE 512
// tau != nullptr: the coarse pass's per-sample optical depths [R,S] instead of its weights: the weights of render_volume_density
// (nerf_helpers.py:107-108), w_i = (1 - alpha_i + 1e-10) prod_{j<i} alpha_j with alpha = exp(-tau), are formed here, per ray, in order.
// zf_out != nullptr: also the NF new depths alone, ascending [R,NF] (hierarchical step with coarse re-use: the second pass evaluates only these);
// zout may then be null.
__global__ void k_fine_depths(const float* zc, int z_per_ray, const float* wc, const float* tau, const float* u, int64_t n_rays,
                              int S, int NF, float* zout, float* zf_out = nullptr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float* z = z_per_ray ? zc + r * S : zc;
  const int nb = S - 1;                  // bins = mid-points, cdf has nb entries
  float cdf[AFX_MAX_COARSE];
  float smp[AFX_MAX_FINE];
  if (tau) {                             // cdf[] holds weights[1 .. S-2] for a moment
    float T = expf(-tau[r * S]);         // exclusive transmittance in front of sample 1
    for (int i = 1; i < S - 1; ++i) {
      const float al = expf(-tau[r * S + i]);
      cdf[i] = __fmul_rn(__fadd_rn(__fsub_rn(1.f, al), 1e-10f), T);
      T = __fmul_rn(T, al);
    }
  } else {
    for (int i = 1; i < S - 1; ++i) cdf[i] = wc[r * S + i];      // weights[..., 1:-1]  -> S-2 values
  }
  float wsum = 0.f;
  for (int i = 1; i < S - 1; ++i) wsum = __fadd_rn(wsum, __fadd_rn(cdf[i], 1e-5f));
  cdf[0] = 0.f;
  float run = 0.f;
  for (int i = 1; i < S - 1; ++i) {
    run = __fadd_rn(run, __fadd_rn(cdf[i], 1e-5f) / wsum);
    cdf[i] = run;
  }
  for (int k = 0; k < NF; ++k) {
    const float uu = u[r * NF + k];
    int lo = 0, hi = nb;                 // searchsorted(cdf, u, right=True)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
    const int below = lo - 1 < 0 ? 0 : lo - 1;
    const int above = lo > nb - 1 ? nb - 1 : lo;
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = 0.5f * __fadd_rn(z[below + 1], z[below]);
    const float b1 = 0.5f * __fadd_rn(z[above + 1], z[above]);
    float den = __fsub_rn(c1, c0);
    if (den < 1e-5f) den = 1.f;
    const float t = __fsub_rn(uu, c0) / den;
    const float v = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    int p = k;                           // insertion sort
    while (p > 0 && smp[p - 1] > v) { smp[p] = smp[p - 1]; --p; }
    smp[p] = v;
  }
  if (zf_out)
    for (int k2 = 0; k2 < NF; ++k2) zf_out[r * NF + k2] = smp[k2];
  if (!zout) return;
  int i = 0, k = 0;
  float* o = zout + r * (S + NF);
  for (int p = 0; p < S + NF; ++p) {
    if (k >= NF || (i < S && z[i] <= smp[k])) o[p] = z[i++];
    else o[p] = smp[k++];
  }
}
#endif  // AFX_TEMPLATES_ONLY
