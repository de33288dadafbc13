// afx_internal.h — shared between the C-ABI host code and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace afx {

constexpr int TILE = 128;            // samples per workgroup tile (4 waves x 32 sample-columns)
constexpr int GROUP = 32;            // samples per wave column group; s_pad is a multiple of this

// Geometry of the bf16 chain kernels' weight ring, shared by the kernels and the host (LDS size, prepared layout).
// A step streams TPS consecutive 32-row output tiles of one layer into one LDS slot (at least 8 KiB: one LDS-DMA
// piece per wave of an 8-wave workgroup); a ring of RING slots is requested RING-1 steps ahead.  Forward-only
// kernels: 2 slots of 4 tiles.  Backward kernel: its ReLU masks fill half the LDS: 2 slots of 2 tiles (a 4-slot
// ring of one-tile steps, requested 3 steps ahead, measured 5 ms slower per 512^2x128 step: twice the barriers).
// tiles per step: 4 in the forward-only kernels and, at width 128 (a layer is 4 tiles, its slabs 32 KiB), also in the backward kernel -
// one barrier per layer; at width 256 the backward kernel's ReLU masks leave LDS room for 2-tile steps only
// PHASE 2 (the backward half of the split training step) at widths <= 128: TWO workgroups per CU (AFX_P2_OCC2, default on).  At those widths
// a tile is 8 MFMAs against an epilogue of the same length as at width 256, so a lone workgroup - whose two waves per SIMD run in lockstep
// behind the step barrier - leaves the matrix pipe idle most of the time; a second, unsynchronised workgroup fills it.  That needs <= 128
// VGPRs per wave (the backward half fits: no forward activations, no in-kernel group sums) and <= 80 KB of LDS: two-tile steps.
#ifndef AFX_P2_OCC2
#define AFX_P2_OCC2 1
#endif
constexpr bool chain_occ2(int nt, int phase) { return AFX_P2_OCC2 && phase == 2 && nt <= 4; }
constexpr int chain_tps(int nt, bool bwd, bool x3, int phase = 0) {
  return chain_occ2(nt, phase) ? (nt >= 2 ? 2 : 1) : ((nt >= 4 && !x3 && (!bwd || nt == 4)) ? 4 : (nt >= 2 ? 2 : 1));
}
constexpr int chain_ring(bool bwd) { return 2; }
constexpr uint32_t chain_slab0_bytes(int nk0) { return ((uint32_t)nk0 * 2048u + 4095u) / 4096u * 4096u; }
constexpr uint32_t chain_slot_bytes(int nt, int nk0, bool bwd, bool x3, int phase = 0) {
  const uint32_t s0 = phase == 2 ? 0u : (uint32_t)chain_tps(nt, bwd, x3, phase) * chain_slab0_bytes(nk0);      // (the backward half streams no first-layer slab)
  const uint32_t sh = (uint32_t)chain_tps(nt, bwd, x3, phase) * (uint32_t)nt * 2048u * (x3 ? 2u : 1u);
  return s0 > sh ? s0 : sh;
}

// Arguments of the fused chain kernels (by value, < 512 B).
struct ChainArgs {
  // prepared weights
  const char* stream_fwd;   // slab0, then n_hidden*NT forward slabs (layer 1..N, tile 0..NT-1)
  const char* stream_bwd;   // n_hidden*NT transposed slabs (layer N..1, tile 0..NT-1); bf16: directly behind stream_fwd
  const char* stream_lo;    // split-bf16 only: the lo parts of the forward hidden slabs, same order as the hi parts
  const float* small;       // permuted biases, output weights, bout, encoding aux
  uint32_t small_floats;    // multiple of 4
  uint32_t small_bytes_pad; // LDS bytes reserved for `small` (multiple of 1024)
  uint32_t slab0_bytes, slabh_bytes, slot_bytes;   // multiples of 4096
  uint32_t slabh_stride, slabt_bytes;              // bf16: distance between forward slabs; transposed slab size
  int32_t n_hidden, k0, nq, enc, n_freq;
  // work range
  int32_t tile0, tile1;     // global tile ids [tile0, tile1), in units of the kernel's tile (128 or 256 samples)
  int64_t n_total;          // samples: n_rays * s_pad, or n_pts
  int32_t mode;             // 0 = points, 1 = rays
  // inputs
  const float* pts;
  const float* org;
  const float* dir;
  const double* poses;
  const int32_t* ray_ids;
  int64_t ray_id0;
  int32_t width, height;
  double focal;
  int32_t n_samples, s_pad, depth_mode;
  float t_near, t_step;
  float t_far;              // AFX_DEPTH_STRATIFIED
  uint64_t jitter_seed, jitter_stream;
  const float* z;
  // outputs
  float* out;               // points mode
  int32_t apply_sigmoid;
  float* od_part;           // [R, s_pad/32]
  float* sigma;             // optional [R,S]
  float* tau;               // optional [R,S]
  // backward only
  const float* dod;         // [R] dL/d(optical depth)
  float* stash_h;           // [(N+1), rows, F]   post-activation H_0..H_N
  float* stash_dz;          // [(N+1), rows, F]   dL/dZ_0..dZ_N
  float* stash_e;           // f32 kernels / raw-coordinate inputs: [rows, 2*nq] fp32 encoded inputs.  16-bit kernels WITH an encoding: 16-bit
                            // chunk-major stash [row>>5][8 chunks][row&31][8 values] of the 64 input columns, and behind it (stash_rows*128 B)
                            // the same layout of d(enc)/d(coef)/(2 pi) when coef_cols > 0
  float* graw;              // [rows]             dL/draw
  int64_t stash_rows;
  int32_t persistent;       // 1: grid = #CUs, workgroups loop over tiles; 0: one workgroup per tile (lets the dispatcher
                            //    interleave this grid with a concurrent kernel on another stream)
  int32_t fused;            // backward kernel computes pixel, MSE gradient and dL/draw itself (train step)
  const float* target;      // [R] (fused)
  float* pixel;             // [R] out (fused)
  float inv_n;              // 1 / global ray count (fused)
  int32_t debug;            // unused
  // in-kernel small gradients (bf16 backward, rays mode, no encoding): per 32-sample group [SW[F] S0[F] S1[F] c[3] d[3] sum_g -]
  float* small_part;        // null: H_N, dZ_0, encoded inputs and dL/draw are stashed for k_small_grads_bf16 instead
  uint32_t* gmax;           // f16 mode: bit pattern of max |dL/draw| over the chunk (integer atomicMax; zeroed per chunk)
  int32_t stash8;           // f16 mode: 8-bit (bf8) stash of H_l and dZ'_l
  int32_t* gexp;            // 8-bit stash: power-of-two exponent of each 32-sample group's largest |dL/draw| [rows/32]
  uint32_t* hexp;           // 6-bit H stash: [N][rows/32] the E8M0 block scales of H_l, one byte per 64-feature tile pair
  int32_t coef_cols;        // 3*n_freq when the fourier coefficients train (the chain kernel then also stashes d(enc)/d(coef)/(2 pi)), else 0
  // hierarchical step with coarse re-use (afx_hier_train_step_mse): PHASE 1 runs before the samples' final step lengths are known (the fine depths
  // depend on this very pass): it stashes H_N as well (the output layer's gradient is contracted from the stash later) and forms no g' sums;
  // PHASE 2 then reads the finished dL/draw per sample from gpart (dod == null)
  int32_t defer_out;
  // packed, group-aligned samples (AFX_DEPTH_PACKED = 4: the occupancy-grid march's output): sample n of the padded list belongs to ray
  // group_ray[n >> 5]; its interval is [z[n], te[n]) (dead padding slots: te <= ts); optical-depth partials are indexed by group
  const float* te;
  const int32_t* group_ray;
  int32_t act;              // activation of the hidden layers: 0 ReLU (every kernel), 1 tanh, 2 sine (forward-only kernels)
  float act_w0;             // sine: the first layer's frequency factor w0 (model/CPPN.py:53-57)
  // split phases of the 8-bit-stash training kernel (rays that straddle workgroup tiles)
  char* masks;              // [tiles of the chunk][(N+1) x NT x 512 x u16]: the ReLU-mask LDS image of each tile (PHASE 1 writes, PHASE 2 reads)
  float* gpart;             // [rows] g' = dt sigma (1 - sigma) of each sample (PHASE 1 writes, PHASE 2 reads)
};

struct WgradArgs {
  const float* stash_h;
  const float* stash_dz;
  const float* stash_e;
  const float* graw;
  int64_t rows;             // rows to contract over (multiple of TILE)
  int64_t stride_rows;      // layer stride of the stash (rows of a full chunk)
  int32_t n_hidden, k0, k0pad;
  int32_t n_splits, rows_per_split;   // rows_per_split even
  float* partial;           // [(N+2), n_splits, F*F]: slot l = layer l; slot N+1 = the d(enc)/d(coef) contraction
  float* partial2;          // [(N+2), n_splits, F+4]
  int32_t debug;            // timing experiments only (bit5: default-policy instead of non-temporal stash loads)
  float* partial_s;         // bf16 path: [n_small, F*k0pad + 2F + 4] first-layer / output-layer partials
  int32_t small_groups;     // 1: the chain kernel left per-group sums where H_N's stash would be (k_small_from_groups)
  const uint32_t* gmax;     // f16 mode: the chunk's max |dL/draw| (scale of the contraction, wgrad_scale_exp)
  int32_t stash_esz;        // bytes per stash element (4 f32, 2 bf16/f16, 1 bf8)
  const int32_t* gexp;      // 8-bit stash: group exponents (block scales of the MX contraction)
  int y0;                   // k_wgrad_s8: grid row of the launch's first block row (hidden layers 0..N-1, then the encoded first layer's rows)
  const uint32_t* hexp;     // 6-bit H stash: [N][stride_rows/32] E8M0 bytes of H_l's tile pairs (B operand's block scales)
  int32_t enc16;            // 16-bit kernels with an encoding: stash_e is the 16-bit chunk-major input stash; k_wgrad_bf16 contracts layer 0 on
                            // the matrix pipe (blockIdx.y = n_hidden) and, with coef_cols > 0, d(enc)/d(coef) as well (blockIdx.y = n_hidden + 1)
  int32_t coef_cols;        // 3*n_freq when the fourier coefficients train, else 0
  // split phases: the output-layer group sums were formed with g' (without the ray's dL/d(optical depth)); k_small_from_groups applies it
  const float* dod;         // [n_rays] or null (fused kernel: the sums already carry it)
  const int32_t* group_ray; // packed samples: ray of each group (then dod[group_ray[g]]), else null (dod[(group0 + g) / gpr])
  const float* records;     // the group records when they do not lie where H_N's stash would be (hierarchical step with coarse re-use), else null
  int32_t no_sw;            // hierarchical step with coarse re-use: the group records hold the first-layer sums only (output layer: k_wout_stash8)
  const float* gfull;       // ... and dL/draw per stash row for k_wout_stash8
  int32_t gpr;              // 32-sample groups per ray (s_pad / 32)
  int64_t group0;           // global index of the chunk's first group
  int64_t n_groups_valid;   // groups that belong to a ray (the chunk's last tile may be padded with dead groups beyond them)
};

struct ReduceArgs {
  const float* partial;
  const float* partial2;
  int32_t n_hidden, k0, k0pad, n_splits;
  int32_t hidden_only;      // bf16 path: layer 0 and the output layer are reduced by k_reduce_small
  int32_t layer0_mfma;      // ... unless layer 0 came out of k_wgrad_bf16 (enc16): then k_reduce_w / k_reduce_b take it like a hidden layer
  int32_t n_small;
  const float* partial_s;
  float* grad;              // flat parameter gradient, accumulated into
  const uint32_t* gmax;     // f16 mode: hidden-layer partials carry the factor 2^-e (wgrad_scale_exp); null otherwise
  int32_t scale_shift;      // 8-bit stash: and the factor 2^scale_shift of the stashed J
  const float* w0;          // k_reduce_coef: the first layer's fp32 weights [F, k0]
  float* d_coef;            // k_reduce_coef: d loss / d fourier coefficients [coef_cols], accumulated into
  int32_t coef_cols;
};

}  // namespace afx
