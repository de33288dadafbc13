// afx_kernels_bf16.hip — bf16-MFMA kernels of the hot path for gfx950 (MI355X).
//
// Same skeleton as afx_kernels_f32.hip (register-resident activations, weights streamed through LDS
// as the MFMA A operand, previous accumulators fed back as the B operand) on
// v_mfma_f32_32x32x16_bf16.  The accumulator of output tile t' converts to the B fragments of k-steps
// 2t', 2t'+1: registers 8s..8s+7 pack pairwise into 8 bf16; element j of lane-half h is feature
// 32t' + 16s + 8(j>>2) + 4h + (j&3), and afx_prepare_weights() stores the weight slabs in that k order.
//
//   X3 = false : bf16 operands, fp32 accumulate; a wave owns 64 sample columns (2 groups of 32), so every
//                A fragment read from LDS feeds two MFMAs.
//   X3 = true  : split-bf16: x = hi + lo (both bf16), products hi*hi + hi*lo + lo*hi, fp32 accumulate:
//                ~2^-17 relative error per product (fp32-grade) for 3 MFMAs; 32 columns per wave.
//   H16 = true : the hidden layers run on v_mfma_f32_32x32x16_f16 (f16 operands, 11 significant bits instead of 8,
//                same rate): forward pixel error vs the fp32 oracle ~1e-5 instead of ~2e-4 relative L2.  The backward
//                kernel then runs the input-gradient chain NORMALISED, J_l = dZ_l / g (g = dL/draw of the sample): J does
//                not inherit g's 40-binade range, so f16 needs no loss scaling; g is stored per sample and re-applied by
//                the weight-gradient kernel (and by the in-kernel first-/output-layer sums).
//   The first layer (raw world coordinates, +-100) is always split bf16; the output layer (width -> 1) and
//   the compositing run in fp32 on the VALU.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include "afx_internal.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {       // -> v_cvt_pk_bf16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
}
// f16 forms (H16 kernels): v_cvt_pk_f16_f32 (RNE) and v_mfma_f32_32x32x16_f16
__device__ __forceinline__ unsigned pack2h(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float h_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2_t, p)[0]; }
__device__ __forceinline__ float h_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2_t, p)[1]; }
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <bool H16> __device__ __forceinline__ unsigned pack2t(float a, float b) { return H16 ? pack2h(a, b) : pack2(a, b); }
template <bool H16> __device__ __forceinline__ f32x16 mfma_t(u32x4 a, u32x4 b, f32x16 c) { return H16 ? mfma_f16(a, b, c) : mfma_bf16(a, b, c); }
template <bool H16> __device__ __forceinline__ float lo_t(unsigned p) { return H16 ? h_lo(p) : bf_lo(p); }
template <bool H16> __device__ __forceinline__ float hi_t(unsigned p) { return H16 ? h_hi(p) : bf_hi(p); }

// 8 fp32 values -> one hi fragment (and the residual lo fragment)
__device__ __forceinline__ void split_frag(const float* v, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned p = pack2(v[2 * q], v[2 * q + 1]);
    hi[q] = p;
    lo[q] = pack2(v[2 * q] - bf_lo(p), v[2 * q + 1] - bf_hi(p));
  }
}

// NW = wavefronts per workgroup.  NW = 4: one wave per SIMD with up to 512 registers (needed by the split
// mode, whose hi+lo activations fill them; plain bf16 then takes 2 column groups per wave).  NW = 8: two
// waves per SIMD with <= 256 registers and one column group each: the partner wave's MFMAs cover a wave's
// epilogue VALU work, barrier and LDS latencies.
// byte offset of the 16-byte chunk `ch` (8 stash positions) of stash row `r` inside one layer's stash
template <int F> __device__ __forceinline__ uint32_t stash_off(uint32_t r, int ch) {
  return ((((r >> 5) * (F / 8) + ch) << 5) + (r & 31)) << 4;      // < 2^32: run_backward bounds a chunk's layer plane to 4 GiB
}

// S8 (8-bit stash, f16 mode): H_l and dZ'_l = g_hat J_l are stashed as bf8 (e5m2: the top byte of the f16 pattern, same
// exponent range, so H needs no scale; dZ' is scaled by 2^AFX_S8_JSHIFT, J sits around 1e-4..1 and |g_hat| <= 1: g of a
// sample divided by the power of two of its 32-sample group's largest |g|, which travels as that group's block scale).  A lane's 16 values of a
// tile are ONE 16-byte store: stash position p8 = 16 (2t + h) + 8 s + j for feature 32t + 16s + 8(j>>2) + 4h + (j&3);
// layout [row>>5][p8>>4][row&31][16 B], so a wave store is again one contiguous 1 KiB run.  Half the bytes of the
// 16-bit stash in both directions.
#define AFX_S8_JSHIFT 10
template <int F> __device__ __forceinline__ uint32_t stash_off8(uint32_t r, int ch16) {
  return ((((r >> 5) * (F / 16) + ch16) << 5) + (r & 31)) << 4;
}
#ifndef AFX_GAPS      // (the MFMA-gap schedule of the 8-bit-stash backward kernel, described below)
#define AFX_GAPS 0
#endif
#define AFX_GAPS_ON AFX_GAPS
// H6 (6-bit H stash; -DAFX_H6=1, build.py --variant=h6 - built, tested, NOT the default: on the 512^2 x 128 step it takes 1 ms off
// k_wgrad_s8 and puts 1.6 ms on the chain kernel): the hidden
// activations H_l (l < N), the B operand of the weight-gradient contraction, are stashed as bf6 (e3m2: the 2 mantissa bits of bf8, 3 exponent
// bits) with ONE power-of-two scale per (32-sample group, 64-feature tile pair) - the E8M0 block scale the matrix instruction takes for B.
// A wave owns exactly that block (32 sample columns x the 64 features of two consecutive tiles), so the scale is a wave-wide max: 15
// v_pk_max_u16 on the packed non-negative f16 patterns, 6 DPP steps, one v_readlane; the largest value lands in [8, 16) of e3m2's [1/16, 28].
// v_cvt_scalef32_pk32_bf6_f16 packs a lane's 32 values of the pair into 6 dwords, element i in bit field 6 i (tools/micro/fp6_probe.hip):
// dwords 0-2 = the lane's 16 values of tile t = 12-byte unit 2t + h, dwords 3-5 = unit 2(t+1) + h; positions inside a unit as in the bf8
// layout (fperm8).  Layout [row>>5][unit][row&31][12 B]: a wave's dwordx3 store is one contiguous 768-byte run.  3/4 of the bf8 bytes; the
// plane stride stays rows x F.  The four (F = 256) E8M0 bytes of a group and layer travel as one dword, a.hexp[l][group].
#ifndef AFX_H6
#define AFX_H6 0
#endif
#define AFX_H6_ON (AFX_H6 && !AFX_GAPS_ON)
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32_t __attribute__((ext_vector_type(32)));
template <int F> __device__ __forceinline__ uint32_t stash_off6(uint32_t r, int unit) {
  return ((((r >> 5) * (F / 16) + unit) << 5) + (r & 31)) * 12u;
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {      // (builtin, not asm: hipcc pads asm with s_nop and cannot interleave it)
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
// max over the wave of a non-negative value; the result is wave-uniform (an SGPR)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));      // row_shr:1
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));      // row_shr:2
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));      // row_shr:4
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));      // row_shr:8: lane 15 of a row holds the row's max
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// 8 packed f16 pairs (16 values, stash order) -> 16 bf8 bytes; value / scale is what is stored
__device__ __forceinline__ u32x4 to_bf8x16(u32x4 a, u32x4 b, float scale) {
  u32x4 r;
  const unsigned src[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s16x2 v = {0, 0};
    v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(v, __builtin_bit_cast(f16x2_t, src[2 * i]), scale, false);
    v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(v, __builtin_bit_cast(f16x2_t, src[2 * i + 1]), scale, true);
    r[i] = __builtin_bit_cast(unsigned, v);
  }
  return r;
}
// one f16 value -> bf8 with stochastic rounding into byte `sel` of `acc` (the builtin wants the byte index as a constant)
__device__ __forceinline__ unsigned sr_bf8(unsigned acc, _Float16 v, unsigned rnd, int sel) {
  switch (sel) {
    case 0: return __builtin_amdgcn_cvt_scalef32_sr_bf8_f16(acc, v, rnd, 1.0f, 0);
    case 1: return __builtin_amdgcn_cvt_scalef32_sr_bf8_f16(acc, v, rnd, 1.0f, 1);
    case 2: return __builtin_amdgcn_cvt_scalef32_sr_bf8_f16(acc, v, rnd, 1.0f, 2);
    default: return __builtin_amdgcn_cvt_scalef32_sr_bf8_f16(acc, v, rnd, 1.0f, 3);
  }
}
// 4 packed f16 pairs (8 values) -> 8 bf8 bytes
__device__ __forceinline__ u32x2 to_bf8x8(unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
  s16x2 lo = {0, 0}, hi = {0, 0};
  lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(lo, __builtin_bit_cast(f16x2_t, p0), 1.0f, false);
  lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(lo, __builtin_bit_cast(f16x2_t, p1), 1.0f, true);
  hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(hi, __builtin_bit_cast(f16x2_t, p2), 1.0f, false);
  hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(hi, __builtin_bit_cast(f16x2_t, p3), 1.0f, true);
  return (u32x2){__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
// one packed f16 pair -> two bf8 bytes in the low (HI = false) or high half of `acc` (the builtin wants HI as a constant)
template <bool HI> __device__ __forceinline__ unsigned bf8_pair(unsigned acc, unsigned pair) {
  if (HI) asm("v_cvt_scalef32_pk_bf8_f16 %0, %1, 1.0 op_sel:[0,0,1]" : "+v"(acc) : "v"(pair));
  else asm("v_cvt_scalef32_pk_bf8_f16 %0, %1, 1.0" : "+v"(acc) : "v"(pair));
  return acc;
}
// one v_pk_mul_f16 on dwords.  (asm: hipcc 7.2 miscompiles f16x2 arithmetic on bit-cast elements of a u32x4 - it folds the
// element index away, see k_wgrad_bf16 - and the gap schedule wants exactly one instruction per slot anyway)
__device__ __forceinline__ unsigned pk_mul_f16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// compile-time unrolled loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int... I, class Fn> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class Fn> __device__ __forceinline__ void static_for(Fn&& f) { static_for_impl(std::make_integer_sequence<int, N_>{}, f); }
// AFX_GAPS=1 (build.py --variant=gaps): the MFMA-gap schedule of the 8-bit-stash backward kernel, kept for A/B.  It measured
// 3 ms SLOWER per step than the plain order, with real and with all-zero operands, so the default is off.
// stash position -> feature for the 8-bit layout (a permutation inside each block of 32)
__device__ __forceinline__ int fperm8(int p) {
  const int hh = (p >> 4) & 1, s2 = (p >> 3) & 1, j = p & 7;
  return (p & ~31) | (s2 << 4) | ((j >> 2) << 3) | (hh << 2) | (j & 3);
}

// 16-byte stash store, non-temporal: the stash is written once and read once by another kernel, so it must not
// displace the weight slabs every workgroup re-streams from L2.  A/B in one process on the 512^2x128 step,
// k_chain<bwd>: plain 116 ms, nt 99 ms, sc1 (write-through) 121 ms.
#ifdef AFX_STASH_WINDOW      // measurement build only (build.py --variant=window): every stash store lands in one 1 MiB window that stays in L2,
__device__ char g_stash_window[1 << 20];      // i.e. the same instruction stream without the HBM write stream (results are garbage)
__device__ __forceinline__ void stash_store(char* p, u32x4 v) { *(u32x4*)(g_stash_window + ((uintptr_t)p & 0xFFFF0u)) = v; }
#else
__device__ __forceinline__ void stash_store(char* p, u32x4 v) { __builtin_nontemporal_store(v, (u32x4*)p); }
#endif

// Packed-bf16 epilogue helpers.  The bf16 sign bit is the int16 sign bit, so ReLU of two packed values is
// one v_pk_max_i16, [h != 0] per half one v_pk_min_u16, and a per-half 0xffff/0 mask from bit q of each
// half is v_pk_lshlrev_b16 + v_pk_ashrrev_i16: half the VALU work of the same operations on fp32 values.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu2(unsigned p) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), (s16x2){0, 0}));
}
__device__ __forceinline__ unsigned nz2(unsigned p, unsigned one2 /* 0x00010001 */) {
  unsigned r;      // asm: hipcc otherwise rewrites min(max(p,0),1) as compare + select + perm, 5 instructions per pair
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(p), "v"(one2));
  return r;
}
__device__ __forceinline__ unsigned halfmask(unsigned b, int q) {
  s16x2 v = __builtin_bit_cast(s16x2, b);
  v = v << (short)(15 - q);
  v = v >> (short)15;
  return __builtin_bit_cast(unsigned, v);
}
// ReLU masks of one 32-row tile: 16 bits per lane.  Bit q = [element 2q > 0], bit 8+q = [element 2q+1 > 0]
// (elements as rounded to bf16).  mask_expand() moves the two bytes to the two halves of a dword for halfmask().
__device__ __forceinline__ unsigned mask_expand(unsigned b16) { return (b16 & 0xffu) | ((b16 & 0xff00u) << 8); }

#ifdef AFX_STAMP      // diagnostic build only: per-phase cycle totals of workgroup 0's waves (s_memtime), read back by afx_destroy
#ifndef AFX_STAMP_F
#define AFX_STAMP_F 256      // the layer width whose backward-family kernels are stamped (-DAFX_STAMP_F=128 for the reference's default model)
#endif
__device__ unsigned long long g_stamps[3][8][10];      // [PHASE][wave][phase of the tile]
#define STAMP(i) do { if (BWD && NW == 8 && F == AFX_STAMP_F) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// A-fragment reads with hand-counted waits.  hipcc waits with lgkmcnt(0) before every MFMA group, i.e. also for the
// fragments it has just requested for LATER MFMAs, which exposes an LDS round trip every few MFMAs.  LDS operations
// return in order, so `lgkmcnt(k)` after k younger reads retires exactly the fragment the next MFMA needs; the "+v"
// operand ties the wait to the register so the MFMA cannot move above it.  (Scalar loads share the counter and return
// out of order, but a read that is still outstanding keeps all k younger reads outstanding with it, so the count
// can only over-wait.)
__device__ __forceinline__ void lds_read_frag(u32x4& dst, uint32_t lds_addr, int imm) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(imm) : "memory");
}
// -DAFX_SAFE_WAITS (the race-detector build, libafx_safe.so): every counted wait of the chain kernels becomes a full
// one - lgkmcnt(0) before each MFMA, vmcnt(0) + barrier at every step.  tests/ assert the production library is
// bit-identical to it at full size: an under-counted wait cannot hide behind run-to-run determinism.
template <int K> __device__ __forceinline__ void lds_wait_frag(u32x4& r) {
#ifdef AFX_SAFE_WAITS
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r) :: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(K) : "memory");
#endif
}
#ifndef AFX_PF_FWD
#define AFX_PF_FWD 4
#define AFX_PF_BWD 4
#endif
#ifndef AFX_PP_FWD
#define AFX_PP_FWD true
#endif
// SG ("small gradients in the kernel", backward, rays mode, no encoding): the first layer's and the output layer's
// weight gradients contract over samples too, but have only 3 / 1 columns, and their operands dZ_0 and H_N were a ninth
// of the stash traffic plus a pass of their own (k_small_grads_bf16).  A wave's 32 samples are one ray's 32-sample
// group, on which the inputs are affine in the ray parameter, x_n = c + (t_n - t_0) d, so per group three 256-vectors
// suffice:  SW = sum_n g_n H_N[n],  S0 = sum_n dZ_0[n],  S1 = sum_n (t_n - t_0) dZ_0[n]
// (dW_out += SW, db_0 += S0, dW_0 += S0 c^T + S1 d^T; k_small_from_groups).  The sums run over the LANE dimension
// of the fragments; an MFMA against a 0/1 selection matrix transposes a fragment exactly (bf16 x 1.0, fp32
// accumulate) into the C layout - lane = feature position, 16 registers = samples - where the sum is 16 VALU FMAs.
// PHASE (8-bit-stash training kernel, rays mode): 0 = the fused kernel (forward, compositing, backward in one pass: needs every ray inside
// one workgroup tile).  1 / 2 = the same work as TWO launches for rays that straddle tiles (samples per ray not a divisor of the
// tile: the reference's own 300, the 128 + 64 of the hierarchical pass): PHASE 1 is the forward half - it stashes H_l, leaves the
// ReLU masks of its tile (the LDS image, as it is) and g' = dt sigma (1 - sigma) per sample in HBM, the optical-depth partials of the
// 32-sample groups and the output-layer group sums weighted by g' (dL/d(optical depth) of the ray is not known yet; it is a
// factor common to a group, applied by k_small_from_groups); after the per-ray reduction (k_finish_mse: pixel, dL/d(optical depth))
// PHASE 2, the backward half, loads the masks back into LDS by LDS-DMA and runs the input-gradient chain with g = dod[ray] g'.
// Nothing is computed twice (the two-launch path it replaces rendered the forward, then recomputed it inside the backward kernel).
// ACTV = 1: the forward-only kernels for tanh / sine models (a.act): the activation is applied to the accumulator in the epilogue instead of
// the packed-ReLU steps.  Separate instantiations, so that the ReLU kernels stay exactly the code they were.
template <int F, bool X3, bool ENC, bool BWD, int NW, bool SG = false, bool H16 = false, bool S8 = false, int PHASE = 0, int ACTV = 0>
__global__ void __launch_bounds__(64 * NW, chain_occ2(F / 32, PHASE) ? 4 : NW / 4) k_chain_bf16(const ChainArgs a) {
  static_assert(PHASE == 0 || (S8 && PHASE <= 2), "split phases: the 8-bit-stash kernel");
  static_assert(ACTV == 0 || (!BWD && !ENC), "tanh / sine: forward-only kernels without an input encoding");
  constexpr bool P1 = PHASE == 1, P2 = PHASE == 2;
  static_assert(!S8 || (SG && H16), "8-bit stash: f16 backward kernel with in-kernel small gradients");
  static_assert(!SG || (BWD && !X3 && (!ENC || S8)), "in-kernel small gradients: plain backward kernel; with an encoding only the 8-bit-stash kernel");
  static_assert(!(H16 && (X3 || NW != 8)), "f16 hidden layers: 8-wave kernels only");
  static_assert(!(X3 && BWD), "the backward chain runs in plain bf16");
  static_assert(!(X3 && NW != 4), "the split mode needs 512 registers per wave");
  constexpr int NT = F / 32;
  constexpr int NCG = (X3 || NW == 8) ? 1 : 2;   // 32-column groups per wave
  constexpr int NTH = 64 * NW;
  constexpr int NK0 = ENC ? 4 : 1;         // 16-wide k-steps of the first layer
  constexpr int MW = (NT + 1) / 2;
  constexpr int TS = NW * 32 * NCG;        // samples per workgroup tile
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: scalar branches, no exec masking
  const int N = a.n_hidden;
  if (a.tile0 + (int)blockIdx.x >= a.tile1) return;

  // A step covers TPS consecutive 32-row output tiles of one layer: one barrier and one LDS-DMA batch per step.
#ifdef AFX_STAMP
  uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = __builtin_amdgcn_s_memtime();
#endif
  constexpr int TPS = chain_tps(NT, BWD, X3, PHASE);
  constexpr int RING = chain_ring(BWD);               // LDS slots of the weight ring
  constexpr int PD = RING - 1;                        // a step's slabs are requested PD steps ahead
  static_assert(!SG || PD == 1, "SG marks its store-less steps for PD = 1 only");
  constexpr int SPL = NT / TPS;                       // steps per layer
  static_assert(NT % TPS == 0, "");
  // slab sizes are fixed by the template parameters (the host lays the prepared buffer out identically)
  constexpr uint32_t SLAB0 = chain_slab0_bytes(NK0);                  // first-layer slab: NK0 x (hi,lo) KiB, zero-padded
  constexpr uint32_t SLABT = NT * 2048u;                              // one hidden slab part (hi or lo)
  constexpr uint32_t STEP0 = TPS * SLAB0, STEPH = TPS * SLABT;        // bytes one step streams (X3: STEPH hi + STEPH lo)
  constexpr uint32_t SLOT = chain_slot_bytes(NT, NK0, BWD, X3, PHASE);
  constexpr int PIECES0 = STEP0 / (NW * 1024u), PIECESH = STEPH / (NW * 1024u);      // LDS-DMA instructions per wave and step
  constexpr int SPS = (S8 ? 1 : 2) * NCG * TPS;                       // stash stores per wave and step (backward kernel)

  char* slot0 = lds + a.small_bytes_pad;
  // The prepared buffer holds [first-layer slabs | forward hidden slabs (hi) | transposed slabs] as ONE contiguous
  // stream in the order a tile consumes it (the lo parts of the split mode are a second stream): the source of the
  // next request is a running scalar pointer, and every request is a straight-line run of 1 KiB LDS-DMA pieces.
  const char* wnext = P2 ? a.stream_bwd : a.stream_fwd;      // (the transposed slabs are the tail of the one contiguous stream)
  const char* lnext = a.stream_lo;
  const uint32_t voff = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
  auto dma_run = [&](const char* src, char* dst, auto bytes_c) {
    constexpr uint32_t BYTES = decltype(bytes_c)::value;
    static_assert(BYTES % (NW * 1024u) == 0, "a step is a whole number of pieces per wave");
#pragma unroll
    for (uint32_t off = 0; off < BYTES; off += NW * 1024u)
      __builtin_amdgcn_global_load_lds(GPTR(src + off + voff), LPTR(dst + off + wave * 1024), 16, 0, 0);
  };
  // Request cursor: runs PD steps ahead of the compute, across tile boundaries, until every step of this
  // workgroup's tiles has been requested.
  const int steps_per_tile = P2 ? SPL * N : SPL * (N + 1) + ((BWD && !P1) ? SPL * N : 0);
  int to_issue = ((a.tile1 - a.tile0 - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * steps_per_tile;
  int cpos = 0;                           // position of the next requested step inside its tile
  uint32_t wslot = 0, rslot = 0;          // ring slot of the next request / of the next step to compute
  auto request = [&]() {
    char* dst = slot0 + wslot * SLOT;
    if (!P2 && cpos < SPL) {
      dma_run(wnext, dst, std::integral_constant<uint32_t, STEP0>{});
      wnext += STEP0;
    } else {
      dma_run(wnext, dst, std::integral_constant<uint32_t, STEPH>{});
      wnext += STEPH;
      if constexpr (X3) {                 // forward-only: every other step is a forward hidden step
        dma_run(lnext, dst + STEPH, std::integral_constant<uint32_t, STEPH>{});
        lnext += STEPH;
      }
    }
    if (++cpos == steps_per_tile) { cpos = 0; wnext = P2 ? a.stream_bwd : a.stream_fwd; lnext = a.stream_lo; }
    wslot = wslot + 1 == RING ? 0 : wslot + 1;
    --to_issue;
  };
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (to_issue > 0) request();

  float* sm = (float*)lds;
  for (uint32_t i = tid * 4; i < a.small_floats; i += NTH * 4) *(f32x4*)(sm + i) = *(const f32x4*)(a.small + i);
  unsigned short* mk16 = (unsigned short*)(slot0 + RING * (size_t)SLOT);   // ReLU masks [((l*NT + t)*NCG + cg)*NTH + tid]
  const float* bias_perm = sm;
  const float* wout_perm = sm + (N + 1) * F;
  const float* aux = sm + (N + 2) * F + 4;
  // the workgroup's largest |dL/draw| (f16 backward kernels), merged into a.gmax once, at the end: a word of the 256 spare bytes behind the
  // mask image (backward kernels only; the per-group optical depths use the first 32) - NOT a static __shared__ variable: the kernels ask for the
  // full 160 KiB as dynamic LDS, and static bytes on top of that make hipFuncSetAttribute refuse
  uint32_t* const wg_gmax_p = (uint32_t*)(slot0 + RING * (size_t)SLOT + (size_t)(N + 1) * MW * NCG * NTH * 4 + 128);
  if constexpr (BWD) { if (tid == 0) *wg_gmax_p = 0; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first PD steps' slabs; every later step counts (below)
  __syncthreads();

  unsigned one2 = 0x00010001u;
  asm volatile("" : "+v"(one2));       // keep the constant in a register (VOP3P takes no literal on gfx9)
  // Step protocol: wait for THIS step's slabs (requested PD steps ago), barrier (which also retires every reader
  // of the slot the next request overwrites), request the slabs of the step PD ahead, compute.  vmcnt counts in
  // issue order: behind this step's request the wave has issued the (PD-1) later requests and, in the backward
  // kernel, at least SPS stash stores in each of the PD steps since, so vmcnt(WAITN) retires the request while
  // the younger requests and stores stay in flight (a vmcnt(0) here would cost a store round trip per step).
  // Raw s_barrier: __syncthreads() would re-insert vmcnt(0).
  constexpr int WAITN = (BWD ? PD * SPS : 0) + (PD - 1) * (PIECES0 < PIECESH ? PIECES0 : PIECESH);
  // `counted` = false: the previous step issued no stash stores (SG: the last forward layer's H_N is not stashed)
  // GAPS (8-bit stash kernel): stash stores move between tiles, so the wave counts them; a step that issued fewer than SPS
  // behind its request is followed by a full wait (wave-uniform scalar counter).
  constexpr bool GAPS = S8 && AFX_GAPS && PHASE == 0;
  constexpr bool H6 = S8 && AFX_H6_ON;      // bf6 H stash (the gap-schedule A/B build keeps the bf8 one)
  static_assert(!H6 || (NCG == 1 && NT % 2 == 0 && TPS % 2 == 0), "6-bit H stash: a wave owns one 32-sample group and whole tile pairs per step");
  constexpr int IPG = 16 / (2 * NT);     // work items per MFMA gap (GAPS)
  int nstores = 0;
  // GAPS: a hidden step's request is issued piece by piece in the first MFMA gaps of the step's first tile (defer = true):
  // issued in one burst behind the barrier, the 8 waves' pieces queue in the vector-memory path and every wave sits in that
  // queue before its first MFMA (12-20 % of the waves' cycles in the stamp build).  All pieces precede the step's stores.
  bool req_pending = false;
  auto request_piece = [&](int i) {
    char* dst = slot0 + wslot * SLOT;
    const uint32_t off = (uint32_t)i * (NW * 1024u);
    if (cpos < SPL ? i < PIECES0 : i < PIECESH)
      __builtin_amdgcn_global_load_lds(GPTR(wnext + off + voff), LPTR(dst + off + wave * 1024), 16, 0, 0);
  };
  auto request_end = [&]() {
    wnext += cpos < SPL ? STEP0 : STEPH;
    if (++cpos == steps_per_tile) { cpos = 0; wnext = a.stream_fwd; }
    wslot = wslot + 1 == RING ? 0 : wslot + 1;
    --to_issue;
    nstores = 0;
    req_pending = false;
  };
  static_assert(!GAPS || (PIECES0 <= PIECESH && PIECESH <= 2 * NT && (PIECESH - 1) * IPG < 7), "GAPS: the request's pieces precede the first stash store");
  auto step_begin = [&](bool counted = true, bool defer = false) -> const char* {
    if constexpr (GAPS) counted = nstores >= SPS;
#ifdef AFX_SAFE_WAITS
    counted = false;
#endif
    if (to_issue > 0 && counted) {
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");
      STAMP(0);
      asm volatile("s_barrier" ::: "memory");
      STAMP(1);
    } else {          // also the last PD steps of the workgroup: nothing younger to count on
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      STAMP(1);
    }
    if (to_issue > 0) {
      if (GAPS && defer) req_pending = true;
      else { request(); nstores = 0; }
    }
    STAMP(2);
    const char* cur = slot0 + rslot * SLOT;
    rslot = rslot + 1 == RING ? 0 : rslot + 1;
    return cur;
  };

  constexpr uint32_t MASKB = 2u * NT * NCG * NTH;      // mask bytes per layer (the LDS image, [((l*NT + t)*NCG + cg)*NTH + tid] u16)
  uint32_t wave_gmax = 0;      // (lane 0) the largest |dL/draw| this wave has merged into wg_gmax so far
  for (int tile = a.tile0 + blockIdx.x; tile < a.tile1; tile += gridDim.x) {
    STAMP(7);
    if constexpr (P2) {
      // masks of this tile: HBM -> LDS by LDS-DMA (1 KiB per wave instruction), behind a barrier that retires every reader of the previous tile's;
      // issued FIRST, so that the ray / dL/draw loads of the tile prologue overlap its latency (waited for in front of the gradient seed)
      asm volatile("s_barrier" ::: "memory");
      const char* mg = a.masks + (size_t)(tile - a.tile0) * ((size_t)(N + 1) * MASKB);
      for (uint32_t off = (uint32_t)wave * 1024u; off < (uint32_t)(N + 1) * MASKB; off += NW * 1024u)
        __builtin_amdgcn_global_load_lds(GPTR(mg + off + lane * 16), LPTR((char*)mk16 + off), 16, 0, 0);
    }
    int32_t n[NCG];
    uint32_t m[NCG], so[NCG];           // sample index, stash row, per-lane stash byte offset (chunk 0)
    uint32_t so6 = 0, hx = 0;           // H6: byte offset of unit h in the 6-bit layout; the layer's E8M0 bytes so far
    Sample sp[NCG];
    u32x4 ehi[NCG][NK0], elo[NCG][NK0];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      n[cg] = tile * TS + wave * (32 * NCG) + cg * 32 + col;
      m[cg] = (uint32_t)(tile - a.tile0) * TS + wave * (32 * NCG) + cg * 32 + col;
      so[cg] = S8 ? stash_off8<F>(m[cg], hh) : stash_off<F>(m[cg], hh);
      if constexpr (H6) so6 = stash_off6<F>(m[cg], hh);
      sp[cg] = make_sample(a, n[cg]);
      // first-layer B fragments: element j of k-step q is encoded input k = 16q + 8*(lane>>5) + j
#pragma unroll
      for (int q = 0; q < (P2 ? 0 : NK0); ++q) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (ENC) e[j] = enc_value(16 * q + 8 * hh + j, sp[cg].px, sp[cg].py, sp[cg].pz, aux, a.enc, a.n_freq, a.k0);
          else e[j] = (hh == 0 && j < 3) ? (j == 0 ? sp[cg].px : (j == 1 ? sp[cg].py : sp[cg].pz)) : 0.f;
        }
        split_frag(e, ehi[cg][q], elo[cg][q]);
        if constexpr (BWD && ENC && S8) {
          // 8-bit stash of the encoded inputs, layout of the activation stash ([row>>5][4 chunks of 16 columns][row&31][16 B], natural
          // column order): this lane's 8 values are columns 16q + 8hh .. +7 = bytes 8hh.. of chunk q.  B operand of the first layer's
          // weight gradient in k_wgrad_s8 (inputs rounded to bf8 like the hidden activations; averaged over all samples like them)
          char* ep = (char*)a.stash_e + ((((size_t)(m[cg] >> 5) * 4 + q) << 5) + (m[cg] & 31)) * 16 + 8 * hh;
          __builtin_nontemporal_store(to_bf8x8(pack2h(e[0], e[1]), pack2h(e[2], e[3]), pack2h(e[4], e[5]), pack2h(e[6], e[7])), (u32x2*)ep);
          if (a.coef_cols > 0) {
            float de[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) de[j] = enc_dcoef(16 * q + 8 * hh + j, sp[cg].px, sp[cg].py, sp[cg].pz, aux, a.n_freq);
            __builtin_nontemporal_store(to_bf8x8(pack2h(de[0], de[1]), pack2h(de[2], de[3]), pack2h(de[4], de[5]), pack2h(de[6], de[7])),
                                        (u32x2*)(ep + (size_t)a.stash_rows * 64));
          }
        }
        if (BWD && !SG) {
          if constexpr (ENC) {
            // 16-bit chunk-major stash of the encoded inputs (the B operand of the first layer's weight gradient in k_wgrad_bf16):
            // this lane's 8 values are input columns 16q + 8hh .. +7 of its sample = chunk 2q + hh of the row, natural order
            char* ep = (char*)a.stash_e + ((((size_t)(m[cg] >> 5) * 8 + (2 * q + hh)) << 5) + (m[cg] & 31)) * 16;
            stash_store(ep, (u32x4){pack2t<H16>(e[0], e[1]), pack2t<H16>(e[2], e[3]), pack2t<H16>(e[4], e[5]), pack2t<H16>(e[6], e[7])});
            if (a.coef_cols > 0) {       // trainable fourier coefficients: the same for d(enc)/d(coef)/(2 pi)
              float de[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) de[j] = enc_dcoef(16 * q + 8 * hh + j, sp[cg].px, sp[cg].py, sp[cg].pz, aux, a.n_freq);
              stash_store(ep + (size_t)a.stash_rows * 128, (u32x4){pack2t<H16>(de[0], de[1]), pack2t<H16>(de[2], de[3]), pack2t<H16>(de[4], de[5]), pack2t<H16>(de[6], de[7])});
            }
          } else {
            float* ep = a.stash_e + (size_t)m[cg] * (16 * NK0) + 16 * q + 8 * hh;
            *(f32x4*)ep = (f32x4){e[0], e[1], e[2], e[3]};
            *(f32x4*)(ep + 4) = (f32x4){e[4], e[5], e[6], e[7]};
          }
        }
      }
    }

    const char* stepbase = nullptr;      // LDS slot of the current step
    u32x4 hf[NCG][NT][2];
    u32x4 hl[X3 ? NCG : 1][X3 ? NT : 1][2];
    float dot[NCG];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) dot[cg] = 0.f;

    auto bias_init = [&](int l, int t) -> f32x16 {
      const f32x4* bp = (const f32x4*)(bias_perm + ((l * 2 + hh) * NT + t) * 16);
      const f32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
      return (f32x16){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3],
                      b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
    };
    // epilogue of one output tile: ReLU, (mask), next fragments, (stash), output-layer dot product
    // (bl, bt >= 0: once the accumulator has been consumed it is reloaded with the bias of the tile this wave computes
    // next, so that the LDS latency of the reload hides behind the rest of the epilogue instead of in front of the MFMAs)
    auto epilogue = [&](int l, int t, f32x16& acc, int cg, u32x4* nf, u32x4* nl, int bl = -1, int bt = -1) {
      // tanh / sine models (forward-only kernels): the activation is applied to the accumulator, the ReLU steps below are skipped
      constexpr bool relu = ACTV == 0;
      if constexpr (!relu) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = act_value(acc[j], a.act, l == 0 ? a.act_w0 : 1.f);
      }
      if (l == N) {       // the output layer (width -> 1) reads the fp32 activations
        asm volatile("" ::: "memory");      // keeps hipcc from hoisting the w_out reads (and their lgkmcnt(0)) into every layer's epilogue
        const f32x4* wp = (const f32x4*)(wout_perm + (hh * NT + t) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w4 = wp[q];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            dot[cg] = fmaf(relu ? __int_as_float(max(__float_as_int(acc[4 * q + e]), 0)) : acc[4 * q + e], w4[e], dot[cg]);
        }
      }
      if constexpr (X3) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = relu ? __int_as_float(max(__float_as_int(acc[j]), 0)) : acc[j];
        split_frag(v, nf[0], nl[0]);
        split_frag(v + 8, nf[1], nl[1]);
      } else {
        // round to bf16 first, ReLU on the packed pairs (rounding is monotone and keeps the sign: same result)
        unsigned p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          p[q] = pack2t<H16>(acc[2 * q], acc[2 * q + 1]);
          if (relu) p[q] = relu2(p[q]);
        }
        if (bt >= 0) acc = bias_init(bl, bt);
        if (BWD && !(GAPS && (l >= 1 || t == NT - 1))) {
          unsigned bits = nz2(p[0], one2);
#pragma unroll
          for (int q = 1; q < 8; ++q) bits |= nz2(p[q], one2) << q;
          mk16[((l * NT + t) * NCG + cg) * NTH + tid] = (unsigned short)(bits | (bits >> 8));
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          nf[s] = (u32x4){p[4 * s], p[4 * s + 1], p[4 * s + 2], p[4 * s + 3]};
          if (S8) continue;
          if (BWD && !(SG && l == N)) {
            // stash position of feature 32t+16s+8(j>>2)+4h+(j&3) is p = 32t+16s+8h+j (bits 2,3 swapped; the
            // weight-gradient kernels undo it with fperm).  Layout [row>>5][p>>3][row&31][8 bf16]: the 64 lanes
            // of this store write one contiguous 1 KiB run (32 samples x 16 B for h = 0, then for h = 1).
            stash_store((char*)a.stash_h + (size_t)l * a.stash_rows * (F * 2) + (so[cg] + (uint32_t)(4 * t + 2 * s) * 512u), nf[s]);
          }
        }
        if constexpr (H6) {
          if (l != N) {
            if (t & 1) {      // tiles t-1 and t: one scale, one conversion, two 12-byte stores per lane (as many stores per step as the bf8 stash: vmcnt protocol unchanged)
              const u32x4* pr = nf - 2;
              unsigned mx = pk_max_u16(pk_max_u16(pk_max_u16(pr[0][0], pr[0][1]), pk_max_u16(pr[0][2], pr[0][3])),
                                       pk_max_u16(pk_max_u16(pr[1][0], pr[1][1]), pk_max_u16(pr[1][2], pr[1][3])));
              mx = pk_max_u16(mx, pk_max_u16(pk_max_u16(pk_max_u16(pr[2][0], pr[2][1]), pk_max_u16(pr[2][2], pr[2][3])),
                                             pk_max_u16(pk_max_u16(pr[3][0], pr[3][1]), pk_max_u16(pr[3][2], pr[3][3]))));
#ifdef AFX_H6_CONST      // measurement build only (build.py --variant=h6c): no max, scale 1 - the cost of the scale computation by difference
              const unsigned e8 = 127u + (mx & 0u);
#else
              const unsigned e8 = 109u + (wave_max_u32(max(mx & 0xffffu, mx >> 16)) >> 10);      // E8M0 of 2^(exponent of the max - 3)
#endif
              hx |= e8 << (8 * (t >> 1));
              const u32x16 v16 = {pr[0][0], pr[0][1], pr[0][2], pr[0][3], pr[1][0], pr[1][1], pr[1][2], pr[1][3],
                                  pr[2][0], pr[2][1], pr[2][2], pr[2][3], pr[3][0], pr[3][1], pr[3][2], pr[3][3]};
              const u32x6 r6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(__builtin_bit_cast(f16x32_t, v16), __uint_as_float(e8 << 23));
              char* hp = (char*)a.stash_h + (size_t)l * a.stash_rows * F + (so6 + (uint32_t)(2 * (t - 1)) * 384u);
              __builtin_nontemporal_store((u32x3){r6[0], r6[1], r6[2]}, (u32x3*)hp);
              __builtin_nontemporal_store((u32x3){r6[3], r6[4], r6[5]}, (u32x3*)(hp + 768));
              if (t == NT - 1) {
                if (lane == 0) a.hexp[(size_t)l * (a.stash_rows >> 5) + (m[cg] >> 5)] = hx;
                hx = 0;
              }
            }
          } else if (P1 && a.defer_out) {     // H_N for k_wout_stash8 (hierarchical step): bf8, as before
            stash_store((char*)a.stash_h + (size_t)l * a.stash_rows * F + (so[cg] + (uint32_t)(2 * t) * 512u), to_bf8x16(nf[0], nf[1], 1.0f));
          }
        } else if constexpr (S8) {
          if ((l != N || (P1 && a.defer_out)) && !(GAPS && (l >= 1 || t == NT - 1))) {     // one 16-byte store per tile: chunk 2t + h of the 8-bit layout
            ++nstores;
            stash_store((char*)a.stash_h + (size_t)l * a.stash_rows * F + (so[cg] + (uint32_t)(2 * t) * 512u), to_bf8x16(nf[0], nf[1], 1.0f));
          }
        }
        // the reloaded bias is waited for HERE (it has long landed), not by an lgkmcnt(0) behind the next tile's first
        // fragment reads, which would expose their round trip at every tile start
        if (bt >= 0) asm volatile("" :: "v"(acc[15]));
      }
    };

    // Backward kernel: acc[cg] += W_tile . B with a rolling PF-deep fragment prefetch and hand-counted LDS waits.
    // `gap(u)` (u a std::integral_constant): independent VALU / store work issued between MFMA u and MFMA u + 1.  The MFMAs of
    // a tile form one dependent chain (same accumulator), so the wave's issue slots between them are free; the partner wave
    // of the SIMD runs the same phase at the same time (the step barrier keeps the pair in lockstep), so work that sits in
    // front of or behind the loop leaves the matrix pipe idle for both.
    auto no_gap = [](auto) {};
    auto rolling_mma_impl = [&](auto pf_c, const u32x4* sl, u32x4 (*bh)[NT][2], f32x16* acc, auto&& gap) {
      constexpr int PF = decltype(pf_c)::value;
      static_assert(PF >= 1 && PF <= 5 && PF <= 2 * NT, "the tail waits below cover PF <= 5");
      const uint32_t la = (uint32_t)(uintptr_t)LPTR(sl) + (uint32_t)lane * 16u;
      u32x4 ar[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) lds_read_frag(ar[i], la, i * 1024);
      static_for<2 * NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        // younger reads outstanding behind fragment u: min(PF - 1, 2 NT - 1 - u)
        if constexpr (2 * NT - 1 - u >= PF - 1) lds_wait_frag<PF - 1>(ar[u % PF]);
        else if constexpr (2 * NT - 1 - u == 3) lds_wait_frag<3>(ar[u % PF]);
        else if constexpr (2 * NT - 1 - u == 2) lds_wait_frag<2>(ar[u % PF]);
        else if constexpr (2 * NT - 1 - u == 1) lds_wait_frag<1>(ar[u % PF]);
        else lds_wait_frag<0>(ar[u % PF]);
        const u32x4 ah = ar[u % PF];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) acc[cg] = mfma_t<H16>(ah, bh[cg][u >> 1][u & 1], acc[cg]);
        if constexpr (u + PF < 2 * NT) lds_read_frag(ar[u % PF], la, (u + PF) * 1024);
        gap(uc);
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    // One hidden-layer tile: acc[cg] += W_tile . B over all 2*NT k-steps.  A fragments are read from the
    // slab in groups of G k-steps, two groups in flight (explicit software pipeline; sched_barrier keeps
    // hipcc from sinking the reads back next to their MFMAs), so LDS latency hides behind >= G MFMAs.
    auto mma_step = [&](const u32x4* sl, const u32x4* sll, u32x4 (*bh)[NT][2], u32x4 (*bl)[X3 ? NT : 1][2], f32x16* acc, auto&& gap) {
      if (BWD) {      // (forward-only kernels: the grouped reads below measure the same, 26.4 vs 26.5 ms)
        rolling_mma_impl(std::integral_constant<int, AFX_PF_FWD>{}, sl, bh, acc, gap);
        return;
      }
      constexpr int G = (X3 || NW == 8) ? 2 : 4;
      constexpr int NG = 2 * NT / G;
      u32x4 ab[2][G], al[2][X3 ? G : 1];
      auto ldgrp = [&](int g, int b) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          ab[b][i] = sl[(g * G + i) * 64 + lane];
          if (X3) al[b][i] = sll[(g * G + i) * 64 + lane];
        }
      };
      ldgrp(0, 0);
      if (NG > 1) ldgrp(1, 1);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int u = g * G + i;
          if (X3) {
            acc[0] = mfma_bf16(ab[g & 1][i], bh[0][u >> 1][u & 1], acc[0]);
            acc[0] = mfma_bf16(ab[g & 1][i], bl[0][X3 ? (u >> 1) : 0][u & 1], acc[0]);
            acc[0] = mfma_bf16(al[g & 1][X3 ? i : 0], bh[0][u >> 1][u & 1], acc[0]);
          } else {
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) acc[cg] = mfma_t<H16>(ab[g & 1][i], bh[cg][u >> 1][u & 1], acc[cg]);
          }
        }
        if (g + 2 < NG) ldgrp(g + 2, g & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    STAMP(8);      // tile prologue: samples, first-layer fragments
    if constexpr (!P2) {      // ======== forward half
    // ---------------- layer 0 (always split: hi*hi + hi*lo + lo*hi)
    {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // (PHASE 1: the step before a tile's first one is the previous tile's last layer, which stores nothing: full wait)
        if (t % TPS == 0) stepbase = step_begin(!(P1 && t == 0));
        const u32x4* sl = (const u32x4*)(stepbase + (t % TPS) * SLAB0);     // [(q*2 + part)*64 + lane]
        f32x16 acc[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) acc[cg] = bias_init(0, t);
#pragma unroll
        for (int q = 0; q < NK0; ++q) {
          const u32x4 ah = sl[(q * 2 + 0) * 64 + lane], al = sl[(q * 2 + 1) * 64 + lane];
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) {
            acc[cg] = mfma_bf16(ah, ehi[cg][q], acc[cg]);
            acc[cg] = mfma_bf16(ah, elo[cg][q], acc[cg]);
            acc[cg] = mfma_bf16(al, ehi[cg][q], acc[cg]);
          }
        }
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) epilogue(0, t, acc[cg], cg, hf[cg][t], hl[X3 ? cg : 0][X3 ? t : 0]);
      }
    }
    // backward kernel: the accumulators carry the next tile's bias from one epilogue to the next MFMA loop
    f32x16 accn[NCG];
    if (BWD) {
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) accn[cg] = bias_init(1, 0);
    }
    // ---------------- hidden layers: fragments hs -> hd
    auto fwd_layer = [&](int l, u32x4 (*hs)[NT][2], u32x4 (*hsl)[X3 ? NT : 1][2], u32x4 (*hd)[NT][2], u32x4 (*hdl)[X3 ? NT : 1][2]) {
      // Forward-only kernels defer the epilogue of tile t-1 into step t (behind that step's MFMAs, which do
      // not depend on it) so the VALU work overlaps the matrix pipe.  The backward kernel keeps each
      // epilogue in its own step: its stash stores are what the step's vmcnt(WAITN) counts.
      constexpr bool DEFER = !BWD;
      f32x16 accp[NCG];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t % TPS == 0) stepbase = step_begin(!(SG && l == N && t > 0), true);
        const u32x4* sl = (const u32x4*)(stepbase + (t % TPS) * SLABT);             // hi block [u*64 + lane]
        const u32x4* sll = (const u32x4*)(stepbase + STEPH + (t % TPS) * SLABT);    // lo block (X3)
        f32x16 acc[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) acc[cg] = BWD ? accn[cg] : bias_init(l, t);
        if constexpr (GAPS) {
          // The part of a forward tile's epilogue that needs only its packed activations - ReLU mask bits, 8-bit stash - is
          // issued in the gaps of the NEXT tile's MFMA loop: tile (l, t-1), or the previous layer's last tile at t == 0.
          static_assert(NCG == 1, "");
          const int pl = t > 0 ? l : l - 1, pt = t > 0 ? t - 1 : NT - 1;
          const u32x4* pfr = t > 0 ? hd[0][t - 1] : hs[0][NT - 1];
          unsigned bits = 0, r[4];
          mma_step(sl, sll, hs, hsl, acc, [&](auto uc) {
            if constexpr (decltype(uc)::value < PIECESH) {
              if (t % TPS == 0 && req_pending) {
                request_piece(decltype(uc)::value);
                if constexpr (decltype(uc)::value == PIECESH - 1) request_end();
              }
            }
            static_for<IPG>([&](auto kc) {       // 16 work items over the 2 NT gaps
              constexpr int j = decltype(uc)::value * IPG + decltype(kc)::value;
              if constexpr (j < 8) {
                const unsigned z = nz2(pfr[j >> 2][j & 3], one2);
                bits = j == 0 ? z : (bits | (z << j));
              } else {
                constexpr int i = j - 8;
                r[i >> 1] = bf8_pair<(i & 1) != 0>((i & 1) ? r[i >> 1] : 0u, pfr[i >> 2][i & 3]);
              }
              if constexpr (j == 15) {
                if (pl != N) {
                  ++nstores;
                  stash_store((char*)a.stash_h + (size_t)pl * a.stash_rows * F + (so[0] + (uint32_t)(2 * pt) * 512u), (u32x4){r[0], r[1], r[2], r[3]});
                }
              }
            });
          });
          mk16[((pl * NT + pt) * NCG) * NTH + tid] = (unsigned short)(bits | (bits >> 8));
        } else {
          mma_step(sl, sll, hs, hsl, acc, no_gap);
        }
        if (DEFER) {
          if (t > 0) {
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg)
              epilogue(l, t - 1, accp[cg], cg, hd[cg][t > 0 ? t - 1 : 0], hdl[X3 ? cg : 0][X3 && t > 0 ? t - 1 : 0]);
          }
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) accp[cg] = acc[cg];
        } else {
          STAMP(3);
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) {
            // next tile of this wave: (l, t + 1), or the first tile of layer l + 1 (none behind the last layer)
            const int bl = t + 1 < NT ? l : l + 1, bt = t + 1 < NT ? t + 1 : (l < N ? 0 : -1);
            epilogue(l, t, acc[cg], cg, hd[cg][t], hdl[X3 ? cg : 0][X3 ? t : 0], bl, bt);
            accn[cg] = acc[cg];
          }
          STAMP(4);
        }
      }
      if (DEFER) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
          epilogue(l, NT - 1, accp[cg], cg, hd[cg][NT - 1], hdl[X3 ? cg : 0][X3 ? NT - 1 : 0]);
      }
      if constexpr (GAPS) {
        if (l == N) {      // nothing follows the last layer's last tile: its mask bits here (H_N is not stashed)
          unsigned bits = nz2(hd[0][NT - 1][0][0], one2);
#pragma unroll
          for (int q = 1; q < 8; ++q) bits |= nz2(hd[0][NT - 1][q >> 2][q & 3], one2) << q;
          mk16[((N * NT + NT - 1) * NCG) * NTH + tid] = (unsigned short)(bits | (bits >> 8));
        }
      }
    };
    {
      u32x4 nf[NCG][NT][2];
      u32x4 nl[X3 ? NCG : 1][X3 ? NT : 1][2];
      int l = 1;
      // Forward-only kernels run two layers per loop trip, ping-ponging the two fragment sets (no register copies
      // between layers); the backward kernel and the encoding variants are register-bound and spill with the doubled body.
      if (!BWD && !ENC && AFX_PP_FWD) {
        for (; l + 1 <= N; l += 2) {
          fwd_layer(l, hf, hl, nf, nl);
          fwd_layer(l + 1, nf, nl, hf, hl);
        }
      }
      for (; l <= N; ++l) {
        fwd_layer(l, hf, hl, nf, nl);
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            hf[cg][t][0] = nf[cg][t][0]; hf[cg][t][1] = nf[cg][t][1];
            if (X3) { hl[cg][t][0] = nl[cg][t][0]; hl[cg][t][1] = nl[cg][t][1]; }
          }
      }
    }

    }      // ======== end of the forward half

    // ---------------- output layer + Beer-Lambert / outputs
    float g[NCG];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      float d = dot[cg];
      d += __shfl_xor(d, 32);
      const float raw = d + sm[(N + 2) * F];
      g[cg] = 0.f;
      if constexpr (P2) {      // backward half: dL/draw = dL/d(optical depth of the ray) * g' (PHASE 1 left g' = dt sigma (1 - sigma))
        g[cg] = sp[cg].live ? (a.dod ? a.dod[sp[cg].ray] : 1.f) * a.gpart[m[cg]] : 0.f;      // (dod null: gpart holds the finished dL/draw)
      } else if (a.mode == 0) {
        if (!BWD) { if (hh == 0 && sp[cg].live) a.out[n[cg]] = a.apply_sigmoid ? sigmoidf_(raw) : raw; }
        else g[cg] = sp[cg].live ? a.dod[n[cg]] : 0.f;
      } else {
        const float sig = sigmoidf_(raw);
        const float tau = sp[cg].live ? __fmul_rn(sig, sp[cg].dt) : 0.f;
        if (hh == 0 && sp[cg].live) {
          if (a.sigma) a.sigma[(int64_t)sp[cg].ray * a.n_samples + sp[cg].s] = sig;
          if (a.tau) a.tau[(int64_t)sp[cg].ray * a.n_samples + sp[cg].s] = tau;
        }
        if (!BWD || P1) {
          float od = tau;
#pragma unroll
          for (int sh = 16; sh >= 1; sh >>= 1) od += __shfl_xor(od, sh);
          if (lane == 0 && n[cg] < a.n_total) {
            const int gpr = a.s_pad / GROUP;
            if (a.depth_mode == 4) a.od_part[n[cg] >> 5] = od;      // packed samples: one partial per group of the padded list
            else a.od_part[(int64_t)sp[cg].ray * gpr + (n[cg] - sp[cg].ray * a.s_pad) / GROUP] = od;
          }
          if constexpr (P1) {      // g' of the sample: everything of dL/draw but the ray's dL/d(optical depth)
            g[cg] = sp[cg].live ? sp[cg].dt * (sig * (1.f - sig)) : 0.f;
            if (hh == 0) a.gpart[m[cg]] = g[cg];
          }
        } else if (a.fused) {
          // fused training step: every ray lies inside this workgroup tile (host guarantees s_pad | TS).
          // optical depth of the ray = ordered sum of its 32-sample group partials through LDS
          float od = tau;
#pragma unroll
          for (int sh = 16; sh >= 1; sh >>= 1) od += __shfl_xor(od, sh);
          float* odb = (float*)(slot0 + RING * (size_t)SLOT + (size_t)(N + 1) * MW * NCG * NTH * 4);   // [NW*NCG]
          if (lane == 0) odb[wave * NCG + cg] = od;
          g[cg] = sig * (1.f - sig);      // finished below, once all groups of the tile are in LDS
        } else if (sp[cg].live) g[cg] = a.dod[sp[cg].ray] * sp[cg].dt * (sig * (1.f - sig));
      }
    }
    if (BWD && PHASE == 0 && a.mode != 0 && a.fused) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const float* odb = (const float*)(slot0 + RING * (size_t)SLOT + (size_t)(N + 1) * MW * NCG * NTH * 4);
      const int gpr = a.s_pad / GROUP;
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        const int grp = wave * NCG + cg, g0 = grp / gpr * gpr;
        float od = 0.f;
        for (int k = 0; k < gpr; ++k) od += odb[g0 + k];
        const float T = expf(-od);
        const bool rayok = n[cg] < a.n_total;
        const float tg = rayok ? a.target[sp[cg].ray] : 0.f;
        if (lane == 0 && grp == g0 && rayok) a.pixel[sp[cg].ray] = T;
        // L = mean_r (T_r - target_r)^2 over the GLOBAL batch: dL/dT = 2 (T - target) * inv_n; dT/d(od) = -T
        const float dod = -T * (2.f * (T - tg) * a.inv_n);
        g[cg] = sp[cg].live ? dod * sp[cg].dt * g[cg] : 0.f;
      }
      asm volatile("s_barrier" ::: "memory");      // odb is rewritten by the next tile
    }

    STAMP(7);
    // SG: selection operands of the transposing MFMAs: B_s[k][j] = [j == 16 s + k]; lane (j, hh) holds k = 8 hh .. 8 hh + 7
    u32x4 sel[2];
    auto transpose_tile = [&](const u32x4* frag) -> f32x16 {       // D[n][j] = position 16 s + k of sample n, j = 16 s + k
      f32x16 d = mfma_t<H16>(frag[0], sel[0], (f32x16){0.f});
      return mfma_t<H16>(frag[1], sel[1], d);
    };
    if constexpr (SG) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int e = col - 16 * s2 - 8 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q)      // 1.0 in the operand type: bf16 0x3f80, f16 0x3c00
          sel[s2][q] = (e == 2 * q) ? (H16 ? 0x00003c00u : 0x00003f80u) : ((e == 2 * q + 1) ? (H16 ? 0x3c000000u : 0x3f800000u) : 0u);
      }
    }
    if constexpr (SG && !P2) {      // (PHASE 1: g = g', the common factor dod of the group's ray is applied by k_small_from_groups)
#pragma unroll
      for (int cg = 0; cg < (P1 && a.defer_out ? 0 : NCG); ++cg) {
        float* rec = a.small_part + (size_t)(m[cg] >> 5) * (3 * F + 8);
        // g of the 16 sample rows this lane's accumulator registers hold: one exact f32 MFMA, D[n][j] = g_n for all j
        const f32x16 gT = __builtin_amdgcn_mfma_f32_32x32x2f32(hh == 0 ? g[cg] : 0.f, 1.f, (f32x16){0.f}, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x16 d = transpose_tile(hf[cg][t]);
          float v = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) v = fmaf(gT[r], d[r], v);
          v += __shfl_xor(v, 32);
          if (hh == 0) rec[32 * t + col] = v;
        }
        float sg = g[cg];
#pragma unroll
        for (int sh = 16; sh >= 1; sh >>= 1) sg += __shfl_xor(sg, sh);
        if (lane == 0) rec[3 * F + 6] = sg;
      }
    }
    if constexpr (P1) {
      // the tile's ReLU masks, as they lie in LDS, go to HBM for the backward half: a cooperative 16-byte copy
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave's mask words are written
      char* mg = a.masks + (size_t)(tile - a.tile0) * ((size_t)(N + 1) * MASKB);
      for (uint32_t off = (uint32_t)tid * 16u; off < (uint32_t)(N + 1) * MASKB; off += NTH * 16u)
        __builtin_nontemporal_store(*(const u32x4*)((const char*)mk16 + off), (u32x4*)(mg + off));
      // (the next tile overwrites the LDS image only behind its first step barrier, which every wave reaches after issuing these stores,
      // i.e. after its LDS reads have returned)
    }
    if constexpr (P2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // the tile's masks (requested at the top of the tile) are in LDS
    STAMP(9);      // output layer, compositing, output-layer sums, mask copy / load
    if constexpr (BWD && !P1) {
      // ---------------- input-gradient chain, 16-bit operands, fp32 accumulate.  bf16: dZ_l.  H16: J_l = dZ_l / g.
      u32x4 dz[NCG][NT][2];
      unsigned ghat2[NCG];
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        ghat2[cg] = 0;
        if (((H16 && !S8) || !SG) && hh == 0) a.graw[m[cg]] = g[cg];
        if constexpr (H16) {
          float gm = fabsf(g[cg]);
#pragma unroll
          for (int sh = 16; sh >= 1; sh >>= 1) gm = fmaxf(gm, __shfl_xor(gm, sh));
          // integer max of non-negative float bit patterns: order-independent, so the result is deterministic.  Nothing reads a.gmax before the
          // launch ends, so a wave merges a NEW maximum of its own (`wave_gmax` lives across the tiles of this workgroup) into an LDS word and the
          // workgroup issues ONE global atomic when it has run out of tiles.  One global atomic per wave and tile to ONE address was 56 000
          // same-address atomics per launch at the reference's batch (~6 ns each, serialised at the memory side, and vmcnt-counted, so every
          // wave waited its turn at the next step): the backward half ran 736 us with real gradients against 366 us with all-zero ones (where
          // `gm > 0` never fires); per-wave new maxima alone still cost 42 us of a 1.18 ms iteration at 4x128, 71 us of 0.66 ms at 4x64.
          const uint32_t gbits = __builtin_bit_cast(uint32_t, gm);
#ifndef AFX_NO_GMAX_ATOMIC      // (measurement build --variant=nogmax: wrong weight-gradient scale, timing only)
          if (lane == 0 && gbits > wave_gmax) { atomicMax(wg_gmax_p, gbits); wave_gmax = gbits; }
#endif
          if constexpr (S8) {
            // 8-bit stash: dZ' = g_hat J with g normalised by its 32-sample group's power of two, |g_hat| <= 1; the group's
            // exponent goes to the weight-gradient kernel as the block scale of the MX matrix instruction
            int eg = 0;
            if (gm > 0.f) (void)frexpf(gm, &eg);
            // (the 2^JSHIFT of the stash is folded in HERE: g_hat J alone would sink into f16's subnormals before the conversion)
            const float gh = ldexpf(g[cg], AFX_S8_JSHIFT - eg);
            ghat2[cg] = pack2h(gh, gh);
            if (lane == 0) a.gexp[m[cg] >> 5] = eg;
          }
        }
        const float gs = H16 ? 1.f : g[cg];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const unsigned b32 = mask_expand(mk16[((N * NT + t) * NCG + cg) * NTH + tid]);
          const f32x4* wp = (const f32x4*)(wout_perm + (hh * NT + t) * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = wp[q];
            dz[cg][t][q >> 1][2 * (q & 1)] = pack2t<H16>(w4[0] * gs, w4[1] * gs) & halfmask(b32, 2 * q);
            dz[cg][t][q >> 1][2 * (q & 1) + 1] = pack2t<H16>(w4[2] * gs, w4[3] * gs) & halfmask(b32, 2 * q + 1);
          }
        }
      }
      auto mma_step_plain = [&](const u32x4* sl, u32x4 (*bh)[NT][2], f32x16* acc) { rolling_mma_impl(std::integral_constant<int, AFX_PF_BWD>{}, sl, bh, acc, no_gap); };
      auto stash_dz_tile = [&](int l, int t) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          if constexpr (S8) {
            const f16x2_t gh = __builtin_bit_cast(f16x2_t, ghat2[cg]);
            const f16x8_t gh8 = {gh[0], gh[0], gh[0], gh[0], gh[0], gh[0], gh[0], gh[0]};
            const u32x4 x0 = __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8_t, dz[cg][t][0]) * gh8);
            const u32x4 x1 = __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8_t, dz[cg][t][1]) * gh8);
            // Stochastic rounding to bf8 (v_cvt_scalef32_sr_bf8_f16).  Round-to-nearest is not good enough here: the last hidden
            // layer's dZ'_N = g_hat w_out (masked) has the SAME mantissa for every sample whose g_hat is the same, and on real targets
            // (uniform background, near-constant density along a ray) that is most samples - the same relative error everywhere,
            // nothing averages out: that one layer carried 3.7e-2 of gradient error on bench.py's phantom targets (6e-3 on random
            // targets, like the other layers).  Random bits: one hashed word per lane and tile (sample position, layer, tile) - deterministic,
            // so the step stays bit-reproducible.
            const unsigned src[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            // (seeded by the sample's POSITION in space, not by its index in the launch: the same sample rounds the same way whichever
            // chunk, launch or rank computes it)
            unsigned h = (__float_as_uint(sp[cg].px) * 0x9E3779B1u) ^ (__float_as_uint(sp[cg].py) * 0x7FEB352Du) ^ (__float_as_uint(sp[cg].pz) * 0x846CA68Bu)
                         ^ ((unsigned)(l * NT + t) * 0x85EBCA77u + (unsigned)hh * 0xC2B2AE3Du);
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
            unsigned r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const f16x2_t pr = __builtin_bit_cast(f16x2_t, src[j >> 1]);
              r[j >> 2] = sr_bf8(r[j >> 2], pr[j & 1], h, j & 3);      // one random word per lane and tile: the 16 features of a sample share
            }                                                         // the rounding threshold, which biases no sum over samples
            const u32x4 r8 = {r[0], r[1], r[2], r[3]};
            stash_store((char*)a.stash_dz + (size_t)l * a.stash_rows * F + (so[cg] + (uint32_t)(2 * t) * 512u), r8);
          } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
              stash_store((char*)a.stash_dz + (size_t)l * a.stash_rows * (F * 2) + (so[cg] + (uint32_t)(4 * t + 2 * s) * 512u), dz[cg][t][s]);
          }
        }
      };
      unsigned pk[8], mwprev = 0;      // GAPS: the previous tile's packed, not yet masked input gradients and its mask word
      for (int l = N; l >= 1; --l) {
        u32x4 dn[NCG][NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t % TPS == 0) stepbase = step_begin(!(SG && l == N && t == 0), true);
          const u32x4* sl = (const u32x4*)(stepbase + (t % TPS) * SLABT);
          if constexpr (!GAPS) stash_dz_tile(l, t);                   // SPS stores per step, after the step's request
          f32x16 acc[NCG];
          unsigned mw[NCG];                      // ReLU mask words, read ahead of the MFMA loop
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) {
            acc[cg] = (f32x16){0.f};
            mw[cg] = mk16[(((l - 1) * NT + t) * NCG + cg) * NTH + tid];
          }
          if constexpr (GAPS) {
            // in the gaps of this tile's MFMA chain: the 8-bit stash of dZ'_l tile t (scale by g_hat, convert, store) and the
            // ReLU masking of the previous tile's result; behind the loop only the fp32 -> f16 packing of this tile is left
            unsigned r[4];
            const unsigned b32p = mask_expand(mwprev);
            rolling_mma_impl(std::integral_constant<int, AFX_PF_BWD>{}, sl, dz, acc, [&](auto uc) {
              if constexpr (decltype(uc)::value < PIECESH) {
                if (t % TPS == 0 && req_pending) {
                  request_piece(decltype(uc)::value);
                  if constexpr (decltype(uc)::value == PIECESH - 1) request_end();
                }
              }
              static_for<IPG>([&](auto kc) {
                constexpr int j = decltype(uc)::value * IPG + decltype(kc)::value;
                if constexpr (j < 8) {
                  const unsigned x = pk_mul_f16(dz[0][t][j >> 2][j & 3], ghat2[0]);
                  r[j >> 1] = bf8_pair<(j & 1) != 0>((j & 1) ? r[j >> 1] : 0u, x);
                  if constexpr (j == 7) {
                    ++nstores;
                    stash_store((char*)a.stash_dz + (size_t)l * a.stash_rows * F + (so[0] + (uint32_t)(2 * t) * 512u), (u32x4){r[0], r[1], r[2], r[3]});
                  }
                } else if (t > 0) {
                  constexpr int q = j - 8;
                  pk[q] &= halfmask(b32p, q);
                  if constexpr (j == 15) {
                    dn[0][t > 0 ? t - 1 : 0][0] = (u32x4){pk[0], pk[1], pk[2], pk[3]};
                    dn[0][t > 0 ? t - 1 : 0][1] = (u32x4){pk[4], pk[5], pk[6], pk[7]};
                  }
                }
              });
            });
            STAMP(5);
#pragma unroll
            for (int q = 0; q < 8; ++q) pk[q] = pack2t<H16>(acc[0][2 * q], acc[0][2 * q + 1]);
            mwprev = mw[0];
            if (t == NT - 1) {
              const unsigned b32 = mask_expand(mw[0]);
#pragma unroll
              for (int q = 0; q < 8; ++q) pk[q] &= halfmask(b32, q);
              dn[0][t][0] = (u32x4){pk[0], pk[1], pk[2], pk[3]};
              dn[0][t][1] = (u32x4){pk[4], pk[5], pk[6], pk[7]};
            }
            STAMP(6);
          } else {
            mma_step_plain(sl, dz, acc);
            STAMP(5);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
              // dZ_{l-1} = dH_{l-1} masked by ReLU'(Z_{l-1}): round to bf16, AND the pairs with their half masks
              const unsigned b32 = mask_expand(mw[cg]);
#pragma unroll
              for (int q = 0; q < 8; ++q)
                dn[cg][t][q >> 2][q & 3] = pack2t<H16>(acc[cg][2 * q], acc[cg][2 * q + 1]) & halfmask(b32, q);
            }
            STAMP(6);
          }
        }
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
          for (int t = 0; t < NT; ++t) { dz[cg][t][0] = dn[cg][t][0]; dz[cg][t][1] = dn[cg][t][1]; }
      }
      if constexpr (SG && !ENC) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          float* rec = a.small_part + (size_t)(m[cg] >> 5) * (3 * F + 8);
          float tn, dx, dy, dzz;
          ray_param(a, sp[cg], tn, dx, dy, dzz);
          const float t0 = __shfl(tn, 0);          // the group's first sample: c = its point, t_0 = its ray parameter
          // per-sample weights of the two sums in the C layout: bf16: (1, t_n - t_0); H16 (d = J = dZ / g): (g_n, g_n (t_n - t_0))
          const float w1 = H16 ? g[cg] * (tn - t0) : tn - t0;
          const f32x16 dT = __builtin_amdgcn_mfma_f32_32x32x2f32(hh == 0 ? w1 : 0.f, 1.f, (f32x16){0.f}, 0, 0, 0);
          f32x16 gT = {0.f};
          if constexpr (H16) gT = __builtin_amdgcn_mfma_f32_32x32x2f32(hh == 0 ? g[cg] : 0.f, 1.f, (f32x16){0.f}, 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const f32x16 d = transpose_tile(dz[cg][t]);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0 = H16 ? fmaf(gT[r], d[r], s0) : s0 + d[r]; s1 = fmaf(dT[r], d[r], s1); }
            s0 += __shfl_xor(s0, 32);
            s1 += __shfl_xor(s1, 32);
            rec[F + hh * F + 32 * t + col] = hh ? s1 : s0;
          }
          if (lane == 0) {
            *(f32x4*)(rec + 3 * F) = (f32x4){sp[cg].px, sp[cg].py, sp[cg].pz, dx};
            *(f32x2*)(rec + 3 * F + 4) = (f32x2){dy, dzz};
          }
        }
      } else {      // encoded inputs are not affine in the ray parameter: dZ_0 is stashed and contracted with the input stash (k_wgrad_*)
#pragma unroll
        for (int t = 0; t < NT; ++t) stash_dz_tile(0, t);
      }
    }
  }
  if constexpr (BWD && H16 && !P1) {      // every wave has run out of tiles: the workgroup's one merge into the chunk-wide max |g|
    __syncthreads();
    if (tid == 0 && *wg_gmax_p) atomicMax(a.gmax, *wg_gmax_p);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AFX_STAMP
  if (BWD && NW == 8 && F == AFX_STAMP_F && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 10; ++i) g_stamps[PHASE][wave][i] += ph[i];
#endif
}

// ---------------------------------------------------------------------------------------
// Weight gradients of the hidden layers from the bf16 stashes:
//   dW_l[o][i] = sum_n dZ_l[n][o] * H_{l-1}[n][i],  db_l[o] = sum_n dZ_l[n][o]
// grid = (n_splits, N) (blockIdx.y = l-1), block = 512 (8 waves).  64-sample stages of both stashes
// ([sample][feature] rows) are copied to LDS by LDS-DMA with an XOR chunk swizzle applied on the
// SOURCE address (LDS destination must stay lane-linear); both MFMA operands need the sample index
// as k, i.e. a column of the image: ds_read_b64_tr_b16 (hardware transpose read) delivers it.
// ---------------------------------------------------------------------------------------
// stash position <-> feature index: swap bits 2 and 3 (self-inverse); see the stash stores of k_chain_bf16
__device__ __forceinline__ int fperm(int p) { return (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1); }

// H16: the stashes hold f16 H_{l-1} and the NORMALISED chain J_l = dZ_l / g; the kernel contracts
//   dW_l Ls = sum_n J_l[n][o] * (g_n Ls) H_{l-1}[n][i]   on v_mfma_f32_32x32x16_f16,
// scaling the B fragments by packed (g Ls) pairs (one v_pk_mul_f16 per dword; the pairs are the same for all lanes of a
// half-wave, a broadcast LDS read).  Ls = 2^-e from the chunk's max |g| (wgrad_scale_exp); the reduce kernels undo it.
__device__ __forceinline__ void lds_tr16(s16x4& dst, uint32_t addr, int imm) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory");
}
template <int F, bool H16 = false>
__global__ void __launch_bounds__(512) k_wgrad_bf16(const WgradArgs a) {
  constexpr int NT = F / 32;
  constexpr int TR = NT >= 2 ? NT / 2 : 1, WR = NT / TR;   // row tiles per wave, waves along rows
  constexpr int TC = NT >= 4 ? NT / 4 : 1, WC = NT / TC;
  constexpr int KB = 64;                     // samples per stage (two 32-row stash groups)
  constexpr int NCH = F / 8;                 // 16-byte chunks per stash row
  constexpr int CS = KB * 16 + 64;           // LDS bytes per chunk column: 64 rows x 16 B, +64 B so that the
                                             // four chunks a transpose-read touches fall on disjoint banks
  constexpr int IMG = NCH * CS;              // one operand image
  constexpr int RB = 2 * F;
  constexpr int GOFF = 4 * IMG;              // H16: behind the two stages, per stage [64 x f16 (g Ls) | 64 x f32 (g Ls)]
  constexpr int GST = KB * 2 + KB * 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hh = lane >> 5;
  // provably wave-uniform: `if (i == wc)` around an MFMA must be a SCALAR branch - MFMA ignores EXEC, so under an
  // exec-masked "branch" all four candidates would execute
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.y < N: hidden layer blockIdx.y + 1.  enc16 (encoded inputs): blockIdx.y = N is the FIRST layer - A = dZ_0, B = the 16-bit
  // stash of the 64 input columns (8 chunk columns, natural column order) - and blockIdx.y = N + 1 contracts dZ_0 with
  // d(enc)/d(coef) for the fourier coefficients (k_reduce_coef); both use the two left-most column tiles only.
  const int first = (int)blockIdx.y - a.n_hidden;           // < 0: hidden layer
  const int layer = first < 0 ? (int)blockIdx.y + 1 : 0, split = blockIdx.x;
  const int slot = first <= 0 ? layer : a.n_hidden + 1;     // partial slot
  const int nchb = first < 0 ? NCH : 8;
  float ls = 1.f;
  if constexpr (H16) ls = ldexpf(1.f, -wgrad_scale_exp(a.gmax));
  const char* A = (const char*)a.stash_dz + (size_t)layer * a.stride_rows * RB;
  const char* B = first < 0 ? (const char*)a.stash_h + (size_t)(layer - 1) * a.stride_rows * RB
                            : (const char*)a.stash_e + (size_t)first * a.stride_rows * 128;
  int64_t r0 = (int64_t)split * a.rows_per_split;
  int64_t r1 = r0 + a.rows_per_split;
  if (r1 > a.rows) r1 = a.rows;
  const int nst = r1 > r0 ? (int)((r1 - r0) / KB) : 0;
  const int wr = wave / WC, wc = wave % WC;
  const bool active = wave < WR * WC;      // (first layer: only column tiles 0, 1 of B exist; the other waves' products are never stored)

  f32x16 acc[TR][TC];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j) acc[i][j] = (f32x16){0.f};

  // one LDS-DMA instruction per chunk column: lane = stage row; source = two contiguous 512-byte runs
  auto stage_load = [&](int st, int buf) {
    char* dA = lds + buf * 2 * IMG;
    char* dB = dA + IMG;
    const int64_t g0 = (r0 + (int64_t)st * KB) >> 5;
    for (int c = wave; c < NCH; c += 8) {
      const size_t src = ((((size_t)(g0 + hh) * NCH + c) << 5) + col) << 4;
      // default cache policy: non-temporal loads (aux = 2) measured 2 ms slower per step here
      __builtin_amdgcn_global_load_lds(GPTR(A + src), LPTR(dA + c * CS), 16, 0, 0);
      if (first < 0) __builtin_amdgcn_global_load_lds(GPTR(B + src), LPTR(dB + c * CS), 16, 0, 0);
    }
    if (first >= 0) {       // 8 chunk columns of the input stash: one per wave
      const size_t src = ((((size_t)(g0 + hh) * 8 + wave) << 5) + col) << 4;
      __builtin_amdgcn_global_load_lds(GPTR(B + src), LPTR(dB + wave * CS), 16, 0, 0);
    }
  };
  // transpose-read (ds_read_b64_tr_b16): lane 4q+p of each 16-lane group supplies the address of stage row
  // (block row q), positions 4p..4p+3; lane i of the group receives position i of the 4 rows.
  const int g4 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  auto tr_off = [&](int ks, int rd, int fb0) -> int {
    const int row = ks * 16 + 8 * (g4 >> 1) + 4 * rd + tq;
    const int c = fb0 + 16 * (g4 & 1) + 4 * tp;
    return (c >> 3) * CS + row * 16 + 8 * (tp & 1);
  };
  // H16: g of the stage's 64 samples: loaded by the first wave with the stage's DMA, written to LDS once it has landed
  float greg = 0.f;
  auto g_load = [&](int st) { if (H16 && tid < KB) greg = a.graw[r0 + (int64_t)st * KB + tid]; };
  // Every LDS read of the loop is inline asm (see k_wgrad_s8: hipcc otherwise drains ALL outstanding LDS-DMA with
  // s_waitcnt vmcnt(0) in front of the first read, i.e. the next stage's DMA never overlapped this stage's MFMAs);
  // bias gradients come out of the matrix pipe: B = the (g Ls) fragment (H16) or a fragment of ones (bf16).
  const uint32_t lbase = (uint32_t)(uintptr_t)LPTR(lds);
  // lane part of a transposed read's address (tr_off) + the operand's first 32-position tile; the rest is an immediate
  const uint32_t lpart = (uint32_t)(((16 * (g4 & 1) + 4 * tp) >> 3) * CS + (8 * (g4 >> 1) + tq) * 16 + 8 * (tp & 1));
  const uint32_t offA = lpart + (uint32_t)(4 * wr * TR * CS), offB = lpart + (uint32_t)(IMG + 4 * wc * TC * CS);
  const bool has_bias = active && wc < TR;
  f32x16 accb = (f32x16){0.f};
  if (nst > 0) { stage_load(0, 0); g_load(0); }
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (H16 && tid < KB) {
      const float gs = greg * ls;
      char* gb = lds + GOFF + (st & 1) * GST;
      *(_Float16*)(gb + tid * 2) = (_Float16)gs;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + 1 < nst) { stage_load(st + 1, (st + 1) & 1); g_load(st + 1); }
    if (active) {
      const uint32_t sA = lbase + (uint32_t)((st & 1) * 2 * IMG);
      const uint32_t sG = lbase + (uint32_t)(GOFF + (st & 1) * GST + 16 * hh);
#pragma unroll
      for (int ks = 0; ks < KB / 16; ++ks) {
        s16x4 ax[TR][2], bx[TC][2];
        u32x4 gpk = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};       // bf16 ones
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int rd = 0; rd < 2; ++rd) lds_tr16(ax[i][rd], sA + offA, 4 * i * CS + ks * 256 + rd * 64);
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
          for (int rd = 0; rd < 2; ++rd) lds_tr16(bx[j][rd], sA + offB, 4 * j * CS + ks * 256 + rd * 64);
        if constexpr (H16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(gpk) : "v"(sG), "n"(ks * 32) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32x4 af[TR], bf[TC];
#pragma unroll
        for (int i = 0; i < TR; ++i) af[i] = __builtin_bit_cast(u32x4, (s16x8){ax[i][0][0], ax[i][0][1], ax[i][0][2], ax[i][0][3], ax[i][1][0], ax[i][1][1], ax[i][1][2], ax[i][1][3]});
#pragma unroll
        for (int j = 0; j < TC; ++j) {
          bf[j] = __builtin_bit_cast(u32x4, (s16x8){bx[j][0][0], bx[j][0][1], bx[j][0][2], bx[j][0][3], bx[j][1][0], bx[j][1][1], bx[j][1][2], bx[j][1][3]});
          // H16: element j of a fragment is stage row 16 ks + 8 hh + j: four packed (g Ls) pairs, the same for the whole half-wave
          // (whole-vector multiply: element-wise writes through bf[j][q] were miscompiled by hipcc 7.2 into a chain on element 0)
          if constexpr (H16) bf[j] = __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8_t, bf[j]) * __builtin_bit_cast(f16x8_t, gpk));
        }
#pragma unroll
        for (int i = 0; i < TR; ++i) {
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = mfma_t<H16>(af[i], bf[j], acc[i][j]);
          if (i == wc) accb = mfma_t<H16>(af[i], gpk, accb);
        }
      }
    }
  }
  float* P = a.partial + ((size_t)slot * a.n_splits + split) * F * F;
  if (active && first < 0) {
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          P[(size_t)fperm(32 * (wr * TR + i) + rowperm(r) + 4 * hh) * F + fperm(32 * (wc * TC + j) + col)] = acc[i][j][r];
  } else if (active) {      // first layer: rows [F] x k0pad = 64 input columns in natural order (the layout k_reduce_w expects)
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
        if (wc * TC + j < 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            P[(size_t)fperm(32 * (wr * TR + i) + rowperm(r) + 4 * hh) * a.k0pad + 32 * (wc * TC + j) + col] = acc[i][j][r];
        }
  }
  if (has_bias && col == 0 && first <= 0) {        // every column of accb holds the row sums
#pragma unroll
    for (int r = 0; r < 16; ++r)
      a.partial2[((size_t)layer * a.n_splits + split) * (F + 4) + fperm(32 * (wr * TR + wc) + rowperm(r) + 4 * hh)] = accb[r];
  }
}

// ---------------------------------------------------------------------------------------
// Weight gradients from the 8-bit stash (f16 mode, S8): H_{l-1} and dZ'_l = g_hat J_l 2^JSHIFT as bf8 (e5m2), layout
// [row>>5][p8>>4][row&31][16 B], contracted on the block-scaled matrix instruction
//   v_mfma_scale_f32_32x32x64_f8f6f4 (bf8 x bf8, K = 64 = one stage of two 32-sample groups, twice the bf16 MFMA rate):
//   dW_l 2^(JSHIFT - E) = sum_groups 2^(e_g - E) * sum_{n in group} dZ'_l[n][o] H_{l-1}[n][i]
// K-blocks of the instruction (measured, tools/micro/mx_probe.hip and mx_probe2.hip): element (lane half, register, byte)
// of A meets the same element of B; block 0 = registers 0-3 of BOTH lane halves, scaled by the E8M0 byte the lanes of
// half 0 supply; block 1 = registers 4-7, scaled by the lanes of half 1.  So group 0 of the stage (32 samples) fills
// registers 0-3 (16 samples per lane half), group 1 registers 4-7, the group's exponent e_g - E is the A operand's block
// scale, and no per-sample factor is left for the VALU: the bf8 bytes go from LDS to the matrix pipe unconverted.
// A 64-sample stage is 2 x F/16 LDS-DMA instructions (lane = stage row, 16 bytes); four ds_read_b64_tr_b8 deliver the
// 32 samples x 1 position of a lane (lane 2q+p of a 16-lane group supplies row q, bytes 8p..8p+7 of a 16-byte chunk; lane
// i receives byte i of the 8 rows).  Chunk columns are padded to 1152 B: the two 16-lane groups of a half-wave read 128
// contiguous bytes each, 32 banks apart.  Bias gradients: one more MFMA per stage against a B operand of ones.
// Pipeline: a ring of NS = 4 stages in LDS (the 8-bit stage is 36 KiB), NS-1 stages of LDS-DMA in flight per workgroup
// (~100 KB per CU: enough to cover the HBM latency at full bandwidth; with one stage in flight the kernel ran at the
// DMA round trip per stage, 2.6 TB/s).  One barrier per stage; every wave waits for its OWN DMA instructions of the
// stage with a counted vmcnt (they complete in order), the barrier then covers the other waves' pieces.  The two group
// exponents of the stage ride along as a 256-byte DMA by wave 0.
// Every LDS read of the loop is inline asm: hipcc orders a plain LDS load (and the ds_read_tr builtins) behind ALL
// outstanding LDS-DMA with an s_waitcnt vmcnt(0), which would put the whole DMA round trip back in front of every stage.
// ---------------------------------------------------------------------------------------
// H6 (6-bit H stash, see stash_off6): B = H_{l-1} arrives as bf6.  Its stage pieces are 12-byte LDS-DMA instructions
// (global_load_lds_dwordx3 writes lane l's 12 bytes at base + 16 l: the LDS image keeps the bf8 image's geometry, rows at a 16-byte stride,
// which is what ds_read_b96_tr_b6 wants - tools/micro/fp6_probe2.hip), two transposed reads per column tile deliver a lane's 32 samples x 1
// position (lane i of a 16-lane group supplies row i and receives field i of the 16 rows), and the instruction takes A = bf8, B = bf6
// (cbsz 1, blgp 3).  K mapping of the mixed formats (tools/micro/fp6_probe.hip): B's lane half h holds K = 32 h + e in field e, A's
// K = 32 (k >> 4) + 16 h + (k & 15) at byte k as before; B's block scale for K-block b comes from the lanes of half b like A's.  The E8M0
// byte of this wave's column tile pair is picked out of the group's dword.
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x3 __attribute__((ext_vector_type(3)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void lds_tr6(i32x3& dst, uint32_t addr, int imm) {
  asm volatile("ds_read_b96_tr_b6 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory");
}
__device__ __forceinline__ void lds_tr8(i32x2& dst, uint32_t addr, int imm) {
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory");
}
__device__ __forceinline__ void lds_rd32(int& dst, uint32_t addr, int imm) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory");
}

template <int F, bool B6>      // B6: B = the 6-bit H stash (hidden layers); otherwise bf8 (the encoded inputs' rows, and every row of the h8 A/B build)
__global__ void __launch_bounds__(512) k_wgrad_s8(const WgradArgs a) {
  constexpr int NT = F / 32;
  constexpr int TR = NT >= 2 ? NT / 2 : 1, WR = NT / TR;
  constexpr int TC = NT >= 4 ? NT / 4 : 1, WC = NT / TC;
  constexpr int KB = 64;
  constexpr int NCH = F / 16;                // 16-byte chunk columns per stash row
  constexpr int CS = KB * 16 + 128;
  constexpr int IMG = NCH * CS;
  constexpr int NS = 4;                      // ring depth
  constexpr int GOFF = NS * 2 * IMG;         // per stage: the two group exponents, replicated over 64 dwords
  constexpr int PER = 2 * ((NCH + 7) / 8);   // LDS-DMA instructions per wave and stage (waves beyond NCH: none, their waits are no-ops)
  constexpr bool H6 = B6;
  constexpr int XW0 = H6 ? 2 : 1;            // wave 0's extra pieces per stage: the group exponents (and H's block scales)
  constexpr int HOFF = GOFF + NS * KB * 4;   // per stage: the two groups' H block-scale dwords, replicated like the exponents
  static_assert(2 * (NT - 1) * CS + 3 * 128 + 1151 < 65536 && IMG < 65536, "ds offset immediates are 16 bits");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.y < N: hidden layer blockIdx.y + 1.  enc16 (encoded inputs): blockIdx.y = N is the FIRST layer - A = dZ'_0, B = the 8-bit stash
  // of the 64 input columns (4 chunk columns, natural column order) - and blockIdx.y = N + 1 the fourier-coefficient contraction
  // (B = d(enc)/d(coef)/(2 pi)); both produce the two left-most column tiles only.
  const int first = (int)blockIdx.y + a.y0 - a.n_hidden;    // < 0: hidden layer (a.y0: first grid row of this launch)
  const int layer = first < 0 ? (int)blockIdx.y + a.y0 + 1 : 0, split = blockIdx.x;
  const int slot = first <= 0 ? layer : a.n_hidden + 1;     // partial slot
  const int E = wgrad_scale_exp(a.gmax);
  const char* A = (const char*)a.stash_dz + (size_t)layer * a.stride_rows * F;
  const char* B = first < 0 ? (const char*)a.stash_h + (size_t)(layer - 1) * a.stride_rows * F
                            : (const char*)a.stash_e + (size_t)first * a.stride_rows * 64;
  const uint32_t* hexp = a.hexp + (size_t)(first < 0 ? layer - 1 : 0) * (size_t)(a.stride_rows >> 5);      // (first layer: read, not used)
  int64_t r0 = (int64_t)split * a.rows_per_split;
  int64_t r1 = r0 + a.rows_per_split;
  if (r1 > a.rows) r1 = a.rows;
  const int nst = r1 > r0 ? (int)((r1 - r0) / KB) : 0;
  const bool active = wave < WR * WC;
  const int wr = wave / WC, wc = wave % WC;
  const bool has_bias = active && wc < TR;   // this wave also sums the bias gradient of its row tile wc

  f32x16 acc[TR][TC], accb = (f32x16){0.f};
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j) acc[i][j] = (f32x16){0.f};
  const i32x8 ones8 = {0x3c3c3c3c, 0x3c3c3c3c, 0x3c3c3c3c, 0x3c3c3c3c, 0x3c3c3c3c, 0x3c3c3c3c, 0x3c3c3c3c, 0x3c3c3c3c};     // bf8 1.0

  auto stage_load = [&](int st) {
    const int buf = st % NS;
    char* dA = lds + buf * 2 * IMG;
    char* dB = dA + IMG;
    const int64_t g0 = (r0 + (int64_t)st * KB) >> 5;
    for (int c = wave; c < NCH; c += 8) {
      const size_t src = ((((size_t)(g0 + hh) * NCH + c) << 5) + col) << 4;
      __builtin_amdgcn_global_load_lds(GPTR(A + src), LPTR(dA + c * CS), 16, 0, 0);
      if constexpr (B6) __builtin_amdgcn_global_load_lds(GPTR(B + (src >> 4) * 12), LPTR(dB + c * CS), 12, 0, 0);
      else if (first < 0) __builtin_amdgcn_global_load_lds(GPTR(B + src), LPTR(dB + c * CS), 16, 0, 0);
      else {      // input stash: 4 chunk columns; every wave still issues one B piece per A piece (column c mod 4, identical data where
                  // two waves meet), so that the counted vmcnt below holds for every grid row
        const size_t srcb = ((((size_t)(g0 + hh) * 4 + (c & 3)) << 5) + col) << 4;
        __builtin_amdgcn_global_load_lds(GPTR(B + srcb), LPTR(dB + (c & 3) * CS), 16, 0, 0);
      }
    }
    if (wave == 0) {
      __builtin_amdgcn_global_load_lds(GPTR(a.gexp + g0 + hh), LPTR(lds + GOFF + buf * (KB * 4)), 4, 0, 0);
      if constexpr (H6) __builtin_amdgcn_global_load_lds(GPTR(hexp + g0 + hh), LPTR(lds + HOFF + buf * (KB * 4)), 4, 0, 0);
    }
  };
  const int g4 = lane >> 4, li = lane & 15;
  // lane 2q+p of a 16-lane group: row q of an 8-row block, bytes 8p..8p+7 of chunk 2T + (g4&1); read r (registers 2r, 2r+1)
  // of lane half h covers stage rows 32 (r>>1) + 16 h + 8 (r&1) + 0..7
  const uint32_t lbase = (uint32_t)(uintptr_t)LPTR(lds);
  const uint32_t troff = (uint32_t)((g4 & 1) * CS + (16 * (g4 >> 1) + (li >> 1)) * 16 + 8 * (li & 1));
  const uint32_t offA = troff + (uint32_t)(2 * wr * TR * CS), offB = troff + (uint32_t)(IMG + 2 * wc * TC * CS);
  // bf6: lane i of 16-lane group g4 supplies row 32 (g4 >> 1) + i (+ 16 for the second read) of unit 2T + (g4 & 1)
  const uint32_t offB6 = (uint32_t)(IMG + 2 * wc * TC * CS + (g4 & 1) * CS + (32 * (g4 >> 1) + li) * 16);

#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nst) stage_load(i);
  for (int st = 0; st < nst; ++st) {
    // stage st has landed once this wave's younger (NS-2 stages of) DMA instructions are all that is outstanding
#ifdef AFX_SAFE_WAITS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if (st + NS - 2 < nst) {
      if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * (PER + XW0)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * PER) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last stages: fewer younger ones to count on
#endif
    // raw barrier (__syncthreads() would drain vmcnt): every wave's pieces of stage st are in LDS, and every wave is done
    // reading stage st-1 (its LDS reads have returned: lgkmcnt(0)), whose slot the next request overwrites
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + NS - 1 < nst) stage_load(st + NS - 1);
    if (active) {
      const uint32_t sbase = lbase + (uint32_t)((st % NS) * 2 * IMG);
      i32x2 ax[TR][4], bx[TC][4];
      i32x3 b6[TC][2];
      int eg, hx = 0;
      constexpr bool six = B6;
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_tr8(ax[i][q], sbase + offA, 2 * i * CS + (q >> 1) * 512 + (q & 1) * 128);
      if constexpr (six) {
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
          for (int q = 0; q < 2; ++q) lds_tr6(b6[j][q], sbase + offB6, 2 * j * CS + q * 256);
        lds_rd32(hx, lbase + (uint32_t)(HOFF + (st % NS) * (KB * 4)) + (uint32_t)lane * 4u, 0);
      } else {
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) lds_tr8(bx[j][q], sbase + offB, 2 * j * CS + (q >> 1) * 512 + (q & 1) * 128);
      }
      lds_rd32(eg, lbase + (uint32_t)(GOFF + (st % NS) * (KB * 4)) + (uint32_t)lane * 4u, 0);
      // the reads are inline asm, so the compiler does not know their results are in flight: every result register is an operand of the
      // wait, or a copy / use of it may be scheduled in front of the wait (seen: the v_mov that assembles the bf6 operand)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eg), "+v"(hx) :: "memory");
#pragma unroll
      for (int i = 0; i < TR; ++i) asm volatile("" : "+v"(ax[i][0]), "+v"(ax[i][1]), "+v"(ax[i][2]), "+v"(ax[i][3]));
      if constexpr (six) {
#pragma unroll
        for (int j = 0; j < TC; ++j) asm volatile("" : "+v"(b6[j][0]), "+v"(b6[j][1]));
      } else {
#pragma unroll
        for (int j = 0; j < TC; ++j) asm volatile("" : "+v"(bx[j][0]), "+v"(bx[j][1]), "+v"(bx[j][2]), "+v"(bx[j][3]));
      }
      // E8M0 block scale supplied by this lane half for ITS group (half 0 -> block 0 = group 0): 2^(e_group - E) (e_group <= E; a group without gradient has e_group = 0 and zeros)
      int sc = 127 + eg - E;
      sc = sc < 0 ? 0 : (sc > 127 ? 127 : sc);
      if constexpr (six) {
#pragma unroll
        for (int i = 0; i < TR; ++i) {
          const i32x8 a8 = {ax[i][0][0], ax[i][0][1], ax[i][1][0], ax[i][1][1], ax[i][2][0], ax[i][2][1], ax[i][3][0], ax[i][3][1]};
#pragma unroll
          for (int j = 0; j < TC; ++j) {
            const i32x8 bb = {b6[j][0][0], b6[j][0][1], b6[j][0][2], b6[j][1][0], b6[j][1][1], b6[j][1][2], 0, 0};
            const int sb = (int)(((unsigned)hx >> (8 * ((wc * TC + j) >> 1))) & 0xffu);      // this column tile's pair
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, bb, acc[i][j], 1, 3, 0, sc, 0, sb);
          }
          if (i == wc) accb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, ones8, accb, 1, 1, 0, sc, 0, 127);      // (wc < TR: wave-uniform)
        }
      } else {
        i32x8 b8[TC];
#pragma unroll
        for (int j = 0; j < TC; ++j) b8[j] = (i32x8){bx[j][0][0], bx[j][0][1], bx[j][1][0], bx[j][1][1], bx[j][2][0], bx[j][2][1], bx[j][3][0], bx[j][3][1]};
#pragma unroll
        for (int i = 0; i < TR; ++i) {
          const i32x8 a8 = {ax[i][0][0], ax[i][0][1], ax[i][1][0], ax[i][1][1], ax[i][2][0], ax[i][2][1], ax[i][3][0], ax[i][3][1]};
#pragma unroll
          for (int j = 0; j < TC; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[j], acc[i][j], 1, 1, 0, sc, 0, 127);
          if (i == wc) accb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, ones8, accb, 1, 1, 0, sc, 0, 127);      // (wc < TR: wave-uniform)
        }
      }
    }
  }
  float* P = a.partial + ((size_t)slot * a.n_splits + split) * F * F;
  if (active && first < 0) {
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          P[(size_t)fperm8(32 * (wr * TR + i) + rowperm(r) + 4 * hh) * F + fperm8(32 * (wc * TC + j) + col)] = acc[i][j][r];
  } else if (active) {      // first layer: rows [F] x k0pad = 64 input columns in natural order (the layout k_reduce_w expects)
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
        if (wc * TC + j < 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            P[(size_t)fperm8(32 * (wr * TR + i) + rowperm(r) + 4 * hh) * a.k0pad + 32 * (wc * TC + j) + col] = acc[i][j][r];
        }
  }
  if (has_bias && col == 0 && first <= 0) {        // every column of accb holds the row sums
#pragma unroll
    for (int r = 0; r < 16; ++r)
      a.partial2[((size_t)layer * a.n_splits + split) * (F + 4) + fperm8(32 * (wr * TR + wc) + rowperm(r) + 4 * hh)] = accb[r];
  }
}

// First-layer weight gradient for RAW-coordinate inputs (dW_0 = dZ_0^T x, 3 columns, fp32 on the VALU), its bias, and the output
// layer (dw_out = sum_n g_n H_N[n], db_out = sum_n g_n) from the 16-bit stashes: one pass over dZ_0, H_N, x, g.  With an
// encoding (a.enc16) the first layer's weights and bias come out of k_wgrad_bf16 (64 input columns on the matrix pipe;
// this VALU kernel took 74 % of a BARF training iteration when it did them) and only the output layer is summed here.
// grid = (n_records, F/64), block = 256 = 8 chunk columns x 32 rows: a wave reads two contiguous 512-byte runs of the
// chunk-major stash; blockIdx.y selects 64 stash positions.  Block b writes its partial record of
// SS = F*k0pad + 2F + 4 floats; k_reduce_small sums the records in order.
template <int F, bool H16 = false>
__global__ void __launch_bounds__(256) k_small_grads_bf16(const WgradArgs a) {
  constexpr int KMAX = 4;
  constexpr int PPT = 8;
  constexpr int NCH = F / 8;
  const int cq = threadIdx.x >> 5, rr = threadIdx.x & 31;
  const int ch = blockIdx.y * 8 + cq;                         // chunk column of this thread
  const int64_t ngroups = a.rows >> 5;
  const int64_t per = (ngroups + gridDim.x - 1) / gridDim.x;
  int64_t g0 = (int64_t)blockIdx.x * per, g1 = g0 + per;
  if (g1 > ngroups) g1 = ngroups;
  const char* dz0 = (const char*)a.stash_dz;
  const char* hN = (const char*)a.stash_h + (size_t)a.n_hidden * a.stride_rows * F * 2;
  const bool raw_inputs = !a.enc16;
  float acc[PPT][KMAX], bs[PPT], so[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    bs[i] = so[i] = 0.f;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) acc[i][c] = 0.f;
  }
  float sg = 0.f;
#pragma unroll 2
  for (int64_t g = g0; g < g1; ++g) {
    const size_t off = ((((size_t)g * NCH + ch) << 5) + rr) << 4;
    const u32x4 x = *(const u32x4*)(dz0 + off), y = *(const u32x4*)(hN + off);
    const int64_t r = (g << 5) + rr;
    const float gr = a.graw[r];
    sg += gr;
    float ev[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) ev[c] = (raw_inputs && c < a.k0) ? a.stash_e[r * a.k0pad + c] : 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const unsigned dw = i < 2 ? x[0] : (i < 4 ? x[1] : (i < 6 ? x[2] : x[3]));
      const unsigned hw = i < 2 ? y[0] : (i < 4 ? y[1] : (i < 6 ? y[2] : y[3]));
      // H16: the stash holds J_0 = dZ_0 / g
      const float d = ((i & 1) ? hi_t<H16>(dw) : lo_t<H16>(dw)) * (H16 ? gr : 1.f);
      const float h = (i & 1) ? hi_t<H16>(hw) : lo_t<H16>(hw);
      bs[i] += d;
      so[i] = fmaf(gr, h, so[i]);
#pragma unroll
      for (int c = 0; c < KMAX; ++c) acc[i][c] = fmaf(d, ev[c], acc[i][c]);
    }
  }
  // sum over the 32 rows (lanes of one half-wave)
  auto red32 = [&](float v) -> float {
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
    return v;
  };
  const size_t SS = (size_t)F * a.k0pad + 2 * F + 4;
  float* P = a.partial_s + (size_t)blockIdx.x * SS;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int f = fperm(ch * 8 + i);
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
      const float v = red32(acc[i][c]);
      if (rr == 0 && raw_inputs && c < a.k0) P[(size_t)f * a.k0pad + c] = v;
    }
    const float vb = red32(bs[i]), vo = red32(so[i]);
    if (rr == 0) { P[(size_t)F * a.k0pad + f] = vb; P[(size_t)F * a.k0pad + F + f] = vo; }
  }
  const float vg = red32(sg);
  if (blockIdx.y == 0 && threadIdx.x == 0) P[(size_t)F * a.k0pad + 2 * F] = vg;
}

// SG mode: first-layer / output-layer partial records (the layout k_small_grads_bf16 writes) from the per-group sums
// the chain kernel left where H_N's stash would have been.  grid = n_small blocks, block = F threads (thread p =
// stash position, feature fperm(p)); block b sums its contiguous range of groups in order.
template <int F>
__global__ void __launch_bounds__(4 * F) k_small_from_groups(const WgradArgs a) {
  // block = (F, 4): four interleaved group sequences per block (combined in fixed order through LDS) - the record loop is a chain of dependent
  // loads (group -> ray -> dod), and at the reference's batch sizes four times the waves in flight is what shortens it
  __shared__ float red[3][6][F];
  const int p = threadIdx.x, f = fperm(p), q = threadIdx.y;
  const int64_t ngroups = a.rows >> 5;
  const int64_t per = (ngroups + gridDim.x - 1) / gridDim.x;
  int64_t g0 = (int64_t)blockIdx.x * per, g1 = g0 + per;
  if (g1 > ngroups) g1 = ngroups;
  constexpr int RS = 3 * F + 8;
  const float* base = a.records ? a.records : (const float*)((const char*)a.stash_h + (size_t)a.n_hidden * a.stride_rows * F * a.stash_esz);
  float aw = 0.f, a0 = 0.f, ax = 0.f, ay = 0.f, az = 0.f, sg = 0.f;
#pragma unroll 2
  for (int64_t g = g0 + q; g < g1; g += 4) {
    const float* rec = base + g * RS;
    // (split phases: SW and sum g were formed with g' = g / dod[ray]; a group never straddles rays)
    // (the groups that pad the last tile belong to no ray: their sums are zero, and neither group_ray nor dod has an entry for them)
    float ds = 1.f;
    if (a.dod) {
      const int64_t gg = a.group0 + g;
      ds = gg < a.n_groups_valid ? a.dod[a.group_ray ? (int64_t)a.group_ray[gg] : gg / a.gpr] : 0.f;
    }
    const float sw = a.no_sw ? 0.f : rec[p] * ds;
    if (a.enc16) {          // encoded inputs: only the output layer's sums are in the records (first layer: k_wgrad_s8)
      aw += sw;
      sg += rec[3 * F + 6] * ds;
      continue;
    }
    const float s0 = rec[F + p], s1 = rec[2 * F + p];
    const f32x4 c4 = *(const f32x4*)(rec + 3 * F);
    const f32x4 d4 = *(const f32x4*)(rec + 3 * F + 4);      // dy, dz, sum g, -
    aw += sw;
    a0 += s0;
    ax += fmaf(c4[0], s0, c4[3] * s1);
    ay += fmaf(c4[1], s0, d4[0] * s1);
    az += fmaf(c4[2], s0, d4[1] * s1);
    sg += a.no_sw ? 0.f : d4[2] * ds;
  }
  if (q > 0) { float* r = &red[q - 1][0][p]; r[0] = aw; r[F] = a0; r[2 * F] = ax; r[3 * F] = ay; r[4 * F] = az; r[5 * F] = sg; }
  __syncthreads();
  if (q > 0) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float* r = &red[i][0][p];
    aw += r[0]; a0 += r[F]; ax += r[2 * F]; ay += r[3 * F]; az += r[4 * F]; sg += r[5 * F];
  }
  const size_t SS = (size_t)F * a.k0pad + 2 * F + 4;
  float* P = a.partial_s + (size_t)blockIdx.x * SS;
  P[(size_t)f * a.k0pad + 0] = ax;
  P[(size_t)f * a.k0pad + 1] = ay;
  P[(size_t)f * a.k0pad + 2] = az;
  P[(size_t)F * a.k0pad + f] = a0;
  if (a.no_sw) return;      // the output layer's slots of the record belong to k_wout_stash8
  P[(size_t)F * a.k0pad + F + f] = aw;
  if (p == 0) P[(size_t)F * a.k0pad + 2 * F] = sg;
}

// Output-layer gradient from the 8-bit stash of H_N (hierarchical step with coarse re-use: dL/draw of a sample is known only after both passes'
// forward halves, so the chain kernel cannot form sum_n g_n H_N[n] in registers): dW_out[f] = sum_n g_n H_N[n][f], db_out = sum_n g_n.
// grid = n_small blocks (the record partition of k_small_from_groups: block b owns a contiguous range of 32-sample groups), block = 2 F threads =
// (32 rows) x (F / 16 chunks): a thread reads one 16-byte chunk (16 stash positions of one sample) per group, converts, scales by the sample's g
// and accumulates; the 32 rows are summed by lane shuffles; results land in the output-layer slots of the block's partial record.
template <int F>
__global__ void __launch_bounds__(2 * F) k_wout_stash8(const WgradArgs a) {
  const int rr = threadIdx.x & 31, ch = threadIdx.x >> 5;      // row within the group, 16-byte chunk column
  constexpr int NCH = F / 16;
  const int64_t ngroups = a.rows >> 5;
  const int64_t per = (ngroups + gridDim.x - 1) / gridDim.x;
  int64_t g0 = (int64_t)blockIdx.x * per, g1 = g0 + per;
  if (g1 > ngroups) g1 = ngroups;
  const char* hN = (const char*)a.stash_h + (size_t)a.n_hidden * a.stride_rows * F;
  float acc[16], sg = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int64_t g = g0; g < g1; ++g) {
    const u32x4 x = *(const u32x4*)(hN + (((size_t)(g * NCH + ch) << 5) + rr) * 16);
    const float gr = a.gfull[(g << 5) + rr];
    if (ch == 0) sg += gr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[4 * q + 0] = fmaf(gr, __builtin_amdgcn_cvt_f32_bf8(x[q], 0), acc[4 * q + 0]);
      acc[4 * q + 1] = fmaf(gr, __builtin_amdgcn_cvt_f32_bf8(x[q], 1), acc[4 * q + 1]);
      acc[4 * q + 2] = fmaf(gr, __builtin_amdgcn_cvt_f32_bf8(x[q], 2), acc[4 * q + 2]);
      acc[4 * q + 3] = fmaf(gr, __builtin_amdgcn_cvt_f32_bf8(x[q], 3), acc[4 * q + 3]);
    }
  }
  const size_t SS = (size_t)F * a.k0pad + 2 * F + 4;
  float* P = a.partial_s + (size_t)blockIdx.x * SS;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float v = acc[j];
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);      // the 32 rows of a chunk column are one half-wave
    if (rr == 0) P[(size_t)F * a.k0pad + F + fperm8(ch * 16 + j)] = v;
  }
#pragma unroll
  for (int sh = 16; sh >= 1; sh >>= 1) sg += __shfl_xor(sg, sh);
  if (threadIdx.x == 0) P[(size_t)F * a.k0pad + 2 * F] = sg;
}

// grid = ceil(SS/64) blocks of 64 x 4 threads: 4 record groups per element (fixed-order partial sums, combined in
// fixed order through LDS), so that the ~5 MB reduction spreads over the whole chip instead of 18 workgroups.
template <int F>
__device__ __forceinline__ void reduce_small_body(const ReduceArgs& a, int bx, int tx, int grp) {
  __shared__ float red[4][64];
  const size_t SS = (size_t)F * a.k0pad + 2 * F + 4;
  const size_t e = (size_t)bx * 64 + tx;
  const bool valid = e <= (size_t)F * a.k0pad + 2 * F;
  float s = 0.f;
  if (valid) {
    const int per = (a.n_small + 3) / 4;
    const int b1 = min((grp + 1) * per, a.n_small);
#pragma unroll 8
    for (int b = grp * per; b < b1; ++b) s += a.partial_s[(size_t)b * SS + e];
  }
  red[grp][tx] = s;
  __syncthreads();
  if (grp != 0 || !valid) return;
  if (a.layer0_mfma && e < (size_t)F * a.k0pad + F) return;      // first layer: reduced by k_reduce_w / k_reduce_b from k_wgrad_bf16's partials
  s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  const size_t wout = (size_t)F * a.k0 + F + (size_t)a.n_hidden * ((size_t)F * F + F);
  size_t dst;
  if (e < (size_t)F * a.k0pad) {
    const int row = (int)(e / a.k0pad), c = (int)(e % a.k0pad);
    if (c >= a.k0) return;
    dst = (size_t)row * a.k0 + c;
  } else if (e < (size_t)F * a.k0pad + F) dst = (size_t)F * a.k0 + (e - (size_t)F * a.k0pad);
  else dst = wout + (e - (size_t)F * a.k0pad - F);       // F output weights, then the output bias
  a.grad[dst] += s;
}
template <int F>
__global__ void __launch_bounds__(256) k_reduce_small(const ReduceArgs a) { reduce_small_body<F>(a, blockIdx.x, threadIdx.x, threadIdx.y); }
// The three reductions of the 8-bit-stash path in ONE launch (a training iteration of the reference's batch is a string of launch-latency-sized
// kernels): blocks [0, nw) sum the hidden layers' weight partials (k_reduce_w's grid, x-major), the next nb the bias partials, the rest the
// first-/output-layer group records.  Block = 256 threads; each block runs exactly one of the three bodies.
template <int F>
__global__ void __launch_bounds__(256) k_reduce_all(const ReduceArgs a, int nwx, int nw, int nb) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (b < nw) reduce_w_body<F>(a, b % nwx, b / nwx, t);
  else if (b < nw + nb) { if (t < F) reduce_b_body<F>(a, b - nw, t); }
  else reduce_small_body<F>(a, b - nw - nb, t & 63, t >> 6);
}

// Fourier coefficients' gradient from k_wgrad_bf16's coefficient pass (slot N+1 of `partial`: G[f][c] = sum_n dZ_0[n][f] D[n][c], D =
// d(enc)/d(coef)/(2 pi), carrying the same power-of-two scale as the hidden layers in f16 mode):
//   d_coef[m] += 2 pi * sum_f ( W_0[f][3+m] G[f][m] + W_0[f][3+nb+m] G[f][nb+m] ).   grid = nb blocks of F threads; fixed order.
template <int F>
__global__ void __launch_bounds__(F) k_reduce_coef(const ReduceArgs a) {
  __shared__ float red[F];
  const int m = blockIdx.x, f = threadIdx.x, nb = a.coef_cols;
  float g1 = 0.f, g2 = 0.f;
  for (int sp = 0; sp < a.n_splits; ++sp) {
    const float* P = a.partial + ((size_t)(a.n_hidden + 1) * a.n_splits + sp) * F * F + (size_t)f * a.k0pad;
    g1 += P[m];
    g2 += P[nb + m];
  }
  float v = fmaf(a.w0[(size_t)f * a.k0 + 3 + m], g1, a.w0[(size_t)f * a.k0 + 3 + nb + m] * g2) * 6.283185307179586f;
  if (a.gmax) v *= ldexpf(1.f, wgrad_scale_exp(a.gmax) - a.scale_shift);
  red[f] = v;
  __syncthreads();
  for (int s = F / 2; s >= 1; s >>= 1) {
    if (f < s) red[f] += red[f + s];
    __syncthreads();
  }
  if (f == 0) a.d_coef[m] += red[0];
}

#ifndef AFX_TEMPLATES_ONLY
// ---------------------------------------------------------------------------------------
// Weight re-tiling for the bf16 kernels.  parts = 2 also stores the lo parts (X3 forward) as a second stream.
// ---------------------------------------------------------------------------------------
struct PrepArgs16 {
  const float* params;
  char* prepared;
  int32_t F, n_hidden, k0, nk0, parts;
  uint32_t slab0_off, slab0_bytes, fwd_off, slabh_stride, bwd_off, slabt_bytes, lo_off;
  int32_t h16;          // hidden slabs (forward and transposed) in f16; the first layer stays split bf16
};

__device__ __forceinline__ unsigned short bf16_rne(float x) { return (unsigned short)(pack2(x, 0.f) & 0xffffu); }

__device__ __forceinline__ void prepare_bf16_body(const PrepArgs16& p, int64_t gid, int64_t gsz) {
  const int F = p.F, NT = F / 32, N = p.n_hidden;
  const size_t hidden0 = (size_t)F * p.k0 + F;
  const size_t stride = (size_t)F * F + F;
  // layer-0 slabs: slab t, element [((q*2 + part)*64 + lane)*8 + j] = W0[32t + (lane&31)][16q + 8(lane>>5) + j]
  {
    unsigned short* s0 = (unsigned short*)(p.prepared + p.slab0_off);
    const int64_t per = p.slab0_bytes / 2;
    for (int64_t i = gid; i < per * NT; i += gsz) {
      const int t = (int)(i / per);
      const int e = (int)(i % per);
      const int j = e & 7, lane = (e >> 3) & 63, qp = e >> 9;
      const int part = qp & 1, q = qp >> 1;
      const int k = 16 * q + 8 * (lane >> 5) + j;
      float w = 0.f;
      if (q < p.nk0 && k < p.k0) w = p.params[(size_t)(32 * t + (lane & 31)) * p.k0 + k];
      const unsigned short hi = bf16_rne(w);
      s0[i] = part == 0 ? hi : bf16_rne(w - __builtin_bit_cast(float, (unsigned)hi << 16));
    }
  }
  // hidden slabs: element [((part*2NT + u)*64 + lane)*8 + j], u = 2t' + s,
  //   k = 32t' + 16s + 8(j>>2) + 4(lane>>5) + (j&3), r = 32t + (lane&31)
  //   fwd: W_l[r][k] (parts hi[,lo]);  bwd: W_l[k][r] (hi only)
  const int64_t per_part = (int64_t)2 * NT * 64 * 8;
  const int64_t nslab = (int64_t)N * NT;
  const int64_t nfwd = nslab * p.parts * per_part;
  for (int64_t i = gid; i < nfwd + nslab * per_part; i += gsz) {
    const bool isb = i >= nfwd;
    const int64_t ii = isb ? i - nfwd : i;
    const int64_t per_slab = isb ? per_part : p.parts * per_part;
    const int64_t slab = ii / per_slab;
    const int64_t e0 = ii % per_slab;
    const int part = (int)(e0 / per_part);
    const int e = (int)(e0 % per_part);
    const int j = e & 7, lane = (e >> 3) & 63, u = e >> 9;
    const int tp = u >> 1, s = u & 1;
    const int t = (int)(slab % NT), lidx = (int)(slab / NT);
    const int l = isb ? N - lidx : 1 + lidx;
    const float* W = p.params + hidden0 + (size_t)(l - 1) * stride;
    const int r = 32 * t + (lane & 31);
    const int k = 32 * tp + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
    const float w = isb ? W[(size_t)k * F + r] : W[(size_t)r * F + k];
    const unsigned short hi = bf16_rne(w);
    unsigned short val = part == 0 ? hi : bf16_rne(w - __builtin_bit_cast(float, (unsigned)hi << 16));
    if (p.h16) val = __builtin_bit_cast(unsigned short, (_Float16)w);
    unsigned short* dst = isb ? (unsigned short*)(p.prepared + p.bwd_off + slab * p.slabt_bytes)
                              : (unsigned short*)(p.prepared + (part == 0 ? p.fwd_off : p.lo_off) + slab * p.slabt_bytes);
    dst[e] = val;
  }
}
__global__ void k_prepare_bf16(const PrepArgs16 p) { prepare_bf16_body(p, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x); }
// both re-tilings of a 16-bit precision in one launch (the weights are re-tiled after every optimizer step: two launch latencies per training
// iteration became one): the first half of the grid writes the fp32 `small` section, the second half the 16-bit slabs
__global__ void k_prepare_both(const PrepArgs p, const PrepArgs16 q) {
  const int half = gridDim.x / 2;
  if ((int)blockIdx.x < half) prepare_f32_body(p, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)half * blockDim.x);
  else prepare_bf16_body(q, (int64_t)(blockIdx.x - half) * blockDim.x + threadIdx.x, (int64_t)half * blockDim.x);
}
#endif  // AFX_TEMPLATES_ONLY
