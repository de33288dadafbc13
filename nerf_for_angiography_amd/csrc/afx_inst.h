// afx_inst.h — the instantiation list of the 16-bit chain kernels, shared by the translation units that define them
// (afx_inst_chain16.hip, compiled once per width and direction, in parallel) and the host code that launches them.
// X(F, X3, ENC, BWD, NW, SG, H16, S8)
#pragma once
#define AFX_CHAIN16_FWD(X, F)          \
  X(F, true, false, false, 4, false, false, false)  \
  X(F, true, true, false, 4, false, false, false)   \
  X(F, false, false, false, 8, false, false, false) \
  X(F, false, true, false, 8, false, false, false)  \
  X(F, false, false, false, 8, false, true, false)  \
  X(F, false, true, false, 8, false, true, false)
#define AFX_CHAIN16_BWD(X, F)          \
  X(F, false, false, true, 8, false, false, false)  \
  X(F, false, true, true, 8, false, false, false)   \
  X(F, false, false, true, 8, true, false, false)   \
  X(F, false, false, true, 8, false, true, false)   \
  X(F, false, true, true, 8, false, true, false)    \
  X(F, false, false, true, 8, true, true, false)    \
  X(F, false, false, true, 8, true, true, true)    \
  X(F, false, true, true, 8, true, true, true)
#define AFX_CHAIN16_DECL(F, X3, ENC, BWD, NW, SG, H16, S8) \
  extern template __global__ void k_chain_bf16<F, X3, ENC, BWD, NW, SG, H16, S8>(const afx::ChainArgs);
#define AFX_CHAIN16_DEF(F, X3, ENC, BWD, NW, SG, H16, S8) \
  template __global__ void k_chain_bf16<F, X3, ENC, BWD, NW, SG, H16, S8>(const afx::ChainArgs);
// forward-only kernels for tanh / sine models: split bf16, bf16, f16
#define AFX_CHAIN16_ACTS(X, F) X(F, true, 4, false) X(F, false, 8, false) X(F, false, 8, true)
#define AFX_CHAIN16_ACT_DECL(F, X3, NW, H16) \
  extern template __global__ void k_chain_bf16<F, X3, false, false, NW, false, H16, false, 0, 1>(const afx::ChainArgs);
#define AFX_CHAIN16_ACT_DEF(F, X3, NW, H16) \
  template __global__ void k_chain_bf16<F, X3, false, false, NW, false, H16, false, 0, 1>(const afx::ChainArgs);
// split phases of the 8-bit-stash training kernel (PHASE 1 = forward half, 2 = backward half)
#define AFX_CHAIN16_PHASES(X, F) X(F, false, 1) X(F, false, 2) X(F, true, 1) X(F, true, 2)
#define AFX_CHAIN16_PH_DECL(F, ENC, PH) \
  extern template __global__ void k_chain_bf16<F, false, ENC, true, 8, true, true, true, PH>(const afx::ChainArgs);
#define AFX_CHAIN16_PH_DEF(F, ENC, PH) \
  template __global__ void k_chain_bf16<F, false, ENC, true, 8, true, true, true, PH>(const afx::ChainArgs);
