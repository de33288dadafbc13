// afx_api.hip — host side of the C-ABI declared in include/afx.h.
// Validates arguments, carves the caller-owned workspace, launches the gfx950 kernels.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/afx.h"
#include "afx_internal.h"
#include "afx_kernels_f32.hip"
#include "afx_kernels_bf16.hip"
#include "afx_kernels_grid.hip"
#include "afx_inst.h"
#include <algorithm>
#include <set>

// The 16-bit chain kernels are compiled in their own translation units (afx_inst_chain16.hip); -DAFX_SINGLE_TU
// instantiates them here instead (diagnostic builds such as -DAFX_STAMP, whose device symbol must be unique).
#ifdef AFX_SINGLE_TU
#define AFX_CHAIN16_HERE AFX_CHAIN16_DEF
#else
#define AFX_CHAIN16_HERE AFX_CHAIN16_DECL
#endif
AFX_CHAIN16_FWD(AFX_CHAIN16_HERE, 64) AFX_CHAIN16_BWD(AFX_CHAIN16_HERE, 64)
AFX_CHAIN16_FWD(AFX_CHAIN16_HERE, 128) AFX_CHAIN16_BWD(AFX_CHAIN16_HERE, 128)
AFX_CHAIN16_FWD(AFX_CHAIN16_HERE, 256) AFX_CHAIN16_BWD(AFX_CHAIN16_HERE, 256)
#ifdef AFX_SINGLE_TU
#define AFX_CHAIN16_PH_HERE AFX_CHAIN16_PH_DEF
#define AFX_CHAIN16_ACT_HERE AFX_CHAIN16_ACT_DEF
#else
#define AFX_CHAIN16_PH_HERE AFX_CHAIN16_PH_DECL
#define AFX_CHAIN16_ACT_HERE AFX_CHAIN16_ACT_DECL
#endif
AFX_CHAIN16_ACTS(AFX_CHAIN16_ACT_HERE, 64) AFX_CHAIN16_ACTS(AFX_CHAIN16_ACT_HERE, 128) AFX_CHAIN16_ACTS(AFX_CHAIN16_ACT_HERE, 256)
AFX_CHAIN16_PHASES(AFX_CHAIN16_PH_HERE, 64) AFX_CHAIN16_PHASES(AFX_CHAIN16_PH_HERE, 128) AFX_CHAIN16_PHASES(AFX_CHAIN16_PH_HERE, 256)

using namespace afx;

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) return fail(AFX_E_HIP, "%s -> %s (%d)", #x, hipGetErrorString(e_), (int)e_); \
  } while (0)

struct ProfRec { hipEvent_t a, b; int which; };
struct afx_ctx {
  afx_model_desc d;
  int k0, nq, k0pad, nt;
  int64_t n_params;
  int n_cu;
  std::set<const void*> attr_done;
  bool profiling;
  hipStream_t side = nullptr;          // overlap mode: weight-gradient kernels run here (fork/join with events)
  hipEvent_t ev_chain[2] = {nullptr, nullptr}, ev_wgrad[2] = {nullptr, nullptr};
  int overlap, persistent_chain;
  int small_in_kernel;   // first-/output-layer gradient sums inside the backward chain kernel (AFX_SMALL_IN_KERNEL=0: off)
  int force_split;       // AFX_FORCE_SPLIT=1 (measurement): the split-phase training step also where the fused kernel applies
  int device;            // the HIP device this context was created on; every entry point checks it is current
  const float* coef_params = nullptr;   // afx_set_encoding_grad: fp32 flat parameters (W_0 is read from them) ...
  float* d_coef = nullptr;              // ... and where d loss / d fourier coefficients accumulates (null: coefficients are constants)
  std::vector<ProfRec> recs;
  int64_t* mailbox = nullptr;           // afx_march_train_step_mse: 2 x 4 host-mapped words the offsets kernel posts its totals to (lazily allocated)
  int64_t* mailbox_dev = nullptr;
  int64_t mail_seq = 0;
};

// Optional HIP-event bracket around one kernel launch, on the launch stream (bench.py's roofline leg).
struct ProfScope {
  afx_ctx* c; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; int which;
  ProfScope(afx_ctx* c_, int which_, hipStream_t st_) : c(c_), st(st_), which(which_) {
    // bounded: a caller that enables profiling and never reads it back stops accumulating events after 64 Ki launches
    if (c->profiling && c->recs.size() < 65536 && hipEventCreate(&a) == hipSuccess) {
      if (hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, st);
      else { (void)hipEventDestroy(a); a = b = nullptr; }
    } else a = b = nullptr;
  }
  ~ProfScope() {
    if (a && b) { (void)hipEventRecord(b, st); c->recs.push_back({a, b, which}); }
  }
};

static inline uint32_t rup(uint64_t v, uint64_t a) { return (uint32_t)((v + a - 1) / a * a); }
static inline size_t rup64(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Layout of the prepared-weights buffer for one precision.
struct PrepLayout {
  uint32_t small_floats, small_bytes_pad, slab0_bytes, slabh_stride, slabt_bytes;
  uint32_t small_off, slab0_off, fwd_off, bwd_off, lo_off;
  int n_slab0;          // slabs of the first layer in the forward stream (1 for f32, NT for bf16)
  size_t total;
};

static inline bool is_f16(int prec) { return prec == AFX_PREC_F16 || prec == AFX_PREC_F16S8; }
static inline bool is_bf16(int prec) { return prec == AFX_PREC_BF16 || prec == AFX_PREC_BF16X3 || is_f16(prec); }   // the 16-bit kernel family
static inline int nk0_of(const afx_ctx* c) { return c->d.enc == AFX_ENC_NONE ? 1 : 4; }
// samples per workgroup tile of the forward / backward chain kernel
static inline int fwd_tile(int prec) { return (prec == AFX_PREC_BF16 || is_f16(prec)) ? 256 : 128; }
static inline int bwd_tile(int prec) { return prec == AFX_PREC_F32 ? 128 : 256; }

static PrepLayout prep_layout(const afx_ctx* c, int prec) {
  PrepLayout L;
  const int F = c->d.width, N = c->d.n_hidden, NT = c->nt;
  const int naux = c->d.enc == AFX_ENC_BARF ? 6 * c->d.n_freq : (c->d.enc == AFX_ENC_FOURIER ? 3 * c->d.n_freq : 0);
  L.small_floats = rup((uint64_t)(N + 2) * F + 4 + naux, 4);
  L.small_bytes_pad = rup((uint64_t)L.small_floats * 4, 1024);
  L.small_off = 0;
  L.slab0_off = rup((uint64_t)L.small_floats * 4, 4096);
  if (prec == AFX_PREC_F32) {
    L.n_slab0 = 1;
    L.slab0_bytes = rup((uint64_t)c->nq * NT * 256, 4096);
    L.slabh_stride = (uint32_t)NT * 4 * 1024;
    L.slabt_bytes = L.slabh_stride;
  } else {
    L.n_slab0 = NT;
    L.slab0_bytes = chain_slab0_bytes(nk0_of(c));
    L.slabt_bytes = (uint32_t)NT * 2 * 1024;
    L.slabh_stride = L.slabt_bytes;
  }
  // [first-layer slabs | forward hidden slabs | transposed slabs] is one contiguous stream in consumption order
  L.fwd_off = L.slab0_off + (uint32_t)L.n_slab0 * L.slab0_bytes;
  L.bwd_off = L.fwd_off + (uint32_t)N * NT * L.slabh_stride;
  L.lo_off = L.bwd_off + (uint32_t)N * NT * L.slabt_bytes;         // split-bf16: lo parts of the forward hidden slabs
  L.total = (size_t)L.lo_off + (prec == AFX_PREC_BF16X3 ? (size_t)N * NT * L.slabt_bytes : 0);
  return L;
}

extern "C" const char* afx_last_error(void) { return g_err.c_str(); }

extern "C" int afx_create(const afx_model_desc* d, afx_ctx** out) {
  if (!d || !out) return fail(AFX_E_INVALID, "afx_create: null argument");
  if (d->n_in != 3) return fail(AFX_E_INVALID, "afx_create: n_in must be 3 (got %d)", d->n_in);
  if (d->width != 64 && d->width != 128 && d->width != 256)
    return fail(AFX_E_INVALID, "afx_create: width must be 64, 128 or 256 (got %d)", d->width);
  if (d->n_hidden < 1 || d->n_hidden > 16) return fail(AFX_E_INVALID, "afx_create: n_hidden must be in 1..16 (got %d)", d->n_hidden);
  if (d->enc < AFX_ENC_NONE || d->enc > AFX_ENC_FOURIER) return fail(AFX_E_INVALID, "afx_create: bad enc %d", d->enc);
  if (d->enc != AFX_ENC_NONE && (d->n_freq < 1 || d->n_freq > 10))
    return fail(AFX_E_INVALID, "afx_create: n_freq must be in 1..10 (got %d)", d->n_freq);
  if (d->act < AFX_ACT_RELU || d->act > AFX_ACT_SINE) return fail(AFX_E_INVALID, "afx_create: bad act %d", d->act);
  if (d->act != AFX_ACT_RELU && d->enc != AFX_ENC_NONE) return fail(AFX_E_INVALID, "afx_create: tanh / sine models are taken without an input encoding only");
  afx_ctx* c = new afx_ctx();
  c->d = *d;
  if (d->act != AFX_ACT_SINE) c->d.act_w0 = 1.f;
  if (d->enc == AFX_ENC_NONE) c->d.n_freq = 0;
  c->k0 = 3 + 6 * c->d.n_freq;
  c->nq = (c->k0 + 1) / 2;
  c->k0pad = 2 * c->nq;
  c->nt = d->width / 32;
  if (c->k0pad > d->width) { delete c; return fail(AFX_E_INVALID, "afx_create: encoded width %d exceeds layer width", c->k0pad); }
  const int64_t F = d->width;
  c->n_params = F * c->k0 + F + (int64_t)d->n_hidden * (F * F + F) + F + 1;
  int dev = -1;
  hipDeviceProp_t prop;
  c->n_cu = 256;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    c->n_cu = prop.multiProcessorCount;
  else dev = -1;         // no GPU: queries and validation still work, launches fail with a HIP error
  c->device = dev;
  c->profiling = false;
  c->overlap = 0; c->persistent_chain = 0;     // AFX_OVERLAP=1: chain(i+1) || wgrad(i) on two streams (+2.6 % on the 512^2x128 step; per-kernel times inflate)
  c->small_in_kernel = 1;
  // environment knobs are read ONCE, here (never on the launch path)
  if (const char* e = getenv("AFX_OVERLAP")) c->overlap = atoi(e);
  if (const char* e = getenv("AFX_PERSISTENT")) c->persistent_chain = atoi(e);
  if (const char* e = getenv("AFX_SMALL_IN_KERNEL")) c->small_in_kernel = atoi(e) != 0;
  c->force_split = 0;
  if (const char* e = getenv("AFX_FORCE_SPLIT")) c->force_split = atoi(e) != 0;
  *out = c;
  return AFX_OK;
}

extern "C" void afx_destroy(afx_ctx* c) {
  if (!c) return;
#ifdef AFX_STAMP
  {
    unsigned long long h[3][8][10];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)) == hipSuccess) {
      static const char* nm[10] = {"vmcnt wait", "barrier", "request", "fwd mfma", "fwd epilogue", "bwd mfma", "bwd epilogue", "other", "tile prologue", "output/sums/masks"};
      for (int p = 0; p < 3; ++p)
        for (int w = 0; w < 8; ++w) {
          unsigned long long tot = 0;
          for (int i = 0; i < 10; ++i) tot += h[p][w][i];
          if (!tot) continue;
          fprintf(stderr, "[stamps] phase %d wave %d total %llu:", p, w, tot);
          for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * h[p][w][i] / tot);
          fprintf(stderr, "\n");
        }
    }
  }
#endif
  for (auto& r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : c->ev_chain) if (e) (void)hipEventDestroy(e);
  for (auto e : c->ev_wgrad) if (e) (void)hipEventDestroy(e);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->mailbox) (void)hipHostFree(c->mailbox);
  delete c;
}

extern "C" int afx_set_encoding_grad(afx_ctx* c, const float* params, float* d_enc_aux) {
  if (!c) return fail(AFX_E_INVALID, "afx_set_encoding_grad: null ctx");
  if (d_enc_aux && c->d.enc != AFX_ENC_FOURIER)
    return fail(AFX_E_INVALID, "afx_set_encoding_grad: only the fourier encoding has trainable coefficients");
  if (d_enc_aux && !params) return fail(AFX_E_INVALID, "afx_set_encoding_grad: params required");
  c->coef_params = d_enc_aux ? params : nullptr;
  c->d_coef = d_enc_aux;
  return AFX_OK;
}

extern "C" int afx_profile_enable(afx_ctx* c, int on) {
  if (!c) return fail(AFX_E_INVALID, "afx_profile_enable: null ctx");
  c->profiling = on != 0;
  return AFX_OK;
}

extern "C" int afx_profile_read(afx_ctx* c, int which, double* ms_total, int64_t* launches) {
  if (!c || !ms_total || !launches) return fail(AFX_E_INVALID, "afx_profile_read: null argument");
  double tot = 0.0;
  int64_t n = 0;
  std::vector<ProfRec> keep, mine;
  for (auto& r : c->recs) (r.which == which ? mine : keep).push_back(r);
  c->recs.swap(keep);          // from here on every record of `which` is owned (and destroyed exactly once) by this call
  hipError_t err = hipSuccess;
  for (auto& r : mine) {
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventSynchronize(r.b);
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, r.a, r.b);
    if (err == hipSuccess) { tot += ms; ++n; }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  if (err != hipSuccess) return fail(AFX_E_HIP, "afx_profile_read: %s (%d)", hipGetErrorString(err), (int)err);
  *ms_total = tot; *launches = n;
  return AFX_OK;
}

extern "C" int afx_param_layout(const afx_ctx* c, int layer, int64_t* w_off, int64_t* b_off, int32_t* rows, int32_t* cols) {
  if (!c) return fail(AFX_E_INVALID, "afx_param_layout: null ctx");
  const int64_t F = c->d.width;
  const int N = c->d.n_hidden;
  if (layer < 0 || layer > N + 1) return fail(AFX_E_INVALID, "afx_param_layout: layer %d out of range", layer);
  int64_t wo, bo;
  int r, cc;
  if (layer == 0) { wo = 0; bo = F * c->k0; r = (int)F; cc = c->k0; }
  else if (layer <= N) { wo = F * c->k0 + F + (int64_t)(layer - 1) * (F * F + F); bo = wo + F * F; r = (int)F; cc = (int)F; }
  else { wo = F * c->k0 + F + (int64_t)N * (F * F + F); bo = wo + F; r = 1; cc = (int)F; }
  if (w_off) *w_off = wo;
  if (b_off) *b_off = bo;
  if (rows) *rows = r;
  if (cols) *cols = cc;
  return AFX_OK;
}

// Backward workspace: fixed part + per-tile part.
struct BwdLayout {
  size_t fixed_bytes, per_tile_bytes;
};
static const int kSplits = 64;
static const int kSmallBlocks = 1024;  // blocks (and partial records) of k_small_grads_bf16 / k_small_from_groups: 4 per CU, their record loop is latency-bound
// per-tile bytes of the 8-bit-stash mode: 1-byte H_l / dZ'_l planes, the input stash, per-sample words (group exponents / g'), and the
// tile's ReLU-mask image (split phases): (N+1) layers x NT tiles x 512 lanes x 2 B
static size_t per_tile_s8(const afx_ctx* c) {
  return (size_t)256 * (2 * ((size_t)c->d.n_hidden + 1) * c->d.width + 4 * 16 * nk0_of(c) + 4 + 4) + ((size_t)c->d.n_hidden + 1) * c->nt * 1024
         + (size_t)c->d.n_hidden * 32;      // H block scales: one dword per layer and 32-sample group
}
static BwdLayout bwd_layout(const afx_ctx* c, int prec, int64_t n_rays, int64_t groups_per_ray = 0) {
  const size_t F = c->d.width, N = c->d.n_hidden;
  BwdLayout B;
  size_t fixed = rup64((size_t)n_rays * 4, 256);   // dod (rays mode only)
  fixed += rup64((size_t)n_rays * (size_t)groups_per_ray * 4, 256);      // optical-depth partials of the 32-sample groups (split training step)
  fixed += rup64((N + 2) * (size_t)kSplits * F * F * 4, 256);     // partial (slot N+1: the fourier-coefficient contraction)
  fixed += rup64((N + 2) * (size_t)kSplits * (F + 4) * 4, 256);   // partial2
  if (prec != AFX_PREC_F32) fixed += rup64((size_t)kSmallBlocks * (F * 16 * nk0_of(c) + 2 * F + 4) * 4, 256);   // partial_s
  fixed += 256;                                                     // gmax words (f16 mode), one per stash buffer
  B.fixed_bytes = fixed;
  if (prec == AFX_PREC_F32) B.per_tile_bytes = (size_t)128 * 4 * (2 * (N + 1) * F + c->k0pad + 1);
  else B.per_tile_bytes = (size_t)256 * (2 * (N + 1) * F * 2 + 4 * 16 * nk0_of(c) + 4);
  return B;
}

static int64_t s_pad_of(int s) { return (int64_t)(s + GROUP - 1) / GROUP * GROUP; }

extern "C" int64_t afx_query(const afx_ctx* c, int what, int64_t a0, int64_t a1, int64_t a2) {
  if (!c) { fail(AFX_E_INVALID, "afx_query: null ctx"); return -1; }
  switch (what) {
    case AFX_Q_PARAM_COUNT: return c->n_params;
    case AFX_Q_K0: return c->k0;
    case AFX_Q_PREPARED_BYTES: return (int64_t)prep_layout(c, (int)a0).total;
    case AFX_Q_FWD_WORKSPACE: return (int64_t)rup64((size_t)a0 * (size_t)(s_pad_of((int)a1) / GROUP) * 4, 256);
    case AFX_Q_BWD_WORKSPACE_MIN: {
      BwdLayout B = bwd_layout(c, (int)a2, a0, a1 > 0 ? s_pad_of((int)a1) / GROUP : 0);
      return (int64_t)(B.fixed_bytes + 32 * B.per_tile_bytes + 1024);
    }
    case AFX_Q_BWD_WORKSPACE_FULL: {
      BwdLayout B = bwd_layout(c, (int)a2, a0, a0 > 0 ? s_pad_of((int)a1) / GROUP : 0);
      const int64_t samples = a0 > 0 ? a0 * s_pad_of((int)a1) : a1;
      int64_t tiles = (samples + bwd_tile((int)a2) - 1) / bwd_tile((int)a2);
      if (is_bf16((int)a2)) {      // a chunk never exceeds the 4 GiB layer plane the 32-bit stash offsets reach (run_backward): more is never used
        const uint64_t plane_rows = ((uint64_t)1 << 32) / ((uint64_t)c->d.width * 2);
        size_t need = (size_t)std::min<int64_t>(tiles, (int64_t)(plane_rows / bwd_tile((int)a2))) * B.per_tile_bytes;
        if ((int)a2 == AFX_PREC_F16S8 && c->small_in_kernel) {      // rays mode stashes 1-byte elements: twice the rows per plane, ~half the bytes per row
          need = std::max(need, (size_t)std::min<int64_t>(tiles, (int64_t)(2 * plane_rows / 256)) * per_tile_s8(c));
        }
        return (int64_t)(B.fixed_bytes + need + 1024);
      }
      return (int64_t)(B.fixed_bytes + (size_t)tiles * B.per_tile_bytes + 1024);
    }
  }
  fail(AFX_E_INVALID, "afx_query: unknown query %d", what);
  return -1;
}

// Launches go to the stream the caller passes, which belongs to the caller's CURRENT device: the context's CU count,
// function attributes and workspace sizing are those of the device it was created on, so the two must agree.
static int check_dev(const afx_ctx* c, const char* who) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return fail(AFX_E_HIP, "%s: no HIP device", who);
  if (dev != c->device) return fail(AFX_E_INVALID, "%s: context belongs to device %d but device %d is current", who, c->device, dev);
  return AFX_OK;
}

static int check_prec(int prec, const char* who) {
  if (prec != AFX_PREC_F32 && prec != AFX_PREC_BF16X3 && prec != AFX_PREC_BF16 && !is_f16(prec)) return fail(AFX_E_INVALID, "%s: unknown precision %d", who, prec);
  return AFX_OK;
}

extern "C" int afx_prepare_weights(afx_ctx* c, int prec, const float* params, const float* enc_aux, void* prepared,
                                   size_t prepared_bytes, void* stream) {
  if (!c || !params || !prepared) return fail(AFX_E_INVALID, "afx_prepare_weights: null argument");
  if (check_prec(prec, "afx_prepare_weights")) return AFX_E_INVALID;
  if (c->d.enc != AFX_ENC_NONE && !enc_aux) return fail(AFX_E_INVALID, "afx_prepare_weights: enc_aux required for this encoding");
  const PrepLayout L = prep_layout(c, prec);
  if (prepared_bytes < L.total) return fail(AFX_E_WORKSPACE, "afx_prepare_weights: prepared buffer %zu < %zu bytes", prepared_bytes, L.total);
  if (int rc = check_dev(c, "afx_prepare_weights")) return rc;
  PrepArgs p = {};
  p.params = params; p.enc_aux = enc_aux; p.prepared = (char*)prepared;
  p.F = c->d.width; p.n_hidden = c->d.n_hidden; p.k0 = c->k0; p.nq = c->nq; p.enc = c->d.enc; p.n_freq = c->d.n_freq;
  p.small_off = L.small_off; p.slab0_off = L.slab0_off; p.fwd_off = L.fwd_off; p.bwd_off = L.bwd_off;
  p.slab0_bytes = L.slab0_bytes; p.slabh_bytes = L.slabh_stride; p.small_floats = L.small_floats;
  p.weights = prec == AFX_PREC_F32 ? 1 : 0;
  if (!is_bf16(prec)) hipLaunchKernelGGL(k_prepare_f32, dim3(512), dim3(256), 0, (hipStream_t)stream, p);
  else {
    PrepArgs16 q = {};
    q.params = params; q.prepared = (char*)prepared;
    q.F = c->d.width; q.n_hidden = c->d.n_hidden; q.k0 = c->k0; q.nk0 = nk0_of(c); q.parts = prec == AFX_PREC_BF16X3 ? 2 : 1;
    q.slab0_off = L.slab0_off; q.slab0_bytes = L.slab0_bytes; q.fwd_off = L.fwd_off; q.slabh_stride = L.slabh_stride;
    q.bwd_off = L.bwd_off; q.slabt_bytes = L.slabt_bytes; q.lo_off = L.lo_off; q.h16 = is_f16(prec) ? 1 : 0;
    hipLaunchKernelGGL(k_prepare_both, dim3(1024), dim3(256), 0, (hipStream_t)stream, p, q);
  }
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

template <class K>
static int launch_chain_k(afx_ctx* c, K kern, int which, const ChainArgs& a, size_t lds_bytes, int grid, hipStream_t st, int threads = 256) {
  if (!c->attr_done.count((const void*)kern)) {
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->attr_done.insert((const void*)kern);
  }
  {
    ProfScope ps(c, which, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, st, a);
  }
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

template <int F>
static int launch_chain_f(afx_ctx* c, int prec, bool bwd, const ChainArgs& a, size_t lds, int grid, hipStream_t st, int phase) {
  const int which = bwd ? AFX_K_CHAIN_BWD : AFX_K_CHAIN_FWD;
  const bool enc = c->d.enc != AFX_ENC_NONE;
  if (!bwd && a.act != 0 && prec != AFX_PREC_F32) {      // tanh / sine: the forward-only kernels with the activation epilogue
    if (prec == AFX_PREC_BF16X3) return launch_chain_k(c, k_chain_bf16<F, true, false, false, 4, false, false, false, 0, 1>, which, a, lds, grid, st);
    if (is_f16(prec)) return launch_chain_k(c, k_chain_bf16<F, false, false, false, 8, false, true, false, 0, 1>, which, a, lds, grid, st, 512);
    return launch_chain_k(c, k_chain_bf16<F, false, false, false, 8, false, false, false, 0, 1>, which, a, lds, grid, st, 512);
  }
  if (phase == 1) return enc ? launch_chain_k(c, k_chain_bf16<F, false, true, true, 8, true, true, true, 1>, which, a, lds, grid, st, 512)
                             : launch_chain_k(c, k_chain_bf16<F, false, false, true, 8, true, true, true, 1>, which, a, lds, grid, st, 512);
  if (phase == 2) return enc ? launch_chain_k(c, k_chain_bf16<F, false, true, true, 8, true, true, true, 2>, which, a, lds, grid, st, 512)
                             : launch_chain_k(c, k_chain_bf16<F, false, false, true, 8, true, true, true, 2>, which, a, lds, grid, st, 512);
  if (prec == AFX_PREC_F32)
    return bwd ? launch_chain_k(c, k_chain_f32<F, true>, which, a, lds, grid, st) : launch_chain_k(c, k_chain_f32<F, false>, which, a, lds, grid, st);
  if (prec == AFX_PREC_BF16X3 && !bwd)
    return enc ? launch_chain_k(c, k_chain_bf16<F, true, true, false, 4>, which, a, lds, grid, st)
               : launch_chain_k(c, k_chain_bf16<F, true, false, false, 4>, which, a, lds, grid, st);
  if (is_f16(prec)) {
    if (bwd && a.small_part && a.stash8)
      return enc ? launch_chain_k(c, k_chain_bf16<F, false, true, true, 8, true, true, true>, which, a, lds, grid, st, 512)
                 : launch_chain_k(c, k_chain_bf16<F, false, false, true, 8, true, true, true>, which, a, lds, grid, st, 512);
    if (bwd && a.small_part) return launch_chain_k(c, k_chain_bf16<F, false, false, true, 8, true, true>, which, a, lds, grid, st, 512);
    if (bwd)
      return enc ? launch_chain_k(c, k_chain_bf16<F, false, true, true, 8, false, true>, which, a, lds, grid, st, 512)
                 : launch_chain_k(c, k_chain_bf16<F, false, false, true, 8, false, true>, which, a, lds, grid, st, 512);
    return enc ? launch_chain_k(c, k_chain_bf16<F, false, true, false, 8, false, true>, which, a, lds, grid, st, 512)
               : launch_chain_k(c, k_chain_bf16<F, false, false, false, 8, false, true>, which, a, lds, grid, st, 512);
  }
  if (bwd && a.small_part) return launch_chain_k(c, k_chain_bf16<F, false, false, true, 8, true>, which, a, lds, grid, st, 512);
  if (bwd)
    return enc ? launch_chain_k(c, k_chain_bf16<F, false, true, true, 8>, which, a, lds, grid, st, 512)
               : launch_chain_k(c, k_chain_bf16<F, false, false, true, 8>, which, a, lds, grid, st, 512);
  return enc ? launch_chain_k(c, k_chain_bf16<F, false, true, false, 8>, which, a, lds, grid, st, 512)
             : launch_chain_k(c, k_chain_bf16<F, false, false, false, 8>, which, a, lds, grid, st, 512);
}

static int launch_chain(afx_ctx* c, int prec, bool bwd, const ChainArgs& a, hipStream_t st, int phase = 0) {
  const int F = c->d.width, N = c->d.n_hidden;
  const bool occ2 = chain_occ2(c->nt, phase);      // backward half at widths <= 128: two workgroups per CU, two-tile steps
  const size_t slot = occ2 ? chain_slot_bytes(c->nt, nk0_of(c), true, false, 2) : a.slot_bytes;
  size_t lds = (size_t)a.small_bytes_pad + (size_t)(is_bf16(prec) ? chain_ring(bwd) : 2) * slot;
  const int ncg = (is_bf16(prec) && (bwd || prec == AFX_PREC_BF16 || is_f16(prec))) ? 2 : 1;
  if (bwd) lds += (size_t)(N + 1) * ((c->nt + 1) / 2) * ncg * 256 * 4 + 256;   // ReLU masks + per-group optical depths
  if (lds > 160 * 1024) return fail(AFX_E_INVALID, "model needs %zu B of LDS (> 160 KiB)", lds);
  const int tiles = a.tile1 - a.tile0;
  if (tiles <= 0) return AFX_OK;
  if (int rc = check_dev(c, "afx chain launch")) return rc;
  const int wgs = occ2 ? 2 * c->n_cu : c->n_cu;
  const int grid = (tiles < wgs || !a.persistent) ? tiles : wgs;    // persistent: one workgroup per CU (occ2: two) loops over tiles
  if (F == 64) return launch_chain_f<64>(c, prec, bwd, a, lds, grid, st, phase);
  if (F == 128) return launch_chain_f<128>(c, prec, bwd, a, lds, grid, st, phase);
  return launch_chain_f<256>(c, prec, bwd, a, lds, grid, st, phase);
}

static void fill_model(const afx_ctx* c, int prec, bool bwd, const void* prepared, ChainArgs& a) {
  const PrepLayout L = prep_layout(c, prec);
  const char* base = (const char*)prepared;
  a.stream_fwd = base + L.slab0_off;
  a.stream_bwd = base + L.bwd_off;
  a.stream_lo = base + L.lo_off;
  a.small = (const float*)(base + L.small_off);
  a.small_floats = L.small_floats; a.small_bytes_pad = L.small_bytes_pad;
  a.slab0_bytes = L.slab0_bytes; a.slabh_stride = L.slabh_stride; a.slabt_bytes = L.slabt_bytes;
  a.slabh_bytes = L.slabh_stride;
  a.slot_bytes = a.slab0_bytes > a.slabh_bytes ? a.slab0_bytes : a.slabh_bytes;
  // the backward kernel of the split mode recomputes in plain bf16: it streams only the hi parts
  if (is_bf16(prec)) a.slot_bytes = chain_slot_bytes(c->nt, nk0_of(c), bwd, prec == AFX_PREC_BF16X3 && !bwd);
  a.n_hidden = c->d.n_hidden; a.k0 = c->k0; a.nq = c->nq; a.enc = c->d.enc; a.n_freq = c->d.n_freq;
  a.act = c->d.act; a.act_w0 = c->d.act_w0;
  a.persistent = 1;
}

extern "C" int afx_mlp_infer(afx_ctx* c, int prec, const void* prepared, const float* pts, int64_t n_pts, float* out,
                             int apply_sigmoid, void* stream) {
  if (!c || !prepared || !pts || !out) return fail(AFX_E_INVALID, "afx_mlp_infer: null argument");
  if (check_prec(prec, "afx_mlp_infer")) return AFX_E_INVALID;
  if (n_pts < 0 || n_pts > ((int64_t)1 << 31) - 256) return fail(AFX_E_INVALID, "afx_mlp_infer: n_pts must be < 2^31 per call");
  if (n_pts == 0) return AFX_OK;
  ChainArgs a = {};
  fill_model(c, prec, false, prepared, a);
  const int64_t tiles = (n_pts + fwd_tile(prec) - 1) / fwd_tile(prec);
  if (tiles > 0x7fffffff) return fail(AFX_E_INVALID, "afx_mlp_infer: too many points for one call");
  a.tile0 = 0; a.tile1 = (int)tiles; a.n_total = n_pts; a.mode = 0;
  a.pts = pts; a.out = out; a.apply_sigmoid = apply_sigmoid;
  return launch_chain(c, prec, false, a, (hipStream_t)stream);
}

static int check_render(const afx_ctx* c, const afx_render_args* r, const char* who) {
  if (!c || !r) return fail(AFX_E_INVALID, "%s: null argument", who);
  if (r->n_rays < 0) return fail(AFX_E_INVALID, "%s: n_rays < 0", who);
  if (r->n_samples < 2 || r->n_samples > 4096) return fail(AFX_E_INVALID, "%s: n_samples must be in 2..4096 (got %d)", who, r->n_samples);
  if (r->ray_mode == AFX_RAYS_ARRAYS) {
    if (!r->origins || !r->dirs) return fail(AFX_E_INVALID, "%s: origins/dirs required", who);
  } else if (r->ray_mode == AFX_RAYS_POSE) {
    if (!r->poses || r->width <= 0 || r->height <= 0 || !(r->focal > 0)) return fail(AFX_E_INVALID, "%s: poses/width/height/focal required", who);
  } else return fail(AFX_E_INVALID, "%s: bad ray_mode %d", who, r->ray_mode);
  if (r->depth_mode == AFX_DEPTH_UNIFORM_MID) {
    if (!(r->t_far > r->t_near)) return fail(AFX_E_INVALID, "%s: need t_far > t_near", who);
  } else if (r->depth_mode == AFX_DEPTH_SHARED_Z || r->depth_mode == AFX_DEPTH_PER_RAY_Z) {
    if (!r->z) return fail(AFX_E_INVALID, "%s: z required for this depth_mode", who);
  } else if (r->depth_mode == AFX_DEPTH_STRATIFIED) {
    if (!(r->t_far > r->t_near)) return fail(AFX_E_INVALID, "%s: need t_far > t_near", who);
  } else return fail(AFX_E_INVALID, "%s: bad depth_mode %d", who, r->depth_mode);
  if (!r->pixel) return fail(AFX_E_INVALID, "%s: pixel required", who);
  if (r->n_rays * s_pad_of(r->n_samples) > ((int64_t)1 << 31) - 256)
    return fail(AFX_E_INVALID, "%s: n_rays*padded samples must be < 2^31 per call; split the ray batch", who);
  return AFX_OK;
}

static void fill_render(const afx_render_args* r, ChainArgs& a) {
  a.mode = 1;
  a.org = r->origins; a.dir = r->dirs;
  a.poses = r->ray_mode == AFX_RAYS_POSE ? r->poses : nullptr;
  a.ray_ids = r->ray_ids; a.ray_id0 = r->ray_id0; a.width = r->width; a.height = r->height; a.focal = r->focal;
  a.n_samples = r->n_samples; a.s_pad = (int)s_pad_of(r->n_samples); a.depth_mode = r->depth_mode;
  a.t_near = r->t_near; a.t_far = r->t_far;
  a.jitter_seed = r->jitter_seed; a.jitter_stream = r->jitter_stream;
  a.t_step = (float)((double)(r->t_far - r->t_near) / r->n_samples);
  a.z = r->z;
  a.n_total = r->n_rays * a.s_pad;
  a.sigma = r->sigma; a.tau = r->tau;
}

extern "C" int afx_render_forward(afx_ctx* c, int prec, const void* prepared, const afx_render_args* r, void* stream) {
  if (c && r && r->n_rays == 0) return AFX_OK;
  int rc = check_render(c, r, "afx_render_forward");
  if (rc) return rc;
  if (check_prec(prec, "afx_render_forward")) return AFX_E_INVALID;
  if (!prepared) return fail(AFX_E_INVALID, "afx_render_forward: null prepared");
  if (r->n_rays == 0) return AFX_OK;
  const size_t need = (size_t)afx_query(c, AFX_Q_FWD_WORKSPACE, r->n_rays, r->n_samples, 0);
  if (!r->workspace || r->workspace_bytes < need) return fail(AFX_E_WORKSPACE, "afx_render_forward: workspace %zu < %zu bytes", r->workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  ChainArgs a = {};
  fill_model(c, prec, false, prepared, a);
  fill_render(r, a);
  a.od_part = (float*)r->workspace;
  a.tile0 = 0; a.tile1 = (int)((a.n_total + fwd_tile(prec) - 1) / fwd_tile(prec));
  rc = launch_chain(c, prec, false, a, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_finish_fwd, dim3((unsigned)((r->n_rays + 255) / 256)), dim3(256), 0, st, a.od_part, a.s_pad / GROUP, r->n_rays, r->pixel);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

template <int F>
static int launch_wgrad_t(afx_ctx* c, const WgradArgs& w, const ReduceArgs& rd, int N, hipStream_t st) {
  {
    const size_t lds = (size_t)2 * 2 * 32 * F * 4;      // 2 stages x (A + B chunk of 32 samples)
    if (!c->attr_done.count((const void*)k_wgrad_f32<F>)) {
      HIPCHK(hipFuncSetAttribute((const void*)k_wgrad_f32<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      c->attr_done.insert((const void*)k_wgrad_f32<F>);
    }
    ProfScope ps(c, AFX_K_WGRAD, st);
    hipLaunchKernelGGL(k_wgrad_f32<F>, dim3(w.n_splits, N + 1), dim3(512), lds, st, w);
  }
  hipLaunchKernelGGL(k_colsum_f32<F>, dim3(w.n_splits, N + 2), dim3(1024), 0, st, w);
  hipLaunchKernelGGL(k_reduce_w<F>, dim3((F * F + 255) / 256, N + 1), dim3(256), 0, st, rd);
  hipLaunchKernelGGL(k_reduce_b<F>, dim3(1, N + 2), dim3(F), 0, st, rd);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

template <int F, bool H16>
static int launch_wgrad16_t(afx_ctx* c, const WgradArgs& w, const ReduceArgs& rd, int N, hipStream_t st) {
  {
    // 2 stages x (dZ + H image of 64 samples, padded chunk columns) (+ f16 mode: 2 stages of scaled dL/draw)
    const size_t lds = (size_t)4 * (F / 8) * (64 * 16 + 64) + (H16 ? 2 * (64 * 2 + 64 * 4) : 0);
    if (!c->attr_done.count((const void*)k_wgrad_bf16<F, H16>)) {
      HIPCHK(hipFuncSetAttribute((const void*)k_wgrad_bf16<F, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      c->attr_done.insert((const void*)k_wgrad_bf16<F, H16>);
    }
    ProfScope ps(c, AFX_K_WGRAD, st);
    // + the first layer (encoded inputs) and the fourier-coefficient contraction as extra grid rows
    hipLaunchKernelGGL((k_wgrad_bf16<F, H16>), dim3(w.n_splits, N + (w.enc16 ? 1 + (w.coef_cols > 0 ? 1 : 0) : 0)), dim3(512), lds, st, w);
  }
  if (w.small_groups) hipLaunchKernelGGL(k_small_from_groups<F>, dim3(rd.n_small), dim3(F, 4), 0, st, w);
  else hipLaunchKernelGGL((k_small_grads_bf16<F, H16>), dim3(rd.n_small, F / 64), dim3(256), 0, st, w);
  hipLaunchKernelGGL(k_reduce_w<F>, dim3((F * F + 255) / 256, N + 1), dim3(256), 0, st, rd);
  hipLaunchKernelGGL(k_reduce_b<F>, dim3(1, N + 2), dim3(F), 0, st, rd);
  hipLaunchKernelGGL(k_reduce_small<F>, dim3((unsigned)((F * rd.k0pad + 2 * F + 1 + 63) / 64)), dim3(64, 4), 0, st, rd);
  if (w.coef_cols > 0) hipLaunchKernelGGL(k_reduce_coef<F>, dim3(rd.coef_cols), dim3(F), 0, st, rd);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

// 8-bit stash (f16 mode, rays, no encoding): k_wgrad_s8 + the group-sum reduction
template <int F>
static int launch_wgrad8_t(afx_ctx* c, const WgradArgs& w, const ReduceArgs& rd, int N, hipStream_t st) {
  {
    const size_t lds = (size_t)4 * (2 * (F / 16) * (64 * 16 + 128) + 2 * 64 * 4);      // 4-stage ring of (J + H image, group exponents, H block scales)
    constexpr bool H6 = AFX_H6_ON;      // hidden layers: B = the 6-bit H stash
    for (const void* fn : {(const void*)k_wgrad_s8<F, H6>, (const void*)k_wgrad_s8<F, false>})
      if (!c->attr_done.count(fn)) {
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        c->attr_done.insert(fn);
      }
    ProfScope ps(c, AFX_K_WGRAD, st);
    const int extra = w.enc16 ? 1 + (w.coef_cols > 0 ? 1 : 0) : 0;      // the encoded first layer (and the fourier-coefficient contraction): B = the bf8 input stash
    if (H6 && extra) {
      hipLaunchKernelGGL((k_wgrad_s8<F, true>), dim3(w.n_splits, N), dim3(512), lds, st, w);
      WgradArgs w1 = w;
      w1.y0 = N;
      hipLaunchKernelGGL((k_wgrad_s8<F, false>), dim3(w.n_splits, extra), dim3(512), lds, st, w1);
    } else hipLaunchKernelGGL((k_wgrad_s8<F, H6>), dim3(w.n_splits, N + extra), dim3(512), lds, st, w);
  }
  hipLaunchKernelGGL(k_small_from_groups<F>, dim3(rd.n_small), dim3(F, 4), 0, st, w);
  if (w.no_sw) hipLaunchKernelGGL(k_wout_stash8<F>, dim3(rd.n_small), dim3(2 * F), 0, st, w);      // output layer from the stash of H_N (same records)
  {
    const int nwx = (F * F + 255) / 256, nw = nwx * (N + 1), nb = N + 2, ns = (int)((F * rd.k0pad + 2 * F + 1 + 63) / 64);
    hipLaunchKernelGGL(k_reduce_all<F>, dim3((unsigned)(nw + nb + ns)), dim3(256), 0, st, rd, nwx, nw, nb);
  }
  if (w.coef_cols > 0) hipLaunchKernelGGL(k_reduce_coef<F>, dim3(rd.coef_cols), dim3(F), 0, st, rd);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

// Shared by afx_render_backward / afx_mlp_backward: chain (recompute + input-gradient chain +
// stash) then the weight-gradient contraction, chunk by chunk.  `a` is fully filled except the
// backward pointers; `head` bytes at the start of the workspace are already in use (dod).
// `split` (fused MSE step whose rays straddle tiles): the chain runs as PHASE 1, per-ray finish (pixel, dL/d(optical depth)), PHASE 2 -
// a.dod / a.od_part / a.target / a.pixel / a.inv_n are set by the caller; chunks hold whole rays.
static int run_backward(afx_ctx* c, int prec, ChainArgs a, size_t head, char* ws, size_t ws_bytes, float* grad_flat, hipStream_t st,
                        bool split = false, int64_t n_rays = 0, const int64_t* goff = nullptr) {
  const int F = c->d.width, N = c->d.n_hidden;
  if (c->d.act != AFX_ACT_RELU && prec != AFX_PREC_F32)
    return fail(AFX_E_INVALID, "backward: tanh / sine models train in the exact-fp32 kernels only (AFX_PREC_F32); the 16-bit kernels are forward-only for them");
  BwdLayout B = bwd_layout(c, prec, 0);
  const size_t fixed = head + B.fixed_bytes;
  const int TILE = bwd_tile(prec);
  const bool b16 = is_bf16(prec);
  if (c->d_coef && c->d.enc == AFX_ENC_FOURIER && !b16)
    return fail(AFX_E_INVALID, "backward: the fourier coefficients' gradient (afx_set_encoding_grad) needs a 16-bit precision");
  // in-kernel small gradients: 8-wave 16-bit backward kernel, rays, raw coordinates as inputs (AFX_SMALL_IN_KERNEL=0: off)
  // (with an input encoding only the 8-bit-stash kernel has the in-kernel output-layer sums; its first layer goes through k_wgrad_s8)
  const bool sg = b16 && a.mode == 1 && c->small_in_kernel && (c->d.enc == AFX_ENC_NONE || prec == AFX_PREC_F16S8);
  const bool s8 = prec == AFX_PREC_F16S8 && sg;          // 8-bit stash: that configuration only; otherwise the 16-bit f16 path
  const size_t esz = s8 ? 1 : (b16 ? 2 : 4);            // stash element size
  const int k0ld = b16 ? 16 * nk0_of(c) : c->k0pad;     // row length of the encoded-input stash
  if (is_bf16(prec) && prec == AFX_PREC_F16S8 && a.mode == 1 && c->small_in_kernel)
    B.per_tile_bytes = per_tile_s8(c);     // 1 byte per stash element
  if (split && !s8) return fail(AFX_E_INVALID, "split training step: needs the 8-bit-stash kernel (AFX_PREC_F16S8)");
  // a chunk of the split step holds whole rays: a multiple of lcm(s_pad, tile) / tile tiles
  int64_t ray_tiles = 1;
  if (split) { int64_t x = a.s_pad, y = TILE; while (y) { const int64_t t_ = x % y; x = y; y = t_; } ray_tiles = a.s_pad / x; }
  if (ws_bytes < fixed + B.per_tile_bytes + 1024) return fail(AFX_E_WORKSPACE, "backward workspace %zu too small (min %zu)", ws_bytes, fixed + 32 * B.per_tile_bytes);
  const int64_t tiles = (a.n_total + TILE - 1) / TILE;
  int64_t chunk = (int64_t)((ws_bytes - fixed - 1024) / B.per_tile_bytes);      // 1 KiB slack for buffer alignment
  if (chunk > tiles) chunk = tiles;
  if (b16) {
    // The 16-/8-bit chain kernels address a layer's stash with 32-bit byte offsets (one VGPR per lane instead of two): a chunk's
    // layer plane must stay below 4 GiB.  (Until round 2 nothing enforced this: with a 128 GiB workspace the 512^2 x 128 projection
    // ran as 2 chunks of 8.3 GB planes and the rows beyond 4 GiB wrapped onto the first ones - wrong weight gradients, same timing.)
    const int64_t max_tiles = (int64_t)(((uint64_t)1 << 32) / ((uint64_t)TILE * F * esz));      // the largest offset used is plane - 16
    if (chunk > max_tiles) chunk = max_tiles;
  }
  if (chunk < tiles) {      // equal chunks instead of full ones and a remainder
    const int64_t n_chunks = (tiles + chunk - 1) / chunk;
    chunk = (tiles + n_chunks - 1) / n_chunks;
  }
  if (split && goff && chunk < tiles)      // packed samples: rays straddle tiles anywhere, the per-ray reduction needs the whole list
    return fail(AFX_E_WORKSPACE, "packed training step: the workspace must hold all %lld tiles in one chunk (%lld fit)", (long long)tiles, (long long)chunk);
  if (split && chunk < tiles) {
    if (chunk < ray_tiles) return fail(AFX_E_WORKSPACE, "backward workspace too small for one group of whole rays (%lld tiles)", (long long)ray_tiles);
    chunk = chunk / ray_tiles * ray_tiles;
  }
  // Overlap mode: two half-size stash buffers; the weight-gradient kernels of chunk i run on a side stream while
  // the chain kernel of chunk i+1 runs on the caller's stream (one is MFMA/HBM-write heavy, the other HBM-read
  // bound).  Fork/join with events only; the side stream always rejoins the caller's stream before returning.
  int nbuf = 1;
  // AFX_OVERLAP=2 (measurement): also a list that fits one chunk is cut in two, and the split-phase step takes part - the idea being that at the
  // reference's batch the chain halves are VALU-bound and the weight-gradient kernel HBM-bound.  Measured (tools/ref_iter.py): SLOWER, 1.12 -> 1.21 ms
  // per 5 625 x 300 iteration at 4x128, 5.32 -> 5.38 ms at 8x256 (two half-size launches of everything); the default stays off.
  const bool force2 = c->overlap == 2 && tiles >= 16 && !goff;
  if (c->overlap && (chunk < tiles || force2) && chunk >= 8 && (!split || force2)) {
    if (!c->side) {
      HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
      for (auto& e : c->ev_chain) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto& e : c->ev_wgrad) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    nbuf = 2;
    const int64_t one = chunk;
    chunk = force2 && chunk >= tiles ? (tiles + 1) / 2 : chunk / 2;
    if (split) chunk = (chunk + ray_tiles - 1) / ray_tiles * ray_tiles;      // whole rays per chunk
    if (chunk >= tiles || chunk < 1 || fixed + 1024 + (size_t)(2 * chunk) * B.per_tile_bytes > ws_bytes) { nbuf = 1; chunk = one; }      // (two buffers must fit)
  }
  size_t off = head;
  float* partial = (float*)(ws + off); off += rup64((size_t)(N + 2) * kSplits * F * F * 4, 256);
  float* partial2 = (float*)(ws + off); off += rup64((size_t)(N + 2) * kSplits * (F + 4) * 4, 256);
  float* partial_s = (float*)(ws + off);
  if (b16) off += rup64((size_t)kSmallBlocks * (F * k0ld + 2 * F + 4) * 4, 256);
  uint32_t* gmax_words = (uint32_t*)(ws + off); off += 256;
  const bool h16 = is_f16(prec);
  const size_t rows = (size_t)chunk * TILE;
  float *stash_h[2], *stash_dz[2], *stash_e[2], *graw[2], *gpart[2];
  char* masks[2];
  uint32_t* hexp[2];
  for (int bI = 0; bI < nbuf; ++bI) {
    stash_h[bI] = (float*)(ws + off); off += (size_t)(N + 1) * rows * F * esz;
    stash_dz[bI] = (float*)(ws + off); off += (size_t)(N + 1) * rows * F * esz;
    stash_e[bI] = (float*)(ws + off); off += rows * k0ld * 4;
    graw[bI] = (float*)(ws + off); off += rup64(rows * 4, 256);
    gpart[bI] = nullptr; masks[bI] = nullptr; hexp[bI] = nullptr;
    if (s8) {
      gpart[bI] = (float*)(ws + off); off += rows * 4;
      masks[bI] = ws + off; off += (size_t)chunk * (N + 1) * c->nt * 1024;
      hexp[bI] = (uint32_t*)(ws + off); off += (size_t)N * (rows / 32) * 4;
    }
  }
  a.stash_rows = (int64_t)rows;
  a.debug = 0;
  // encoded inputs, 16-bit kernels: the inputs are stashed in 16-bit chunk-major form and the first layer's weight gradient (and the
  // fourier coefficients' gradient, afx_set_encoding_grad) is contracted on the matrix pipe by k_wgrad_bf16
  const bool enc16 = b16 && c->d.enc != AFX_ENC_NONE;
  a.coef_cols = (enc16 && c->d_coef && c->d.enc == AFX_ENC_FOURIER) ? 3 * c->d.n_freq : 0;
  // in-kernel small gradients: 8-wave bf16 backward kernel, rays, raw coordinates as inputs (AFX_SMALL_IN_KERNEL=0: off)
  a.stash8 = s8 ? 1 : 0;
  a.persistent = (nbuf == 2 && !c->persistent_chain) ? 0 : 1;
  int64_t ci = 0;
  for (int64_t t0 = 0; t0 < tiles; t0 += chunk, ++ci) {
    const int64_t t1 = t0 + chunk < tiles ? t0 + chunk : tiles;
    const int bI = nbuf == 2 ? (int)(ci & 1) : 0;
    a.tile0 = (int)t0; a.tile1 = (int)t1;
    a.stash_h = stash_h[bI]; a.stash_dz = stash_dz[bI]; a.stash_e = stash_e[bI]; a.graw = graw[bI];
    a.gexp = (int32_t*)graw[bI];       // 8-bit stash: dL/draw itself is not stashed; its slot holds the group exponents
    a.small_part = sg ? (float*)((char*)stash_h[bI] + (size_t)N * rows * F * esz) : nullptr;     // H_N's stash is not written then
    if (nbuf == 2 && ci >= 2) HIPCHK(hipStreamWaitEvent(st, c->ev_wgrad[bI], 0));     // buffer bI has been consumed
    a.gmax = h16 ? gmax_words + 16 * bI : nullptr;
    if (h16) HIPCHK(hipMemsetAsync(a.gmax, 0, 4, st));
    a.gpart = gpart[bI]; a.masks = masks[bI]; a.hexp = hexp[bI];
    int rc;
    if (split) {
      // forward half of the chunk, then the per-ray reduction over the chunk's (whole) rays, then the backward half
      ChainArgs p = a;
      p.fused = 0;
      rc = launch_chain(c, prec, true, p, st, 1);
      if (rc) return rc;
      const int64_t ray0 = t0 * TILE / a.s_pad;
      int64_t ray1 = t1 * TILE / a.s_pad;
      if (ray1 > n_rays || t1 == tiles) ray1 = n_rays;
      const int gpr = a.s_pad / GROUP;
      if (goff)
        hipLaunchKernelGGL(k_finish_mse_packed, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, (const float*)a.od_part, goff, n_rays, a.target,
                           a.inv_n, a.pixel, (float*)a.dod);
      else
        hipLaunchKernelGGL(k_finish_mse, dim3((unsigned)((ray1 - ray0 + 255) / 256)), dim3(256), 0, st, (const float*)a.od_part + ray0 * gpr, gpr,
                           ray1 - ray0, a.target + ray0, a.inv_n, a.pixel + ray0, (float*)a.dod + ray0);
      rc = launch_chain(c, prec, true, p, st, 2);
    } else rc = launch_chain(c, prec, true, a, st);
    if (rc) return rc;
    hipStream_t ws_st = st;
    if (nbuf == 2) {
      HIPCHK(hipEventRecord(c->ev_chain[bI], st));
      HIPCHK(hipStreamWaitEvent(c->side, c->ev_chain[bI], 0));
      ws_st = c->side;
    }
    WgradArgs w = {};      // (zero: fields a path does not use must read as "off")
    w.stash_h = a.stash_h; w.stash_dz = a.stash_dz; w.stash_e = a.stash_e; w.graw = a.graw;
    w.rows = (t1 - t0) * TILE;
    w.stride_rows = (int64_t)rows;   // a short last chunk keeps the full-chunk layer stride
    w.n_hidden = N; w.k0 = c->k0; w.k0pad = k0ld;
    // (splits x (N+1)) workgroups of 8 waves, one per CU at a time: fill the chip in whole rounds
    int splits = b16 ? c->n_cu / N : (2 * c->n_cu) / (N + 1);
    if (splits > (int)(w.rows / 256)) splits = (int)(w.rows / 256);
    if (splits < 1) splits = 1;
    if (splits > kSplits) splits = kSplits;
    w.n_splits = splits;
    int64_t rps = (w.rows + splits - 1) / splits;
    rps = (rps + 63) / 64 * 64;        // whole 32-/64-sample stages
    w.rows_per_split = (int)rps;
    w.partial = partial; w.partial2 = partial2; w.partial_s = partial_s; w.debug = a.debug; w.small_groups = sg ? 1 : 0;
    w.gmax = a.gmax; w.stash_esz = (int)esz; w.gexp = a.gexp; w.hexp = a.hexp; w.enc16 = enc16 ? 1 : 0; w.coef_cols = a.coef_cols;
    w.dod = split ? a.dod : nullptr; w.gpr = a.s_pad / GROUP; w.group0 = t0 * (TILE / GROUP); w.group_ray = goff ? a.group_ray : nullptr;
    w.n_groups_valid = goff ? a.n_total / GROUP : n_rays * (int64_t)(a.s_pad / GROUP);
    ReduceArgs rd = {};
    rd.partial = partial; rd.partial2 = partial2; rd.n_hidden = N; rd.k0 = c->k0; rd.k0pad = k0ld; rd.n_splits = splits;
    rd.grad = grad_flat; rd.hidden_only = b16 ? 1 : 0; rd.partial_s = partial_s;
    // records (every one is written, possibly with zero rows): a block per ~4 groups of a small list, so that a sparse grid iteration (a few hundred
    // groups) does not write and sum 1 024 records of which most are zeros
    rd.n_small = (int)std::min<int64_t>(kSmallBlocks, std::max<int64_t>(64, (w.rows / GROUP + 3) / 4));
    rd.gmax = a.gmax; rd.scale_shift = s8 ? AFX_S8_JSHIFT : 0; rd.layer0_mfma = enc16 ? 1 : 0;
    rd.w0 = c->coef_params; rd.d_coef = c->d_coef; rd.coef_cols = a.coef_cols;
    if (!b16) rc = F == 64 ? launch_wgrad_t<64>(c, w, rd, N, ws_st) : (F == 128 ? launch_wgrad_t<128>(c, w, rd, N, ws_st) : launch_wgrad_t<256>(c, w, rd, N, ws_st));
    else if (s8) rc = F == 64 ? launch_wgrad8_t<64>(c, w, rd, N, ws_st) : (F == 128 ? launch_wgrad8_t<128>(c, w, rd, N, ws_st) : launch_wgrad8_t<256>(c, w, rd, N, ws_st));
    else if (h16) rc = F == 64 ? launch_wgrad16_t<64, true>(c, w, rd, N, ws_st) : (F == 128 ? launch_wgrad16_t<128, true>(c, w, rd, N, ws_st) : launch_wgrad16_t<256, true>(c, w, rd, N, ws_st));
    else rc = F == 64 ? launch_wgrad16_t<64, false>(c, w, rd, N, ws_st) : (F == 128 ? launch_wgrad16_t<128, false>(c, w, rd, N, ws_st) : launch_wgrad16_t<256, false>(c, w, rd, N, ws_st));
    if (rc) return rc;
    if (nbuf == 2) HIPCHK(hipEventRecord(c->ev_wgrad[bI], c->side));
  }
  if (nbuf == 2) {       // join: everything on the side stream is ordered before whatever follows on the caller's stream
    HIPCHK(hipStreamWaitEvent(st, c->ev_wgrad[0], 0));
    if (ci >= 2) HIPCHK(hipStreamWaitEvent(st, c->ev_wgrad[1], 0));
  }
  return AFX_OK;
}

extern "C" int afx_render_backward(afx_ctx* c, int prec, const void* prepared, const afx_render_args* r,
                                   const float* dL_dpixel, float* grad_flat, void* stream) {
  if (c && r && r->n_rays == 0) return AFX_OK;
  int rc = check_render(c, r, "afx_render_backward");
  if (rc) return rc;
  if (check_prec(prec, "afx_render_backward")) return AFX_E_INVALID;
  if (!prepared || !dL_dpixel || !grad_flat) return fail(AFX_E_INVALID, "afx_render_backward: null argument");
  if (r->n_rays == 0) return AFX_OK;
  if (!r->workspace) return fail(AFX_E_WORKSPACE, "afx_render_backward: workspace required");
  hipStream_t st = (hipStream_t)stream;
  ChainArgs a = {};
  fill_model(c, prec, true, prepared, a);
  fill_render(r, a);
  a.sigma = nullptr; a.tau = nullptr;
  const size_t head = rup64((size_t)r->n_rays * 4, 256);
  if (r->workspace_bytes < head) return fail(AFX_E_WORKSPACE, "afx_render_backward: workspace too small");
  float* dod = (float*)r->workspace;
  hipLaunchKernelGGL(k_finish_bwd, dim3((unsigned)((r->n_rays + 255) / 256)), dim3(256), 0, st, r->pixel, dL_dpixel, r->n_rays, dod);
  a.dod = dod;
  return run_backward(c, prec, a, head, (char*)r->workspace, r->workspace_bytes, grad_flat, st);
}

extern "C" int afx_train_step_mse(afx_ctx* c, int prec, const void* prepared, const afx_render_args* r,
                                  const float* target, float inv_n, float* grad_flat, void* stream) {
  if (c && r && r->n_rays == 0) return AFX_OK;
  int rc = check_render(c, r, "afx_train_step_mse");
  if (rc) return rc;
  if (!is_bf16(prec)) return fail(AFX_E_INVALID, "afx_train_step_mse: needs a bf16 precision (use afx_render_forward/backward for f32)");
  if (!prepared || !target || !grad_flat || !r->workspace) return fail(AFX_E_INVALID, "afx_train_step_mse: null argument");
  const int64_t s_pad = s_pad_of(r->n_samples);
  // rays inside one workgroup tile: ONE kernel per chunk.  Otherwise (the reference's own 300 samples per ray, the 128 + 64 of the
  // hierarchical pass) the same work as two launches per chunk - forward half, per-ray reduction, backward half - for the
  // 8-bit-stash kernel; other precisions refuse (the Python layer renders, then calls afx_render_backward, which recomputes the forward)
  const bool split = 256 % s_pad != 0;
  const bool can_split = prec == AFX_PREC_F16S8 && c->small_in_kernel;
  if (split && !can_split)
    return fail(AFX_E_INVALID, "afx_train_step_mse: padded samples per ray (%d) must divide 256 at this precision", (int)s_pad);
  ChainArgs a = {};
  fill_model(c, prec, true, prepared, a);
  fill_render(r, a);
  a.sigma = nullptr; a.tau = nullptr;
  a.fused = 1; a.target = target; a.pixel = r->pixel; a.inv_n = inv_n;
  const bool do_split = split || (can_split && c->force_split);
  size_t head = 0;
  if (do_split) {
    const size_t dod_bytes = rup64((size_t)r->n_rays * 4, 256), od_bytes = rup64((size_t)r->n_rays * (size_t)(s_pad / GROUP) * 4, 256);
    if (r->workspace_bytes < dod_bytes + od_bytes) return fail(AFX_E_WORKSPACE, "afx_train_step_mse: workspace too small");
    a.dod = (const float*)r->workspace;
    a.od_part = (float*)((char*)r->workspace + dod_bytes);
    head = dod_bytes + od_bytes;
  }
  return run_backward(c, prec, a, head, (char*)r->workspace, r->workspace_bytes, grad_flat, (hipStream_t)stream, do_split, r->n_rays);
}

extern "C" int afx_pack_groups(const int64_t* offsets, const int64_t* group_offsets, int64_t n_rays, const float* t_starts, const float* t_ends,
                               float* ts_pad, float* te_pad, int32_t* group_ray, void* stream) {
  if (n_rays <= 0) return AFX_OK;
  if (!offsets || !group_offsets || !ts_pad || !te_pad || !group_ray) return fail(AFX_E_INVALID, "afx_pack_groups: null argument");
  hipLaunchKernelGGL(k_pack_groups, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, offsets, group_offsets, n_rays, t_starts, t_ends,
                     ts_pad, te_pad, group_ray);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_train_step_packed_mse(afx_ctx* c, int prec, const void* prepared, const float* origins, const float* dirs, int64_t n_rays,
                                         const int64_t* group_offsets, const int32_t* group_ray, int64_t n_groups, const float* ts_pad,
                                         const float* te_pad, const float* target, float inv_n, float* pixel, float* grad_flat,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !prepared || !origins || !dirs || !group_offsets || !target || !pixel || !grad_flat || !workspace)
    return fail(AFX_E_INVALID, "afx_train_step_packed_mse: null argument");
  if (n_rays <= 0) return AFX_OK;
  if (n_groups < 0 || n_groups * 32 > ((int64_t)1 << 31) - 256) return fail(AFX_E_INVALID, "afx_train_step_packed_mse: n_groups out of range");
  if (prec != AFX_PREC_F16S8 || !c->small_in_kernel)
    return fail(AFX_E_INVALID, "afx_train_step_packed_mse: AFX_PREC_F16S8 only");
  hipStream_t st = (hipStream_t)stream;
  const size_t dod_bytes = rup64((size_t)n_rays * 4, 256), od_bytes = rup64((size_t)std::max<int64_t>(n_groups, 1) * 4, 256);
  if (workspace_bytes < dod_bytes + od_bytes) return fail(AFX_E_WORKSPACE, "afx_train_step_packed_mse: workspace too small");
  float* dod = (float*)workspace;
  float* od_part = (float*)((char*)workspace + dod_bytes);
  if (n_groups == 0) {      // nothing survived the march: every pixel is the empty product, no gradient (the reference skips the step, :293)
    hipLaunchKernelGGL(k_finish_mse_packed, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, (const float*)od_part, group_offsets, n_rays, target,
                       inv_n, pixel, dod);
    HIPCHK(hipGetLastError());
    return AFX_OK;
  }
  if (!group_ray || !ts_pad || !te_pad) return fail(AFX_E_INVALID, "afx_train_step_packed_mse: null argument");
  ChainArgs a = {};
  fill_model(c, prec, true, prepared, a);
  a.mode = 1; a.org = origins; a.dir = dirs; a.poses = nullptr;
  a.depth_mode = 4; a.z = ts_pad; a.te = te_pad; a.group_ray = group_ray;
  a.n_samples = GROUP; a.s_pad = GROUP; a.n_total = n_groups * GROUP;
  a.fused = 1; a.target = target; a.pixel = pixel; a.inv_n = inv_n;
  a.dod = dod; a.od_part = od_part;
  return run_backward(c, prec, a, dod_bytes + od_bytes, (char*)workspace, workspace_bytes, grad_flat, st, true, n_rays, group_offsets);
}

// ---- hierarchical training step with coarse re-use
// Bytes of one sample set of a ray chunk (rows padded to whole tiles): 8-bit H_l and dZ'_l planes (N + 1 each: H_N is stashed here), group exponents,
// dL/draw per row, the tiles' mask images, the group records.
static size_t hier_set_bytes(const afx_ctx* c, size_t rows) {
  const size_t F = c->d.width, N = c->d.n_hidden, tiles = rows / 256;
  return 2 * (N + 1) * rows * F + rup64(rows * 4, 256) + rows * 4 + tiles * (N + 1) * c->nt * 1024 + rup64((rows / 32) * (3 * F + 8) * 4, 256) +
         rup64(N * (rows / 32) * 4, 256);      // + H block scales
}
static size_t hier_fixed_bytes(const afx_ctx* c, int64_t n_rays, int S, int NF) {
  const size_t F = c->d.width, N = c->d.n_hidden;
  size_t b = 2 * rup64((size_t)n_rays * S * 4, 256) + 2 * rup64((size_t)n_rays * NF * 4, 256);                    // sigma / tau of the coarse set, new depths, sigma of the new set
  b += rup64((size_t)n_rays * (size_t)(std::max(s_pad_of(S), s_pad_of(NF)) / GROUP) * 4, 256);                    // optical-depth partials (written by PHASE 1, unused here)
  b += rup64((N + 2) * (size_t)kSplits * F * F * 4, 256) + rup64((N + 2) * (size_t)kSplits * (F + 4) * 4, 256);
  b += rup64((size_t)kSmallBlocks * (F * 16 + 2 * F + 4) * 4, 256) + 256;
  return b;
}

extern "C" int64_t afx_hier_workspace_bytes(const afx_ctx* c, int64_t n_rays, int32_t n_coarse, int32_t n_fine) {
  if (!c || n_rays <= 0) return 0;
  const uint64_t plane_rows = ((uint64_t)1 << 32) / (uint64_t)c->d.width;
  int64_t nr = (n_rays + 7) / 8 * 8;
  const int64_t cap = (int64_t)(plane_rows / (uint64_t)s_pad_of(n_coarse)) / 8 * 8;
  if (nr > cap) nr = cap;
  return (int64_t)(hier_fixed_bytes(c, n_rays, n_coarse, n_fine) + hier_set_bytes(c, rup64((size_t)nr * s_pad_of(n_coarse), 256)) +
                   hier_set_bytes(c, rup64((size_t)nr * s_pad_of(n_fine), 256)) + 4096);
}

extern "C" int afx_hier_train_step_mse(afx_ctx* c, int prec, const void* prepared, const afx_render_args* r, int32_t n_fine, const float* u,
                                       const float* target, float inv_n, float* z_all, float* grad_flat, void* stream) {
  if (c && r && r->n_rays == 0) return AFX_OK;
  int rc = check_render(c, r, "afx_hier_train_step_mse");
  if (rc) return rc;
  if (!prepared || !u || !target || !grad_flat || !r->workspace) return fail(AFX_E_INVALID, "afx_hier_train_step_mse: null argument");
  if (prec != AFX_PREC_F16S8 || c->d.enc != AFX_ENC_NONE || !c->small_in_kernel || c->d.act != AFX_ACT_RELU)
    return fail(AFX_E_INVALID, "afx_hier_train_step_mse: AFX_PREC_F16S8, ReLU, no input encoding only");
  if (r->depth_mode != AFX_DEPTH_SHARED_Z && r->depth_mode != AFX_DEPTH_PER_RAY_Z)
    return fail(AFX_E_INVALID, "afx_hier_train_step_mse: coarse depths z[S] or z[R,S] (dense convention) required");
  const int S = r->n_samples, NF = n_fine;
  if (S < 3 || S > AFX_MAX_COARSE || NF < 1 || NF > AFX_MAX_FINE) return fail(AFX_E_INVALID, "afx_hier_train_step_mse: n_coarse in 3..%d, n_fine in 1..%d", AFX_MAX_COARSE, AFX_MAX_FINE);
  if (NF < 2) return fail(AFX_E_INVALID, "afx_hier_train_step_mse: n_fine must be >= 2");
  hipStream_t st = (hipStream_t)stream;
  const int F = c->d.width, N = c->d.n_hidden;
  const int64_t R = r->n_rays, spA = s_pad_of(S), spB = s_pad_of(NF);
  char* ws = (char*)r->workspace;
  const size_t fixed = hier_fixed_bytes(c, R, S, NF);
  if (r->workspace_bytes < fixed + 4096) return fail(AFX_E_WORKSPACE, "afx_hier_train_step_mse: workspace %zu too small", r->workspace_bytes);
  // rays per chunk: a multiple of 8 (so that a chunk's first sample of either set starts a 256-sample tile), both sets' planes below 4 GiB
  const uint64_t plane_rows = ((uint64_t)1 << 32) / (uint64_t)F;
  int64_t nr = (R + 7) / 8 * 8;
  nr = std::min<int64_t>(nr, (int64_t)(plane_rows / (uint64_t)spA) / 8 * 8);
  while (nr >= 8 && fixed + hier_set_bytes(c, rup64((size_t)nr * spA, 256)) + hier_set_bytes(c, rup64((size_t)nr * spB, 256)) + 4096 > r->workspace_bytes) {
    if (nr == 8) { nr = 0; break; }
    const int64_t n_chunks = (R + nr - 1) / nr + 1;          // next larger chunk count, equal chunks
    const int64_t next = ((R + n_chunks - 1) / n_chunks + 7) / 8 * 8;
    nr = next < nr ? next : nr - 8;                         // (strictly decreasing: the rounding to 8 rays can stall the chunk count)
  }
  if (nr < 8) return fail(AFX_E_WORKSPACE, "afx_hier_train_step_mse: workspace %zu too small for 8 rays per chunk", r->workspace_bytes);
  if (int rc2 = check_dev(c, "afx_hier_train_step_mse")) return rc2;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = ws + off; off += rup64(bytes, 256); return p; };
  float* sigA = (float*)take((size_t)R * S * 4);  float* tauA = (float*)take((size_t)R * S * 4);
  float* zf = (float*)take((size_t)R * NF * 4);   float* sigB = (float*)take((size_t)R * NF * 4);
  float* od_part = (float*)take((size_t)R * (size_t)(std::max(spA, spB) / GROUP) * 4);
  float* partial = (float*)take((size_t)(N + 2) * kSplits * F * F * 4);
  float* partial2 = (float*)take((size_t)(N + 2) * kSplits * (F + 4) * 4);
  float* partial_s = (float*)take((size_t)kSmallBlocks * (F * 16 + 2 * F + 4) * 4);
  uint32_t* gmax = (uint32_t*)take(256);
  struct SetBuf { char *stash_h, *stash_dz, *gexp, *masks, *hexp; float *gpart, *records; size_t rows; };
  auto carve = [&](size_t rows) {
    SetBuf b; b.rows = rows;
    b.stash_h = take((size_t)(N + 1) * rows * F); b.stash_dz = take((size_t)(N + 1) * rows * F);
    b.gexp = take(rows * 4); b.gpart = (float*)take(rows * 4);
    b.masks = take((rows / 256) * (size_t)(N + 1) * c->nt * 1024);
    b.records = (float*)take((rows / 32) * (size_t)(3 * F + 8) * 4);
    b.hexp = take((size_t)N * (rows / 32) * 4);
    return b;
  };
  const SetBuf A = carve(rup64((size_t)nr * spA, 256)), B = carve(rup64((size_t)nr * spB, 256));
  if (off > r->workspace_bytes) return fail(AFX_E_WORKSPACE, "afx_hier_train_step_mse: workspace layout exceeds the buffer (%zu > %zu)", off, r->workspace_bytes);

  ChainArgs base = {};
  fill_model(c, prec, true, prepared, base);
  fill_render(r, base);
  base.fused = 0; base.stash8 = 1; base.coef_cols = 0; base.debug = 0; base.persistent = 1;
  base.od_part = od_part; base.gmax = gmax; base.defer_out = 1; base.dod = nullptr; base.pixel = r->pixel;
  auto set_args = [&](const ChainArgs& src, const SetBuf& b, int64_t r0, int64_t r1, int64_t sp) {
    ChainArgs a = src;
    a.tile0 = (int)(r0 * sp / 256); a.tile1 = (int)((r1 * sp + 255) / 256);
    a.stash_h = (float*)b.stash_h; a.stash_dz = (float*)b.stash_dz; a.stash_e = nullptr; a.graw = (float*)b.gexp; a.gexp = (int32_t*)b.gexp;
    a.gpart = b.gpart; a.masks = b.masks; a.small_part = b.records; a.stash_rows = (int64_t)b.rows; a.hexp = (uint32_t*)b.hexp;
    return a;
  };
  ChainArgs argsA = base;                     // the coarse set: the caller's depths
  argsA.sigma = sigA; argsA.tau = tauA;
  ChainArgs argsB = base;                     // the new set: per-ray depths zf[R, NF]
  argsB.depth_mode = AFX_DEPTH_PER_RAY_Z; argsB.z = zf; argsB.n_samples = NF; argsB.s_pad = (int)spB; argsB.n_total = R * spB;
  argsB.sigma = sigB; argsB.tau = nullptr;
  auto wgrad_set = [&](const ChainArgs& a, const SetBuf& b) -> int {
    WgradArgs w = {};
    w.stash_h = a.stash_h; w.stash_dz = a.stash_dz; w.stash_e = nullptr; w.graw = a.graw;
    w.rows = (int64_t)(a.tile1 - a.tile0) * 256; w.stride_rows = (int64_t)b.rows;
    w.n_hidden = N; w.k0 = c->k0; w.k0pad = 16;
    int splits = c->n_cu / N;
    if (splits > (int)(w.rows / 256)) splits = (int)(w.rows / 256);
    if (splits < 1) splits = 1;
    if (splits > kSplits) splits = kSplits;
    w.n_splits = splits;
    w.rows_per_split = (int)(((w.rows + splits - 1) / splits + 63) / 64 * 64);
    w.partial = partial; w.partial2 = partial2; w.partial_s = partial_s; w.debug = 0; w.small_groups = 1;
    w.gmax = gmax; w.stash_esz = 1; w.gexp = a.gexp; w.hexp = a.hexp; w.enc16 = 0; w.coef_cols = 0;
    w.dod = nullptr; w.gpr = 1; w.group0 = 0; w.n_groups_valid = 0; w.group_ray = nullptr;
    w.records = b.records; w.no_sw = 1; w.gfull = b.gpart;
    ReduceArgs rd = {};
    rd.partial = partial; rd.partial2 = partial2; rd.n_hidden = N; rd.k0 = c->k0; rd.k0pad = 16; rd.n_splits = splits;
    rd.grad = grad_flat; rd.hidden_only = 1; rd.partial_s = partial_s; rd.n_small = kSmallBlocks;
    rd.gmax = gmax; rd.scale_shift = AFX_S8_JSHIFT; rd.layer0_mfma = 0; rd.w0 = nullptr; rd.d_coef = nullptr; rd.coef_cols = 0;
    return F == 64 ? launch_wgrad8_t<64>(c, w, rd, N, st) : (F == 128 ? launch_wgrad8_t<128>(c, w, rd, N, st) : launch_wgrad8_t<256>(c, w, rd, N, st));
  };
  for (int64_t r0 = 0; r0 < R; r0 += nr) {
    const int64_t r1 = std::min<int64_t>(r0 + nr, R), n = r1 - r0;
    const ChainArgs a = set_args(argsA, A, r0, r1, spA), b = set_args(argsB, B, r0, r1, spB);
    // coarse set: forward half (H_l incl. H_N, masks, sigma, tau) -> the new depths -> new set: forward half -> per-ray composite of both
    if ((rc = launch_chain(c, prec, true, a, st, 1))) return rc;
    hipLaunchKernelGGL(k_fine_depths, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, r->depth_mode == AFX_DEPTH_PER_RAY_Z ? r->z + r0 * S : r->z,
                       r->depth_mode == AFX_DEPTH_PER_RAY_Z ? 1 : 0, (const float*)nullptr, (const float*)(tauA + r0 * S), u + r0 * NF, n, S, NF,
                       (float*)nullptr, zf + r0 * NF);
    if ((rc = launch_chain(c, prec, true, b, st, 1))) return rc;
    hipLaunchKernelGGL(k_hier_composite, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a, r0, n, (const float*)zf, NF, (const float*)sigA, (const float*)sigB,
                       target, inv_n, r->pixel, A.gpart, (int)spA, B.gpart, (int)spB, z_all);
    HIPCHK(hipMemsetAsync(gmax, 0, 4, st));
    // backward halves (dL/draw per sample comes finished from the composite), then the weight gradients of both sets
    if ((rc = launch_chain(c, prec, true, a, st, 2))) return rc;
    if ((rc = launch_chain(c, prec, true, b, st, 2))) return rc;
    if ((rc = wgrad_set(a, A))) return rc;
    if ((rc = wgrad_set(b, B))) return rc;
  }
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_mlp_backward(afx_ctx* c, int prec, const void* prepared, const float* pts, int64_t n_pts,
                                const float* d_out, float* grad_flat, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !prepared || !pts || !d_out || !grad_flat || !workspace) return fail(AFX_E_INVALID, "afx_mlp_backward: null argument");
  if (check_prec(prec, "afx_mlp_backward")) return AFX_E_INVALID;
  if (n_pts < 0 || n_pts > ((int64_t)1 << 31) - 256) return fail(AFX_E_INVALID, "afx_mlp_backward: n_pts must be < 2^31 per call");
  if (n_pts == 0) return AFX_OK;
  ChainArgs a = {};
  fill_model(c, prec, true, prepared, a);
  a.n_total = n_pts; a.mode = 0; a.pts = pts; a.dod = d_out;
  return run_backward(c, prec, a, 0, (char*)workspace, workspace_bytes, grad_flat, (hipStream_t)stream);
}

extern "C" int afx_project_volume(const float* vol, int32_t nx, int32_t ny, int32_t nz, const double origin[3], const double spacing[3],
                                  float fill_value, const afx_render_args* r, int type_ct, void* stream) {
  if (!vol || !origin || !spacing || !r) return fail(AFX_E_INVALID, "afx_project_volume: null argument");
  if (nx < 2 || ny < 2 || nz < 2) return fail(AFX_E_INVALID, "afx_project_volume: volume needs >= 2 voxels per axis");
  if (!(spacing[0] > 0 && spacing[1] > 0 && spacing[2] > 0)) return fail(AFX_E_INVALID, "afx_project_volume: spacing must be > 0");
  if (r->n_rays == 0) return AFX_OK;
  if (r->n_rays < 0 || r->n_samples < 1 || !r->z || !r->pixel || r->depth_mode != AFX_DEPTH_SHARED_Z)
    return fail(AFX_E_INVALID, "afx_project_volume: needs n_samples >= 1, shared z[S] (AFX_DEPTH_SHARED_Z) and pixel");
  if (r->ray_mode == AFX_RAYS_ARRAYS) { if (!r->origins || !r->dirs) return fail(AFX_E_INVALID, "afx_project_volume: origins/dirs required"); }
  else if (r->ray_mode == AFX_RAYS_POSE) { if (!r->poses || r->width <= 0 || r->height <= 0 || !(r->focal > 0)) return fail(AFX_E_INVALID, "afx_project_volume: poses/width/height/focal required"); }
  else return fail(AFX_E_INVALID, "afx_project_volume: bad ray_mode");
  ChainArgs a = {};
  a.org = r->origins; a.dir = r->dirs; a.poses = r->ray_mode == AFX_RAYS_POSE ? r->poses : nullptr;
  a.ray_ids = r->ray_ids; a.ray_id0 = r->ray_id0; a.width = r->width; a.height = r->height; a.focal = r->focal;
  a.n_samples = r->n_samples; a.z = r->z; a.n_total = r->n_rays; a.pixel = r->pixel;
  VolArgs v = {};
  v.vol = vol; v.nx = nx; v.ny = ny; v.nz = nz; v.x0 = origin[0]; v.y0 = origin[1]; v.z0 = origin[2];
  v.dx = spacing[0]; v.dy = spacing[1]; v.dz = spacing[2]; v.fill = fill_value; v.type_ct = type_ct;
  hipLaunchKernelGGL(k_project_volume, dim3((unsigned)((r->n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, v);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_composite_dense(const float* raw, const float* dirs, const float* z, int z_per_ray, int64_t n_rays,
                                   int32_t n_samples, float* rgb_map, float* depth_map, float* weights, float* entropy,
                                   float* sigma, void* stream) {
  if (!raw || !dirs || !z || !rgb_map) return fail(AFX_E_INVALID, "afx_composite_dense: null argument");
  if (n_samples < 1) return fail(AFX_E_INVALID, "afx_composite_dense: n_samples < 1");
  if (n_rays <= 0) return AFX_OK;
  hipLaunchKernelGGL(k_composite_dense, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, (hipStream_t)stream, raw, dirs, z,
                     z_per_ray, n_rays, n_samples, rgb_map, depth_map, weights, entropy, sigma);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_composite_dense_backward(const float* raw, const float* dirs, const float* z, int z_per_ray,
                                            int64_t n_rays, int32_t n_samples, const float* rgb_map,
                                            const float* d_rgb_map, float* d_raw, void* stream) {
  if (!raw || !dirs || !z || !rgb_map || !d_rgb_map || !d_raw) return fail(AFX_E_INVALID, "afx_composite_dense_backward: null argument");
  if (n_rays <= 0) return AFX_OK;
  const int64_t n = n_rays * n_samples;
  hipLaunchKernelGGL(k_composite_dense_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, dirs, z,
                     z_per_ray, n_rays, n_samples, rgb_map, d_rgb_map, d_raw);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_composite_packed(const float* pred, const int32_t* ri, const float* ts, const float* te, int64_t n,
                                    int64_t n_rays, float* rgb_map, void* stream) {
  if (!rgb_map || (n > 0 && (!pred || !ri || !ts || !te))) return fail(AFX_E_INVALID, "afx_composite_packed: null argument");
  if (n_rays <= 0) return AFX_OK;
  hipLaunchKernelGGL(k_composite_packed, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pred, ri, ts, te,
                     n, n_rays, rgb_map);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_composite_packed_backward(const float* pred, const int32_t* ri, const float* ts, const float* te,
                                             int64_t n, int64_t n_rays, const float* rgb_map, const float* d_rgb_map,
                                             float* d_pred, void* stream) {
  (void)n_rays;
  if (n <= 0) return AFX_OK;
  if (!pred || !ri || !ts || !te || !rgb_map || !d_rgb_map || !d_pred) return fail(AFX_E_INVALID, "afx_composite_packed_backward: null argument");
  hipLaunchKernelGGL(k_composite_packed_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, ri, ts,
                     te, n, rgb_map, d_rgb_map, d_pred);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

static int fine_depths_impl(const char* who, const float* z_coarse, int z_per_ray, const float* w_coarse, const float* tau, const float* u,
                            int64_t n_rays, int32_t n_coarse, int32_t n_fine, float* z_out, void* stream) {
  if (!z_coarse || (!w_coarse && !tau) || !u || !z_out) return fail(AFX_E_INVALID, "%s: null argument", who);
  if (n_coarse < 3 || n_coarse > AFX_MAX_COARSE) return fail(AFX_E_INVALID, "%s: n_coarse must be in 3..%d", who, AFX_MAX_COARSE);
  if (n_fine < 1 || n_fine > AFX_MAX_FINE) return fail(AFX_E_INVALID, "%s: n_fine must be in 1..%d", who, AFX_MAX_FINE);
  if (n_rays <= 0) return AFX_OK;
  hipLaunchKernelGGL(k_fine_depths, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, (hipStream_t)stream, z_coarse, z_per_ray,
                     w_coarse, tau, u, n_rays, n_coarse, n_fine, z_out, (float*)nullptr);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_fine_depths(const float* z_coarse, int z_per_ray, const float* w_coarse, const float* u, int64_t n_rays,
                               int32_t n_coarse, int32_t n_fine, float* z_out, void* stream) {
  return fine_depths_impl("afx_fine_depths", z_coarse, z_per_ray, w_coarse, nullptr, u, n_rays, n_coarse, n_fine, z_out, stream);
}

extern "C" int afx_fine_depths_from_tau(const float* z_coarse, int z_per_ray, const float* tau_coarse, const float* u, int64_t n_rays,
                                        int32_t n_coarse, int32_t n_fine, float* z_out, void* stream) {
  return fine_depths_impl("afx_fine_depths_from_tau", z_coarse, z_per_ray, nullptr, tau_coarse, u, n_rays, n_coarse, n_fine, z_out, stream);
}


// ---------------------------------------------------------------------------------------- occupancy grid, march, sampler
static GridDesc grid_of(const afx_grid_desc* g) {
  GridDesc d;
  for (int i = 0; i < 3; ++i) { d.lo[i] = g->roi_aabb[i]; d.hi[i] = g->roi_aabb[3 + i]; d.res[i] = g->resolution[i]; }
  return d;
}
static int check_grid(const afx_grid_desc* g, const char* who, int64_t* n_cells) {
  if (!g) return fail(AFX_E_INVALID, "%s: null grid", who);
  for (int i = 0; i < 3; ++i) {
    if (g->resolution[i] < 1 || g->resolution[i] > 2048) return fail(AFX_E_INVALID, "%s: resolution must be in 1..2048", who);
    if (!(g->roi_aabb[3 + i] > g->roi_aabb[i])) return fail(AFX_E_INVALID, "%s: empty roi_aabb", who);
  }
  *n_cells = (int64_t)g->resolution[0] * g->resolution[1] * g->resolution[2];
  return AFX_OK;
}
static inline dim3 blocks_for(int64_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

extern "C" int afx_grid_points(const afx_grid_desc* grid, const int32_t* cell_idx, int64_t n, const float* jitter, uint64_t seed,
                               uint64_t stream_id, float* pts, void* stream) {
  int64_t nc;
  if (int rc = check_grid(grid, "afx_grid_points", &nc)) return rc;
  if (n < 0 || (!cell_idx && n > nc)) return fail(AFX_E_INVALID, "afx_grid_points: n out of range");
  if (n == 0) return AFX_OK;
  if (!pts) return fail(AFX_E_INVALID, "afx_grid_points: null pts");
  hipLaunchKernelGGL(k_grid_points, blocks_for(n), dim3(256), 0, (hipStream_t)stream, cell_idx, n, jitter, seed, stream_id, grid_of(grid), pts);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_grid_update(const afx_grid_desc* grid, float* occs, float* occs_scratch, const int32_t* cell_idx, int64_t n,
                               const float* occ_new, float ema_decay, void* stream) {
  int64_t nc;
  if (int rc = check_grid(grid, "afx_grid_update", &nc)) return rc;
  if (n < 0 || (!cell_idx && n > nc)) return fail(AFX_E_INVALID, "afx_grid_update: n out of range");
  if (n == 0) return AFX_OK;
  if (!occs || !occs_scratch || !occ_new) return fail(AFX_E_INVALID, "afx_grid_update: null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(occs_scratch, occs, (size_t)nc * 4, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_grid_decay, blocks_for(n), dim3(256), 0, st, occs, (const float*)occs_scratch, cell_idx, n, ema_decay);
  hipLaunchKernelGGL(k_grid_ema, blocks_for(n), dim3(256), 0, st, occs, cell_idx, occ_new, n);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_grid_binarize(const afx_grid_desc* grid, const float* occs, float occ_thre, uint8_t* binary, uint32_t* bits,
                                 double* partial_ws, void* stream) {
  int64_t nc;
  if (int rc = check_grid(grid, "afx_grid_binarize", &nc)) return rc;
  if (!occs || !binary || !bits || !partial_ws) return fail(AFX_E_INVALID, "afx_grid_binarize: null argument");
  hipStream_t st = (hipStream_t)stream;
  const int np = 256;
  hipLaunchKernelGGL(k_grid_sum, dim3(np), dim3(256), 0, st, occs, nc, partial_ws);
  hipLaunchKernelGGL(k_grid_binarize, blocks_for((nc + 31) / 32), dim3(256), 0, st, occs, nc, (const double*)partial_ws, np, occ_thre, binary, bits);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_grid_pack(const afx_grid_desc* grid, const uint8_t* binary, uint32_t* bits, void* stream) {
  int64_t nc;
  if (int rc = check_grid(grid, "afx_grid_pack", &nc)) return rc;
  if (!binary || !bits) return fail(AFX_E_INVALID, "afx_grid_pack: null argument");
  hipLaunchKernelGGL(k_grid_pack, blocks_for((nc + 31) / 32), dim3(256), 0, (hipStream_t)stream, binary, nc, bits);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

static int fill_march(const afx_march_args* m, MarchArgs& a, const char* who) {
  if (!m) return fail(AFX_E_INVALID, "%s: null args", who);
  if (m->n_rays < 0) return fail(AFX_E_INVALID, "%s: n_rays < 0", who);
  if (m->n_rays > 0 && (!m->origins || !m->dirs)) return fail(AFX_E_INVALID, "%s: origins/dirs required", who);
  if (!(m->step > 0.f)) return fail(AFX_E_INVALID, "%s: step must be > 0", who);
  a.org = m->origins; a.dir = m->dirs; a.n_rays = m->n_rays;
  a.has_aabb = m->has_aabb;
  for (int i = 0; i < 6; ++i) a.aabb[i] = m->scene_aabb[i];
  a.has_near = m->has_near; a.has_far = m->has_far; a.near_plane = m->near_plane; a.far_plane = m->far_plane;
  a.dt = m->step; a.bits = m->grid_bits;
  if (m->grid_bits) {
    int64_t nc;
    if (int rc = check_grid(&m->grid, who, &nc)) return rc;
    a.g = grid_of(&m->grid);
  } else a.g = GridDesc{};
  if (!m->has_aabb && !m->has_far) return fail(AFX_E_INVALID, "%s: an unbounded ray needs scene_aabb or far_plane", who);
  return AFX_OK;
}

extern "C" int afx_march_count(const afx_march_args* args, int32_t* counts, void* stream) {
  MarchArgs a = {};
  if (int rc = fill_march(args, a, "afx_march_count")) return rc;
  if (a.n_rays == 0) return AFX_OK;
  if (!counts) return fail(AFX_E_INVALID, "afx_march_count: null counts");
  hipLaunchKernelGGL(k_march_count, blocks_for(a.n_rays, 4), dim3(256), 0, (hipStream_t)stream, a, counts);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_march_write(const afx_march_args* args, const int64_t* offsets, int32_t* ray_indices, float* t_starts,
                               float* t_ends, float* mid_points, void* stream) {
  MarchArgs a = {};
  if (int rc = fill_march(args, a, "afx_march_write")) return rc;
  if (a.n_rays == 0) return AFX_OK;
  if (!offsets || !ray_indices || !t_starts || !t_ends) return fail(AFX_E_INVALID, "afx_march_write: null argument");
  hipLaunchKernelGGL(k_march_write, blocks_for(a.n_rays, 4), dim3(256), 0, (hipStream_t)stream, a, offsets, ray_indices, t_starts, t_ends, mid_points);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_march_visibility(const float* raw, int32_t input_is_alpha, const float* t_starts, const float* t_ends, const int64_t* offsets,
                                    int64_t n_rays, float early_stop_eps, float alpha_thre, uint8_t* keep, int32_t* counts, void* stream) {
  if (n_rays <= 0) return AFX_OK;
  if (!offsets || !keep || !counts) return fail(AFX_E_INVALID, "afx_march_visibility: null argument");
  hipLaunchKernelGGL(k_march_visibility, blocks_for(n_rays, 4), dim3(256), 0, (hipStream_t)stream, raw, (int)input_is_alpha, t_starts, t_ends, offsets, n_rays,
                     early_stop_eps, alpha_thre, keep, counts);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_march_compact(const uint8_t* keep, const int64_t* offsets_in, const int64_t* offsets_out, int64_t n_rays,
                                 const float* t_starts_in, const float* t_ends_in, int32_t* ray_indices_out, float* t_starts_out,
                                 float* t_ends_out, void* stream) {
  if (n_rays <= 0) return AFX_OK;
  if (!keep || !offsets_in || !offsets_out) return fail(AFX_E_INVALID, "afx_march_compact: null argument");
  hipLaunchKernelGGL(k_march_compact, blocks_for(n_rays, 4), dim3(256), 0, (hipStream_t)stream, keep, offsets_in, offsets_out, n_rays,
                     t_starts_in, t_ends_in, ray_indices_out, t_starts_out, t_ends_out);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_ray_offsets(const int32_t* counts, int64_t n_rays, int64_t* offsets, int64_t* group_offsets, int64_t* totals, void* stream) {
  if (n_rays < 0 || !offsets || (n_rays > 0 && !counts)) return fail(AFX_E_INVALID, "afx_ray_offsets: null argument");
  hipLaunchKernelGGL(k_ray_offsets, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, n_rays, offsets, group_offsets, totals);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_sample_keys(const float* weights, int64_t n, const float* u, uint64_t seed, uint64_t stream_id, float* keys, void* stream) {
  if (n <= 0) return AFX_OK;
  if (!keys) return fail(AFX_E_INVALID, "afx_sample_keys: null keys");
  hipLaunchKernelGGL(k_sample_keys, blocks_for(n), dim3(256), 0, (hipStream_t)stream, weights, n, u, seed, stream_id, keys);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" size_t afx_topk_workspace_bytes(int64_t n) {
  const int64_t nb = (n + SEL_PER_BLOCK - 1) / SEL_PER_BLOCK;
  return (size_t)SEL_BINS * 4 + 256 + (size_t)2 * (nb > 0 ? nb : 1) * 4;
}

extern "C" int afx_topk_indices(const float* keys, int64_t n, int64_t k, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (k == 0) return AFX_OK;
  if (n <= 0 || k < 0 || k > n || n >= ((int64_t)1 << 32)) return fail(AFX_E_INVALID, "afx_topk_indices: need 0 <= k <= n < 2^32");
  if (!keys || !out_idx || !workspace) return fail(AFX_E_INVALID, "afx_topk_indices: null argument");
  if (workspace_bytes < afx_topk_workspace_bytes(n)) return fail(AFX_E_WORKSPACE, "afx_topk_indices: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (n + SEL_PER_BLOCK - 1) / SEL_PER_BLOCK;
  uint32_t* hist = (uint32_t*)workspace;
  SelState* state = (SelState*)((char*)workspace + SEL_BINS * 4);
  uint32_t* cnt_gt = (uint32_t*)((char*)workspace + SEL_BINS * 4 + 256);
  uint32_t* cnt_eq = cnt_gt + nb;
  int hb = (int)((n + 256 * 16 - 1) / (256 * 16));      // ~16 keys per thread in the histogram passes
  if (hb < 1) hb = 1;
  if (hb > 2048) hb = 2048;
  hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(256), 0, st, state, (uint32_t)k, hist);
  const int shifts[3] = {21, 10, 0};
  const uint32_t binmask[3] = {0x7ffu, 0x7ffu, 0x3ffu}, himask[3] = {0u, 0xffe00000u, 0xfffffc00u};
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(k_sel_hist, dim3(hb), dim3(256), 0, st, keys, n, state, himask[p], shifts[p], binmask[p], hist);
    hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(256), 0, st, hist, state, shifts[p]);
  }
  hipLaunchKernelGGL(k_sel_count, dim3((unsigned)nb), dim3(256), 0, st, keys, n, state, cnt_gt, cnt_eq);
  hipLaunchKernelGGL(k_sel_offsets, dim3(1), dim3(1024), 0, st, cnt_gt, cnt_eq, nb);
  hipLaunchKernelGGL(k_sel_write, dim3((unsigned)nb), dim3(256), 0, st, keys, n, state, cnt_gt, cnt_eq, k, out_idx);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

// B independent draws per launch sequence: draw b = afx_sample_keys(stream_id0 + b) followed by afx_topk_indices, bit for bit
extern "C" size_t afx_sample_batches_workspace_bytes(int64_t n, int32_t n_batches) {
  const int64_t nb = (n + SEL_PER_BLOCK - 1) / SEL_PER_BLOCK, B = n_batches > 0 ? n_batches : 1;
  return (size_t)B * ((size_t)(n > 0 ? n : 0) * 4 + (size_t)SEL_BINS * 4 + sizeof(SelState) + (size_t)2 * (nb > 0 ? nb : 1) * 4) + 1024;
}

extern "C" int afx_sample_batches(const float* weights, int64_t n, uint64_t seed, uint64_t stream_id0, int32_t n_batches, int64_t k,
                                  int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
  if (k == 0 || n_batches == 0) return AFX_OK;
  if (n <= 0 || k < 0 || k > n || n >= ((int64_t)1 << 32) || n_batches < 0 || n_batches > 65535)
    return fail(AFX_E_INVALID, "afx_sample_batches: need 0 <= k <= n < 2^32 and 0 <= n_batches <= 65535");
  if (!out_idx || !workspace) return fail(AFX_E_INVALID, "afx_sample_batches: null argument");
  if (workspace_bytes < afx_sample_batches_workspace_bytes(n, n_batches)) return fail(AFX_E_WORKSPACE, "afx_sample_batches: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const unsigned B = (unsigned)n_batches;
  const int64_t nb = (n + SEL_PER_BLOCK - 1) / SEL_PER_BLOCK;
  char* w = (char*)workspace;
  float* keys = (float*)w;            w += ((size_t)B * n * 4 + 255) / 256 * 256;
  uint32_t* hist = (uint32_t*)w;      w += (size_t)B * SEL_BINS * 4;
  SelState* state = (SelState*)w;     w += ((size_t)B * sizeof(SelState) + 255) / 256 * 256;
  uint32_t* cnt_gt = (uint32_t*)w;
  uint32_t* cnt_eq = cnt_gt + (size_t)B * nb;
  int hb = (int)((n + 256 * 16 - 1) / (256 * 16));
  if (hb < 1) hb = 1;
  if (hb > 2048) hb = 2048;
  hipLaunchKernelGGL(k_sample_keys, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, weights, n, (const float*)nullptr, seed, stream_id0, keys);
  hipLaunchKernelGGL(k_sel_init, dim3(1, B), dim3(256), 0, st, state, (uint32_t)k, hist);
  const int shifts[3] = {21, 10, 0};
  const uint32_t binmask[3] = {0x7ffu, 0x7ffu, 0x3ffu}, himask[3] = {0u, 0xffe00000u, 0xfffffc00u};
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(k_sel_hist, dim3(hb, B), dim3(256), 0, st, (const float*)keys, n, (const SelState*)state, himask[p], shifts[p], binmask[p], hist);
    hipLaunchKernelGGL(k_sel_scan, dim3(1, B), dim3(256), 0, st, hist, state, shifts[p]);
  }
  hipLaunchKernelGGL(k_sel_count, dim3((unsigned)nb, B), dim3(256), 0, st, (const float*)keys, n, (const SelState*)state, cnt_gt, cnt_eq);
  hipLaunchKernelGGL(k_sel_offsets, dim3(1, B), dim3(1024), 0, st, cnt_gt, cnt_eq, nb);
  hipLaunchKernelGGL(k_sel_write, dim3((unsigned)nb, B), dim3(256), 0, st, (const float*)keys, n, (const SelState*)state, (const uint32_t*)cnt_gt,
                     (const uint32_t*)cnt_eq, k, out_idx);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_gather_rays(const float* origins, const float* dirs, const float* pixels, const int64_t* idx, int64_t k,
                               float* origins_out, float* dirs_out, float* pixels_out, void* stream) {
  if (k <= 0) return AFX_OK;
  if (!origins || !dirs || !idx || !origins_out || !dirs_out) return fail(AFX_E_INVALID, "afx_gather_rays: null argument");
  hipLaunchKernelGGL(k_gather_rays, blocks_for(k), dim3(256), 0, (hipStream_t)stream, origins, dirs, pixels, idx, k, origins_out, dirs_out, pixels_out);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

extern "C" int afx_philox_uniform(uint64_t seed, uint64_t stream_id, int64_t n, float* out, void* stream) {
  if (n <= 0) return AFX_OK;
  if (!out) return fail(AFX_E_INVALID, "afx_philox_uniform: null out");
  hipLaunchKernelGGL(k_philox_fill, blocks_for(n), dim3(256), 0, (hipStream_t)stream, seed, stream_id, n, out);
  HIPCHK(hipGetLastError());
  return AFX_OK;
}

// ---- the reference's grid iteration in ONE call (nerf/run_nerf_acc.py:284-306): march -> alpha pass -> visibility -> packed training step.
// The same entry points, in the same order, that the Python mirror calls one by one (occupancy.ray_marching, render.train_step_packed_mse) -
// composed here because at the reference's batch the GPU work of an iteration is 0.27 ms and ~30 launches issued from Python are not.
extern "C" int afx_march_train_step_mse(afx_ctx* c, int prec, const void* prepared, afx_march_train_args* t, void* stream) {
  if (!c || !prepared || !t) return fail(AFX_E_INVALID, "afx_march_train_step_mse: null argument");
  t->n_candidates = t->n_kept = t->n_groups = 0; t->workspace_needed = 0;
  const afx_march_args& m = t->march;
  const int64_t R = m.n_rays;
  if (R <= 0) return AFX_OK;
  if (!m.origins || !m.dirs || !t->target || !t->pixel || !t->grad_flat || !t->workspace)
    return fail(AFX_E_INVALID, "afx_march_train_step_mse: null argument");
  if (prec != AFX_PREC_F16S8 || !c->small_in_kernel) return fail(AFX_E_INVALID, "afx_march_train_step_mse: AFX_PREC_F16S8 only");
  if (int rc = check_dev(c, "afx_march_train_step_mse")) return rc;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)t->workspace;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += rup64(bytes, 256); return at; };
  auto too_small = [&](size_t more) {
    t->workspace_needed = off + more + 4096;
    return fail(AFX_E_WORKSPACE, "afx_march_train_step_mse: workspace %zu < %zu bytes", t->workspace_bytes, t->workspace_needed);
  };
  // 1. candidates: steps of the march whose cell is occupied
  const size_t o_counts = take((size_t)R * 4), o_offsets = take((size_t)(R + 1) * 8), o_totals = take(4 * 8);
  const size_t o_counts2 = take((size_t)R * 4), o_off2 = take((size_t)(R + 1) * 8), o_goff = take((size_t)(R + 1) * 8);
  if (off > t->workspace_bytes) return too_small(0);
  int32_t* counts = (int32_t*)(ws + o_counts);
  int64_t* offsets = (int64_t*)(ws + o_offsets);
  int64_t* totals = (int64_t*)(ws + o_totals);
  int rc;
  // The two size read-backs: the offsets kernel posts its totals and a sequence tag to host-mapped memory and the host polls the tag
  // (AFX_MAILBOX=0: a 16-byte copy + stream synchronisation behind the kernel instead - 27 us per read-back slower, tools/grid_iter.py).
  static const bool use_mailbox = !(getenv("AFX_MAILBOX") && atoi(getenv("AFX_MAILBOX")) == 0);
  if (use_mailbox && !c->mailbox) {
    if (hipHostMalloc((void**)&c->mailbox, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&c->mailbox_dev, c->mailbox, 0) != hipSuccess) {
      if (c->mailbox) (void)hipHostFree(c->mailbox);
      c->mailbox = c->mailbox_dev = nullptr;
    } else for (int i = 0; i < 8; ++i) c->mailbox[i] = 0;
  }
  const bool mail = use_mailbox && c->mailbox;
  int64_t h[4] = {0, 0, 0, 0};
  auto read_totals = [&](const int32_t* cnt, int64_t* offs, int64_t* goffs, int slot) -> int {
    if (!mail) {
      if (int r = afx_ray_offsets(cnt, R, offs, goffs, totals + 2 * slot, stream)) return r;
      HIPCHK(hipMemcpyAsync(h + 2 * slot, totals + 2 * slot, 16, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      return AFX_OK;
    }
    const int64_t tag = ++c->mail_seq;
    hipLaunchKernelGGL(k_ray_offsets, dim3(1), dim3(1024), 0, st, cnt, R, offs, goffs, totals + 2 * slot, (volatile int64_t*)(c->mailbox_dev + 4 * slot), tag);
    HIPCHK(hipGetLastError());
    volatile int64_t* mb = c->mailbox + 4 * slot;
    for (uint64_t spins = 0; mb[2] != tag; ++spins) {
      if ((spins & 0xfffff) == 0xfffff && hipStreamQuery(st) != hipErrorNotReady) {      // the stream has drained (or failed): one last look
        HIPCHK(hipStreamSynchronize(st));
        if (mb[2] != tag) return fail(AFX_E_HIP, "afx_march_train_step_mse: the offsets kernel never posted its totals");
        break;
      }
      __builtin_ia32_pause();
    }
    h[2 * slot] = mb[0]; h[2 * slot + 1] = mb[1];
    return AFX_OK;
  };
  if ((rc = afx_march_count(&m, counts, stream))) return rc;
  if ((rc = read_totals(counts, offsets, nullptr, 0))) return rc;      // (sizes are data: the candidate count)
  const int64_t n = h[0];
  t->n_candidates = n;
  if (n == 0) return AFX_OK;
  if (n > ((int64_t)1 << 31) - 256) return fail(AFX_E_INVALID, "afx_march_train_step_mse: %lld candidates", (long long)n);
  const size_t o_ri = take((size_t)n * 4), o_ts = take((size_t)n * 4), o_te = take((size_t)n * 4), o_pts = take((size_t)n * 12),
               o_raw = take((size_t)n * 4), o_keep = take((size_t)n);
  if (off > t->workspace_bytes) return too_small(0);
  int32_t* ri = (int32_t*)(ws + o_ri);
  float *ts = (float*)(ws + o_ts), *te = (float*)(ws + o_te), *pts = (float*)(ws + o_pts), *raw = (float*)(ws + o_raw);
  uint8_t* keep = (uint8_t*)(ws + o_keep);
  if ((rc = afx_march_write(&m, offsets, ri, ts, te, pts, stream))) return rc;
  // 2. alpha pass (alpha_fn, nerf_helpers_acc.py:11-25) + render_visibility
  if ((rc = afx_mlp_infer(c, prec, prepared, pts, n, raw, 0, stream))) return rc;
  int32_t* counts2 = (int32_t*)(ws + o_counts2);
  int64_t *off2 = (int64_t*)(ws + o_off2), *goff = (int64_t*)(ws + o_goff);
  if ((rc = afx_march_visibility(raw, 0, ts, te, offsets, R, t->early_stop_eps, t->alpha_thre, keep, counts2, stream))) return rc;
  if ((rc = read_totals(counts2, off2, goff, 1))) return rc;      // (the kept count and the group count)
  const int64_t n2 = h[2], ng = h[3];
  t->n_kept = n2; t->n_groups = ng;
  if (n2 == 0) return AFX_OK;            // nothing survived: the reference skips the step (:293)
  // 3. compaction, group-aligned copy, fused training step
  const size_t o_ri2 = take((size_t)n2 * 4), o_ts2 = take((size_t)n2 * 4), o_te2 = take((size_t)n2 * 4);
  const size_t o_tsp = take((size_t)ng * 32 * 4), o_tep = take((size_t)ng * 32 * 4), o_gray = take((size_t)ng * 4);
  const size_t step_ws = (size_t)afx_query(c, AFX_Q_BWD_WORKSPACE_FULL, 0, ng * 32, prec) + 4 * (size_t)(R + ng) + 1024;
  if (off + step_ws > t->workspace_bytes) return too_small(step_ws);
  int32_t* ri2 = (int32_t*)(ws + o_ri2);
  float *ts2 = (float*)(ws + o_ts2), *te2 = (float*)(ws + o_te2), *tsp = (float*)(ws + o_tsp), *tep = (float*)(ws + o_tep);
  int32_t* gray = (int32_t*)(ws + o_gray);
  if ((rc = afx_march_compact(keep, offsets, off2, R, ts, te, ri2, ts2, te2, stream))) return rc;
  if ((rc = afx_pack_groups(off2, goff, R, ts2, te2, tsp, tep, gray, stream))) return rc;
  return afx_train_step_packed_mse(c, prec, prepared, m.origins, m.dirs, R, goff, gray, ng, tsp, tep, t->target, t->inv_n, t->pixel, t->grad_flat,
                                   ws + off, t->workspace_bytes - off, stream);
}
