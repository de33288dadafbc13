/* afx.h — C-ABI of the MI355X-native angiography-NeRF hot path ("afx").
 *
 * The upstream reference (kirstenmaas/nerf-for-angiography) has no FFI/plugin
 * boundary: its hot path is Python calling torch.  The boundary this library
 * replaces is therefore the set of Python functions cited on each entry point
 * below (file:line in the upstream tree).  Everything here is plain C: raw
 * device pointers, sizes and a hipStream_t passed as void*.  No torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative afx_status otherwise;
 *     afx_last_error() gives a thread-local message.  Nothing throws or exits.
 *   - the CALLER owns every buffer (parameters, gradients, outputs, prepared
 *     weights, workspace).  The library allocates nothing on the device and
 *     holds only immutable descriptors => calls are hipGraph-capturable.
 *   - all calls are asynchronous on the stream passed in; no hidden syncs.
 *   - results are deterministic: no floating-point atomics anywhere.
 */
#ifndef AFX_H
#define AFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct afx_ctx afx_ctx;

enum afx_status {
  AFX_OK = 0,
  AFX_E_INVALID = -1,      /* bad argument / unsupported configuration */
  AFX_E_WORKSPACE = -2,    /* workspace or prepared-weights buffer too small */
  AFX_E_HIP = -3           /* a HIP runtime call failed (message has the code) */
};

/* Positional encodings of CPPN.pos_enc (model/CPPN.py:207-234). */
enum { AFX_ENC_NONE = 0, AFX_ENC_BARF = 1, AFX_ENC_FOURIER = 2 };

/* Arithmetic of the MLP contractions.
 *   F32    : v_mfma_f32_32x32x2_f32, exact fp32 (strict parity mode)
 *   BF16X3 : split-bf16 (hi+lo) operands, 3 bf16 MFMAs per product, fp32
 *            accumulate: fp32-grade accuracy at 1/3 of the bf16 rate
 *   BF16   : bf16 operands, fp32 accumulate (first layer always split)
 *   F16    : f16 operands in the hidden layers (v_mfma_f32_32x32x16_f16: 11
 *            significant bits at the bf16 rate), fp32 accumulate, first layer
 *            split bf16.  Rendered pixels within ~1e-5 relative L2 of fp32.
 *            Backward: the input-gradient chain runs normalised by dL/draw
 *            (no loss scaling needed), the weight-gradient contraction carries
 *            a power-of-two scale taken from the batch's largest |dL/draw|.
 *            RANGE: f16 holds |x| <= 65504 - a hidden activation beyond that becomes inf and the pixel NaN (the
 *            conversion is v_cvt_pk_f16_f32, round-to-nearest, no saturation: a clamp would cost two VALU
 *            instructions per packed pair in the hottest epilogue).  With nn.Linear-initialised weights and world
 *            coordinates of +-100 the activations of an 8x256 model stay below ~1e2; a model that can exceed the
 *            range belongs on BF16 / BF16X3 (fp32 exponent range).  The training driver stops on a non-finite loss.
 *   F16S8  : F16 arithmetic; the backward pass keeps its per-sample stash
 *            (H_l, and the normalised chain J_l) as bf8 (e5m2) instead of f16:
 *            half the HBM round trip that bounds a training step.  Forward
 *            results are identical to F16; weight gradients carry the extra
 *            bf8 rounding of the two contraction operands (H to nearest, dZ'
 *            stochastically: its errors are systematic otherwise), averaged over
 *            the samples: 2e-3 relative L2 of the whole gradient (F16: 5e-4).  Rays mode (with an input encoding
 *            the encoded inputs are stashed as bf8 as well); points mode
 *            (afx_mlp_backward) runs exactly as F16.                           */
enum { AFX_PREC_F32 = 0, AFX_PREC_BF16X3 = 1, AFX_PREC_BF16 = 2, AFX_PREC_F16 = 3, AFX_PREC_F16S8 = 4 };

/* CPPN(model_definition) — model/CPPN.py:10-139.  The configuration
 * nerf/run_nerf_acc.py:168-183 builds is accelerated end to end (ReLU, no skip block,
 * no view-direction head, one output channel); tanh / sine variants of it in the forward direction at every precision and in the
 * backward direction at AFX_PREC_F32. */
/* act_func of the hidden layers (model/CPPN.py:53-60).  ReLU is what nerf/run_nerf_acc.py trains and what every kernel takes.
 * tanh and sine (Sine(w0) on the first layer, Sine() behind it, CPPN.py:278-300) are evaluated by the FORWARD entry points
 * (afx_mlp_infer, afx_render_forward: inference, evaluation renders, density grids; without an input encoding) at every precision.  The
 * backward entry points (afx_mlp_backward, afx_render_backward) take them at AFX_PREC_F32 - the exact-fp32 chain kernel keeps d act / dz per
 * element in the slot of the dZ_l stash (ReLU: one mask bit in LDS) - and return AFX_E_INVALID for them at the 16-bit precisions. */
enum { AFX_ACT_RELU = 0, AFX_ACT_TANH = 1, AFX_ACT_SINE = 2 };

typedef struct afx_model_desc {
  int32_t n_in;        /* num_input_channels (3)                              */
  int32_t enc;         /* AFX_ENC_*                                           */
  int32_t n_freq;      /* pos_enc_basis L (ignored for AFX_ENC_NONE)          */
  int32_t width;       /* num_filters: 64, 128 or 256                         */
  int32_t n_hidden;    /* num_early_layers N  (Linear count = N + 2)          */
  int32_t act;         /* AFX_ACT_*                                           */
  float act_w0;        /* AFX_ACT_SINE: sine_weights (first-layer frequency factor); ignored otherwise */
} afx_model_desc;

/* Flat fp32 parameter buffer layout (same order as the state-dict of the
 * reference): W0[width,K0] b0[width] W1[width,width] b1 ... WN bN
 * Wout[1,width] bout[1], K0 = n_in (+ 2*n_in*n_freq when encoded).           */

int  afx_create(const afx_model_desc* desc, afx_ctx** out);
void afx_destroy(afx_ctx* ctx);
const char* afx_last_error(void);

enum afx_query_what {
  AFX_Q_PARAM_COUNT = 0,       /* floats in the flat parameter buffer               */
  AFX_Q_K0 = 1,                /* encoded input width                               */
  AFX_Q_PREPARED_BYTES = 2,    /* bytes of the prepared-weights buffer (arg = prec) */
  AFX_Q_FWD_WORKSPACE = 3,     /* bytes needed by afx_render_forward  (arg0 = n_rays, arg1 = n_samples) */
  AFX_Q_BWD_WORKSPACE_MIN = 4, /* smallest sensible backward workspace (arg0 = n_rays, 0 for afx_mlp_backward; arg1 = n_samples when the split training step is meant) */
  AFX_Q_BWD_WORKSPACE_FULL = 5 /* workspace that lets backward run in one chunk (arg0 = n_rays, arg1 = n_samples; afx_mlp_backward: arg0 = 0, arg1 = n_pts) */
};
int64_t afx_query(const afx_ctx* ctx, int what, int64_t arg0, int64_t arg1, int64_t arg2);

/* Offset (in floats) and shape of linear layer `layer` (0 .. n_hidden+1) inside
 * the flat parameter buffer. */
int afx_param_layout(const afx_ctx* ctx, int layer, int64_t* w_off, int64_t* b_off,
                     int32_t* rows, int32_t* cols);

/* Re-tile the flat fp32 parameters into the MFMA operand order the kernels
 * stream through LDS (forward slabs, transposed slabs for the input-gradient
 * chain, permuted biases).  Must be re-run after every optimizer step.
 * enc_aux: BARF: 2*n_in*n_freq floats = [freq | weight] (CPPN.barf_freq,
 * CPPN.barf_weights, model/CPPN.py:84-85,244-259); FOURIER: n_in*n_freq
 * coefficients (CPPN.py:73-75); may be NULL for AFX_ENC_NONE. */
int afx_prepare_weights(afx_ctx* ctx, int prec, const float* params, const float* enc_aux,
                        void* prepared, size_t prepared_bytes, void* stream);

/* get_predictions(model, flattened_query_points, chunksize) — nerf/nerf_helpers.py:31-45
 * and get_predictions_vis — visualization/helpers.py:21-45 (density grid).
 * pts[P,3] fp32 -> out[P] = raw, or sigmoid(raw) when apply_sigmoid != 0.
 * The chunk loop of the reference is unnecessary (nothing is materialised). */
int afx_mlp_infer(afx_ctx* ctx, int prec, const void* prepared, const float* pts, int64_t n_pts,
                  float* out, int apply_sigmoid, void* stream);

/* Backward of afx_mlp_infer (apply_sigmoid = 0): grad_flat += d(sum_p d_out[p]*raw[p])/d(params).
 * This is what loss.backward() does through get_predictions (nerf/run_nerf_acc.py:294,306)
 * when compositing is done outside the fused kernel. */
int afx_mlp_backward(afx_ctx* ctx, int prec, const void* prepared, const float* pts, int64_t n_pts,
                     const float* d_out, float* grad_flat, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Where rays come from. */
enum { AFX_RAYS_ARRAYS = 0,   /* origins[R,3], dirs[R,3] fp32 (sample_pixel_rays output, nerf_helpers.py:137-150) */
       AFX_RAYS_POSE = 1 };   /* generated in-kernel: get_ray_values, phantomdata/helpers.py:156-175 */
/* Where depths along a ray come from, and which compositing convention. */
enum { AFX_DEPTH_UNIFORM_MID = 0, /* t_s = near+i*step, t_e = t_s+step, evaluate at (t_s+t_e)/2, dt = t_e-t_s:
                                     acc_ray_marching w/o grid + acc_render_volume_density, nerf_helpers_acc.py:10-63 */
       AFX_DEPTH_SHARED_Z = 1,    /* z[S] shared by all rays; render_volume_density, nerf_helpers.py:59-123:
                                     dt_i = (z[i+1]-z[i])*||d||, last dt = 1e10*||d||                                   */
       AFX_DEPTH_PER_RAY_Z = 2,   /* z[R,S] (hierarchical sampling, nerf_helpers.py:178-195); same convention   */
       AFX_DEPTH_STRATIFIED = 3 };/* randomize_depth (nerf_helpers.py:13-22) of z = linspace(t_near, t_far, S) IN the kernel:
                                     one jitter vector per call from Philox stream (jitter_seed, jitter_stream);
                                     render_volume_density convention.  Perf mode: parity mode passes the host's z (SHARED_Z) */

typedef struct afx_render_args {
  int64_t n_rays;
  int32_t n_samples;           /* S, samples per ray                                     */
  int32_t ray_mode;            /* AFX_RAYS_*                                             */
  const float* origins;        /* [R,3]  (AFX_RAYS_ARRAYS)                               */
  const float* dirs;           /* [R,3]                                                  */
  const double* poses;         /* [n_proj,3,4] row-major cam->world (AFX_RAYS_POSE), device */
  const int32_t* ray_ids;      /* [R] index into [n_proj,H,W]; NULL => ray r is index ray_id0 + r */
  int64_t ray_id0;
  int32_t width, height;
  double focal;
  int32_t depth_mode;          /* AFX_DEPTH_*                                            */
  float t_near, t_far;         /* AFX_DEPTH_UNIFORM_MID                                  */
  const float* z;              /* [S] or [R,S]                                           */
  float* pixel;                /* out [R]: rgb_map (transmittance)                       */
  float* sigma;                /* optional out [R,S]: sigmoid(raw)                       */
  float* tau;                  /* optional out [R,S]: sigma*dt (optical depth per sample) */
  void* workspace;
  size_t workspace_bytes;
  uint64_t jitter_seed, jitter_stream;   /* AFX_DEPTH_STRATIFIED */
} afx_render_args;

/* Fused ray generation -> sampling -> encoding -> MLP -> Beer-Lambert product.
 * Replaces the body of nerf/run_nerf_acc.py:287-296 (and :340-349 for eval). */
int afx_render_forward(afx_ctx* ctx, int prec, const void* prepared, const afx_render_args* args,
                       void* stream);

/* Backward of afx_render_forward w.r.t. the flat parameters:
 * grad_flat[param_count] += d(sum_r dL_dpixel[r] * pixel[r]) / d(params).
 * `pixel` in args must hold the forward result.  Replaces loss.backward()
 * through nerf/run_nerf_acc.py:289-296.  The workspace bounds the ray chunk
 * processed per pass (activations are recomputed, then stashed per chunk for
 * the weight-gradient contraction over samples). */
int afx_render_backward(afx_ctx* ctx, int prec, const void* prepared, const afx_render_args* args,
                        const float* dL_dpixel, float* grad_flat, void* stream);

/* ---- The reference's own iteration body on packed samples (nerf/run_nerf_acc.py:287-306): after the occupancy-grid march
 * (afx_march_* -> packed, ray-sorted t_starts / t_ends, offsets[R+1]) the reference gathers positions, evaluates the MLP (get_predictions),
 * multiplies the per-ray transmittances (acc_render_volume_density), takes the MSE and backpropagates.  afx_train_step_packed_mse does
 * all of that as the split-phase training step (forward half / per-ray reduction / backward half, see afx_train_step_mse) on a
 * GROUP-ALIGNED copy of the list: ray r's samples start at padded index 32 * group_offsets[r] (group_offsets = exclusive scan of
 * ceil(count_r / 32), int64 [R+1], the caller's cumsum), the tail of its last 32-sample group is dead padding (t_end <= t_start),
 * group_ray[g] = r - so a wavefront's 32 samples always belong to one ray.  afx_pack_groups builds ts_pad / te_pad / group_ray.
 * pixel[r] = prod exp(-sigmoid(raw) (t_e - t_s)) (1 for a ray without samples), L = inv_n sum_r (pixel_r - target_r)^2, grad_flat += dL/dparams.
 * AFX_PREC_F16S8; the workspace (AFX_Q_BWD_WORKSPACE_FULL with arg0 = n_rays, arg1 = 32 * n_groups / n_rays rounded up,
 * or simply arg0 = 0, arg1 = 32 * n_groups plus n_rays + n_groups floats) must hold the whole list in one chunk. */
int afx_pack_groups(const int64_t* offsets, const int64_t* group_offsets, int64_t n_rays, const float* t_starts, const float* t_ends,
                    float* ts_pad, float* te_pad, int32_t* group_ray, void* stream);
int afx_train_step_packed_mse(afx_ctx* ctx, int prec, const void* prepared, const float* origins, const float* dirs, int64_t n_rays,
                              const int64_t* group_offsets, const int32_t* group_ray, int64_t n_groups, const float* ts_pad,
                              const float* te_pad, const float* target, float inv_n, float* pixel, float* grad_flat,
                              void* workspace, size_t workspace_bytes, void* stream);

/* One fused training pass over a ray batch — the body of nerf/run_nerf_acc.py:287-306 (render, mse_loss,
 * backward) without materialising anything between the steps: the backward kernel's forward recompute IS the
 * forward pass; it composites each ray in-kernel, forms dL/dpixel = 2 (pixel - target) * inv_n
 * (L = inv_n * sum_r (pixel_r - target_r)^2, inv_n = 1 / global ray count) and runs the gradient chain.
 * grad_flat += dL/dparams; args->pixel receives the rendered pixels.  16-bit precisions only.
 * Rays whose padded sample count divides 256 lie inside one workgroup tile: ONE kernel per ray chunk.  Any other count (the
 * reference's own 300 samples per ray, nerf/run_nerf_acc.py:129; the 128 + 64 of the hierarchical pass) is taken by
 * AFX_PREC_F16S8 (with or without an input encoding) as the SAME work in two kernels per chunk - the forward half stashes H_l, the ReLU
 * masks and g' = dt sigma (1 - sigma) per sample, a per-ray reduction forms pixel and dL/d(optical depth), the backward half
 * runs the input-gradient chain from the masks - so nothing is computed twice; the workspace must then also hold
 * n_rays * (1 + padded samples / 32) floats (AFX_Q_BWD_WORKSPACE_FULL with arg0 = n_rays, arg1 = n_samples includes them).
 * Other precisions return AFX_E_INVALID for such counts (render, then afx_render_backward). */
int afx_train_step_mse(afx_ctx* ctx, int prec, const void* prepared, const afx_render_args* args,
                       const float* target, float inv_n, float* grad_flat, void* stream);

/* render_volume_density(radiance_field, ray_directions, depth_values) for one output channel —
 * nerf/nerf_helpers.py:59-123 (C == 1 branch), from a raw tensor already in memory.
 * raw[R,S], dirs[R,3], z[S] (z_per_ray=0) or [R,S]; all outputs optional except rgb_map.
 * weights = (1-alpha+1e-10)*cumprod_exclusive(alpha); depth_map = sum(alpha*z) (sic);
 * entropy per nerf_helpers.py:125-135. */
int afx_composite_dense(const float* raw, const float* dirs, const float* z, int z_per_ray,
                        int64_t n_rays, int32_t n_samples, float* rgb_map, float* depth_map,
                        float* weights, float* entropy, float* sigma, void* stream);
/* d raw = backward of rgb_map only (the quantity the loss uses). */
int afx_composite_dense_backward(const float* raw, const float* dirs, const float* z, int z_per_ray,
                                 int64_t n_rays, int32_t n_samples, const float* rgb_map,
                                 const float* d_rgb_map, float* d_raw, void* stream);

/* acc_render_volume_density(predictions, ray_indices, t_starts, t_ends, n_rays, ...) —
 * nerf/nerf_helpers_acc.py:45-63.  ray_indices must be sorted ascending (packed samples,
 * as nerfacc.ray_marching returns them).  rgb_map[r] = prod_{i in ray r} exp(-sigmoid(pred_i)*(te_i-ts_i)). */
int afx_composite_packed(const float* pred, const int32_t* ray_indices, const float* t_starts,
                         const float* t_ends, int64_t n, int64_t n_rays, float* rgb_map, void* stream);
int afx_composite_packed_backward(const float* pred, const int32_t* ray_indices, const float* t_starts,
                                  const float* t_ends, int64_t n, int64_t n_rays, const float* rgb_map,
                                  const float* d_rgb_map, float* d_pred, void* stream);

/* sample_pdf(bins, weights, N_samples) — nerf/nerf_helpers.py:197-222, with the uniform
 * draw u[R,n_fine] supplied by the caller; and the depth part of fine_sampling (:179-186):
 * bins = mid-points of z_coarse, weights = w_coarse[:,1:-1], output = sort(cat(z_coarse, samples)).
 * z_coarse [S] (z_per_ray=0) or [R,S]; w_coarse[R,S]; z_out[R,S+n_fine]. */
int afx_fine_depths(const float* z_coarse, int z_per_ray, const float* w_coarse, const float* u,
                    int64_t n_rays, int32_t n_coarse, int32_t n_fine, float* z_out, void* stream);

/* The same from the coarse pass's per-sample optical depths tau[R,S] (afx_render_forward's `tau` output) instead of its weights:
 * weights = (1 - alpha + 1e-10) * cumprod_exclusive(alpha), alpha = exp(-tau) (nerf/nerf_helpers.py:107-108) are formed per ray inside
 * the kernel, so the coarse -> fine hand-over of the hierarchical step needs no [R,S] passes in between. */
int afx_fine_depths_from_tau(const float* z_coarse, int z_per_ray, const float* tau_coarse, const float* u, int64_t n_rays,
                             int32_t n_coarse, int32_t n_fine, float* z_out, void* stream);

/* The hierarchical training step `fine_sampling` belongs to (nerf/nerf_helpers.py:178-195: coarse pass, sample_pdf on its weights, re-evaluation
 * of all S + N_f depths, compositing, MSE, backward through the fine pass), WITHOUT evaluating the coarse depths twice: coarse and fine network
 * are the same model here (fine_model = None, `:190`), and the coarse depths are a subset of the merged ones, so the coarse pass runs as the
 * forward HALF of the training kernel over the S coarse depths (stash, masks, sigma, tau), sample_pdf draws the N_f new depths from its weights, a
 * second forward half evaluates ONLY those, a per-ray kernel composites the merged list (step lengths of the merged order, last 1e10, x ||d||),
 * forms pixel, the MSE gradient and every sample's finished dL/draw, and the two backward halves + weight gradients follow.  The samples are
 * detached as upstream (`:186`).  args: rays (either ray mode), depth_mode AFX_DEPTH_SHARED_Z / PER_RAY_Z with the COARSE depths z, n_samples = S,
 * pixel out [R], workspace of afx_hier_workspace_bytes (fewer bytes: more ray chunks).  u[R, n_fine]: the uniform draws of sample_pdf.
 * z_all: optional out [R, S + n_fine], the merged depths.  AFX_PREC_F16S8, ReLU, no input encoding. */
int64_t afx_hier_workspace_bytes(const afx_ctx* ctx, int64_t n_rays, int32_t n_coarse, int32_t n_fine);
int afx_hier_train_step_mse(afx_ctx* ctx, int prec, const void* prepared, const afx_render_args* args, int32_t n_fine, const float* u,
                            const float* target, float inv_n, float* z_all, float* grad_flat, void* stream);

/* Ground-truth projector ray_tracing(interpolator, ...) — phantomdata/helpers.py:192-224 — for a voxel volume
 * vol[nx,ny,nz] (C order) on the regular grid axis_k = origin_k + i*spacing_k, trilinear interpolation with
 * `fill_value` outside (scipy RegularGridInterpolator(method='linear', bounds_error=False, fill_value) as built by
 * get_interpolator_from_* :72-154).  Rays and depths come from `args` (ray_mode, z[S] shared, pixel out [R]).
 * type_ct != 0: img = prod exp(-mu*dz*||d||), last dz = 1e10;  type_ct == 0: img = prod exp(-mu). */
int afx_project_volume(const float* vol, int32_t nx, int32_t ny, int32_t nz, const double origin[3], const double spacing[3],
                       float fill_value, const afx_render_args* args, int type_ct, void* stream);

/* ---- Occupancy-grid acceleration of the training loop (nerf/run_nerf_acc.py:196-198,284-287;
 * nerf/nerf_helpers_acc.py:10-31,65-78).  Upstream these are nerfacc 0.3.x calls (OccupancyGrid.every_n_step,
 * ray_marching with alpha_fn, render_visibility); nerfacc is neither vendored nor pinned by the reference and absent
 * here, so the entry points implement its published algorithm (parity unpinned at that boundary; restated for the
 * tests in oracle/).  The MLP evaluations between the steps are afx_mlp_infer calls on the points emitted here;
 * exclusive scans of the per-ray counts are the caller's (one cumsum).  All buffers are the caller's. */
typedef struct afx_grid_desc {
  float roi_aabb[6];            /* xmin ymin zmin xmax ymax zmax */
  int32_t resolution[3];
} afx_grid_desc;

/* OccupancyGrid._update, step 1: x = (cell + u) / resolution * (hi - lo) + lo for the n selected cells (cell_idx NULL:
 * cells 0..n-1).  jitter[n,3] in [0,1) supplied by the caller (parity mode), or NULL: drawn in-kernel from the
 * counter-based Philox4x32-10 stream (seed, stream_id) (perf mode). */
int afx_grid_points(const afx_grid_desc* grid, const int32_t* cell_idx, int64_t n, const float* jitter, uint64_t seed,
                    uint64_t stream_id, float* pts, void* stream);
/* step 2: occs[c] = max(occs[c] * ema_decay, occ_new) for the selected cells; deterministic when a cell is selected more
 * than once (max over its draws).  occs_scratch: n_cells floats (snapshot of occs). */
int afx_grid_update(const afx_grid_desc* grid, float* occs, float* occs_scratch, const int32_t* cell_idx, int64_t n,
                    const float* occ_new, float ema_decay, void* stream);
/* step 3: binary[c] = occs[c] > min(mean(occs), occ_thre) as bytes (the module's `binary` tensor) and as the packed
 * bitfield the march reads (n_cells/32 words, rounded up).  partial_ws: 256 doubles. */
int afx_grid_binarize(const afx_grid_desc* grid, const float* occs, float occ_thre, uint8_t* binary, uint32_t* bits,
                      double* partial_ws, void* stream);

/* Restoring a trained grid: the reference assigns `acc_grid._binary = grid_occupancy` (visualization/visualization.py:162)
 * before query_occ / acc_ray_marching.  Packs a caller-supplied byte mask binary[n_cells] (non-zero = occupied) into the
 * bitfield the march reads. */
int afx_grid_pack(const afx_grid_desc* grid, const uint8_t* binary, uint32_t* bits, void* stream);

/* nerfacc.ray_marching: t range = ray / scene_aabb intersection clipped to [near, far]; fixed-step lattice
 * t_min + k*step; a step belongs to the ray while its mid-point lies before t_max (nerfacc marches `while (t_mid < far)`) and
 * is kept when the cell holding its mid-point is occupied (grid_bits NULL: every step). */
typedef struct afx_march_args {
  const float* origins;         /* [R,3] */
  const float* dirs;            /* [R,3] */
  int64_t n_rays;
  int32_t has_aabb;  float scene_aabb[6];
  int32_t has_near, has_far;  float near_plane, far_plane;
  float step;                   /* render_step_size */
  const uint32_t* grid_bits;    /* from afx_grid_binarize, or NULL */
  afx_grid_desc grid;
} afx_march_args;
int afx_march_count(const afx_march_args* args, int32_t* counts, void* stream);                    /* kept steps per ray */
/* packed, ray-sorted samples at offsets = exclusive scan of the counts; mid_points optional [n,3] = o + d*(t_s+t_e)/2 */
int afx_march_write(const afx_march_args* args, const int64_t* offsets, int32_t* ray_indices, float* t_starts,
                    float* t_ends, float* mid_points, void* stream);
/* alpha_fn of nerf_helpers_acc.py:11-25 on the raw MLP output of the candidates + nerfacc's render_visibility: steps with
 * alpha < alpha_thre are dropped without attenuating the transmittance, the ray ends once it falls below early_stop_eps.
 * input_is_alpha != 0: `raw` already holds the caller's alpha_fn values.  offsets[R+1]; keep[n] in {0,1}; counts[r] = kept steps. */
int afx_march_visibility(const float* raw, int32_t input_is_alpha, const float* t_starts, const float* t_ends, const int64_t* offsets,
                         int64_t n_rays, float early_stop_eps, float alpha_thre, uint8_t* keep, int32_t* counts, void* stream);
/* counts[R] (afx_march_count / afx_march_visibility) -> offsets[R+1], the exclusive prefix sums both take back; optionally (non-null)
 * group_offsets[R+1], the same over ceil(count / 32) - the offsets of the group-aligned copy afx_pack_groups lays out - and
 * totals[2] = {samples, groups} for the caller's one host read.  One launch in place of the caller's zeros / cumsum sequence. */
int afx_ray_offsets(const int32_t* counts, int64_t n_rays, int64_t* offsets, int64_t* group_offsets, int64_t* totals, void* stream);
int afx_march_compact(const uint8_t* keep, const int64_t* offsets_in, const int64_t* offsets_out, int64_t n_rays,
                      const float* t_starts_in, const float* t_ends_in, int32_t* ray_indices_out, float* t_starts_out,
                      float* t_ends_out, void* stream);

/* The whole grid iteration of nerf/run_nerf_acc.py:284-306 in ONE call: afx_march_count / afx_ray_offsets / afx_march_write (candidates),
 * afx_mlp_infer at the candidates' mid-points + afx_march_visibility (the reference's alpha_fn + nerfacc's render_visibility),
 * afx_march_compact / afx_pack_groups, afx_train_step_packed_mse - the same entry points in the same order, so the results are those of
 * the call-by-call sequence bit for bit.  (At the reference's batch the GPU is busy for 0.27 ms of an iteration; the ~30 launches cost more
 * when a Python loop issues them.)  Two host read-backs inside (sizes are data), so the call is NOT graph-capturable.  Every array
 * between the steps lives in `workspace`; when it is too small the call returns AFX_E_WORKSPACE with `workspace_needed` set (nothing the
 * caller owns has been written) - grow and call again.  n_kept == 0 on return: no sample survived, pixel / grad_flat untouched (the
 * reference skips the optimizer step, :293). */
typedef struct afx_march_train_args {
  afx_march_args march;             /* rays, scene box, planes, step, occupancy bits: as for afx_march_count */
  float early_stop_eps, alpha_thre; /* render_visibility thresholds (run_nerf_acc.py:68-70) */
  const float* target;              /* [R] */
  float inv_n;                      /* 1 / rays of the global batch */
  float* pixel;                     /* [R] out */
  float* grad_flat;                 /* += dL/dparams */
  void* workspace; size_t workspace_bytes;
  int64_t n_candidates, n_kept, n_groups;      /* out */
  size_t workspace_needed;                     /* out, with AFX_E_WORKSPACE */
} afx_march_train_args;
int afx_march_train_step_mse(afx_ctx* ctx, int prec, const void* prepared, afx_march_train_args* args, void* stream);

/* Indices of the k largest of keys[n] (ties: lowest index first), written in ASCENDING INDEX order - the selection step of the
 * weighted ray sampler (the batch of nerf/nerf_helpers.py:137-150 is a set; its order carries no meaning).  Radix select:
 * three histogram passes + a counted compaction, deterministic, no full sort.  workspace: afx_topk_workspace_bytes(n) bytes
 * of device memory; n < 2^32. */
size_t afx_topk_workspace_bytes(int64_t n);
int afx_topk_indices(const float* keys, int64_t n, int64_t k, int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream);

/* The ray batches of `n_batches` consecutive training iterations in one launch sequence: out_idx[b][k] = what afx_sample_keys(weights, n,
 * u = NULL, seed, stream_id0 + b) followed by afx_topk_indices(k) returns, bit for bit (sample_pixel_rays draws one batch per iteration,
 * nerf/run_nerf_acc.py:277; a draw is 11 launches of a few microseconds - launch latency, 5 % of the reference's 1.3 ms iteration - and the
 * draws of different iterations are independent, so they share the launches: blockIdx.y = b).  workspace:
 * afx_sample_batches_workspace_bytes(n, n_batches) bytes of device memory (the keys of every batch: n_batches x n floats); n < 2^32,
 * n_batches <= 65535. */
size_t afx_sample_batches_workspace_bytes(int64_t n, int32_t n_batches);
int afx_sample_batches(const float* weights, int64_t n, uint64_t seed, uint64_t stream_id0, int32_t n_batches, int64_t k, int64_t* out_idx,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- Device-resident ray batches (sample_pixel_rays, nerf/nerf_helpers.py:137-150: weighted sampling without
 * replacement over all pixels of all training projections).  keys[i] = log(u_i) / w_i (Efraimidis-Spirakis): the k
 * largest keys are a weighted sample without replacement; u[n] supplied, or NULL: Philox stream (seed, stream_id).
 * afx_gather_rays copies the selected rows of the resident ray table. */
int afx_sample_keys(const float* weights, int64_t n, const float* u, uint64_t seed, uint64_t stream_id, float* keys, void* stream);
int afx_gather_rays(const float* origins, const float* dirs, const float* pixels, const int64_t* idx, int64_t k,
                    float* origins_out, float* dirs_out, float* pixels_out, void* stream);
/* out[i] = uniform [0,1) number i of Philox4x32-10 stream (seed, stream_id): the generator of every "perf mode" draw */
int afx_philox_uniform(uint64_t seed, uint64_t stream_id, int64_t n, float* out, void* stream);

/* Trainable fourier coefficients (model/CPPN.py:92 makes them an nn.Parameter; fourier_pos_enc, CPPN.py:320-327, is
 * differentiable in them).  After this call every backward entry point (afx_mlp_backward, afx_render_backward,
 * afx_train_step_mse) at a 16-bit precision also does d_enc_aux[3*n_freq] += d loss / d coefficients; `params` is the
 * fp32 flat parameter vector the prepared buffer was made from (W_0 is read from it).  d_enc_aux = NULL switches it
 * off (the default: the coefficients are constants).  AFX_PREC_F32 backward calls fail while it is on. */
int afx_set_encoding_grad(afx_ctx* ctx, const float* params, float* d_enc_aux);

/* Measurement aid (bench.py's roofline leg): when enabled, every launch of the three MFMA kernels is
 * bracketed by HIP events recorded on the launch stream.  afx_profile_read blocks on those events
 * (the only call in this library that synchronises), returns the summed device time and the launch
 * count for one kernel kind, and forgets them. */
enum { AFX_K_CHAIN_FWD = 0, AFX_K_CHAIN_BWD = 1, AFX_K_WGRAD = 2 };
int afx_profile_enable(afx_ctx* ctx, int on);
int afx_profile_read(afx_ctx* ctx, int which, double* ms_total, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* AFX_H */
