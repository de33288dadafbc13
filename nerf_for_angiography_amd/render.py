"""Fused renderer: ray generation -> sampling -> CPPN -> Beer-Lambert product in one kernel pass per ray
chunk, with autograd.  This is the path that replaces the body of the reference's training/eval
iteration (nerf/run_nerf_acc.py:287-296, :340-349)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .engine import RenderSpec
from ._lib import AfxError


@dataclass
class RenderOutput:
    rgb_map: torch.Tensor                       # [R] transmittance = predicted pixel
    depth_map: Optional[torch.Tensor] = None    # only with want_aux (dense convention)
    weights: Optional[torch.Tensor] = None
    entropy: Optional[torch.Tensor] = None
    sigma: Optional[torch.Tensor] = None


# Optional hook set by nerf_for_angiography_amd.dist: called on the flat gradient before it is split.
_grad_hook = None


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, spec, want_st, *params):
        prepared = model._prepared()
        pixel, sigma, tau = model.engine.render_forward(prepared, spec, model.precision, want_sigma=want_st,
                                                        want_tau=want_st)
        ctx.model, ctx.spec = model, spec
        ctx.save_for_backward(pixel)
        if want_st:
            ctx.mark_non_differentiable(sigma, tau)
            return pixel, sigma, tau
        return pixel

    @staticmethod
    def backward(ctx, d_pixel, *unused):
        model, spec = ctx.model, ctx.spec
        (pixel,) = ctx.saved_tensors
        flat_grad = torch.zeros(model.engine.param_count, dtype=torch.float32, device=pixel.device)
        coef_grad = model._coef_grad_buffer()
        with model.engine.encoding_grad(model.flat_params, coef_grad):
            model.engine.render_backward(model._prepared(), spec, pixel, d_pixel.contiguous(), flat_grad, model.precision)
        if _grad_hook is not None:
            _grad_hook(flat_grad)
            if coef_grad is not None:
                _grad_hook(coef_grad)
        return (None, None, None) + model._fn_grads(flat_grad, coef_grad)


def _check_model(model):
    if not getattr(model, "fused", False):
        if getattr(model, "fused_forward", False) and not torch.is_grad_enabled():
            pass      # tanh / sine models: forward kernels only (evaluation renders, density grids under torch.no_grad())
        else:
            raise NotImplementedError("render_rays: this CPPN configuration is outside the fused kernels (ReLU - or tanh / sine under "
                                      "torch.no_grad() -, no skip block, no view directions, one output channel)")
    if model.flat_params is None or not model.flat_params.is_cuda:
        raise AfxError("render_rays: the model must live on a GPU; there is no CPU fallback")


def render_spec(spec: RenderSpec, model, want_aux: bool = False) -> RenderOutput:
    _check_model(model)
    if not want_aux:
        return RenderOutput(_RenderFn.apply(model, spec, False, *model._fn_params()))
    pixel, sigma, tau = _RenderFn.apply(model, spec, True, *model._fn_params())
    out = RenderOutput(pixel, sigma=sigma)
    if spec.mode == "dense":
        # weights / depth_map / entropy of render_volume_density (nerf_helpers.py:107-119) from the
        # per-sample optical depths the kernel wrote; cheap per-ray scans, no MLP work.
        alpha = torch.exp(-tau)
        excl = torch.cat([torch.ones_like(alpha[:, :1]), torch.cumprod(alpha, -1)[:, :-1]], -1)
        out.weights = (1 - alpha + 1e-10) * excl
        z = spec.z if spec.z.dim() == 2 else spec.z[None, :]
        out.depth_map = (alpha * z).sum(-1)
        dens = sigma / (sigma.sum(-1, keepdim=True) + 1e-10)
        ent = -(dens * torch.log(dens + 1e-10)).sum(-1)
        out.entropy = ent * ((1 - pixel.detach()) > 0.4)
    return out


def render_rays(model, ray_origins, ray_directions, depth_samples_per_ray: int = 0, near_thresh: float = 0.0,
                far_thresh: float = 0.0, mode: str = "acc", z: Optional[torch.Tensor] = None,
                want_aux: bool = False) -> RenderOutput:
    """Pixels for rays given as origin/direction arrays (the output of sample_pixel_rays).

    mode='acc'  : depth_samples_per_ray uniform steps in [near, far], mid-point evaluation, dt = step —
                  acc_ray_marching without a grid + acc_render_volume_density (nerf_helpers_acc.py:10-63).
    mode='dense': explicit depths z [S] or [R,S]; render_volume_density convention (nerf_helpers.py:59-123)."""
    n_rays = ray_origins.shape[0]
    if mode == "dense":
        depth_samples_per_ray = z.shape[-1]
    spec = RenderSpec(n_rays=n_rays, n_samples=int(depth_samples_per_ray), origins=ray_origins, dirs=ray_directions,
                      mode=mode, t_near=float(near_thresh), t_far=float(far_thresh), z=z)
    return render_spec(spec, model, want_aux)


def render_projection(model, poses, width: int, height: int, focal: float, depth_samples_per_ray: int,
                      near_thresh: float, far_thresh: float, ray_ids: Optional[torch.Tensor] = None,
                      ray_id0: int = 0, n_rays: Optional[int] = None, mode: str = "acc",
                      z: Optional[torch.Tensor] = None, want_aux: bool = False) -> RenderOutput:
    """Pixels for rays generated in-kernel from C-arm poses (get_ray_values, phantomdata/helpers.py:156-175).
    poses: float64 [n_proj,4,4] or [n_proj,3,4] cam->world (source_matrix); rays are indexed into
    [n_proj, H, W] by `ray_ids` (int32) or enumerated from `ray_id0`."""
    poses = poses[:, :3, :].contiguous()
    if n_rays is None:
        n_rays = ray_ids.numel() if ray_ids is not None else poses.shape[0] * width * height - ray_id0
    if mode == "dense":
        depth_samples_per_ray = z.shape[-1]
    spec = RenderSpec(n_rays=int(n_rays), n_samples=int(depth_samples_per_ray), poses=poses, ray_ids=ray_ids,
                      ray_id0=int(ray_id0), width=int(width), height=int(height), focal=float(focal), mode=mode,
                      t_near=float(near_thresh), t_far=float(far_thresh), z=z)
    return render_spec(spec, model, want_aux)


def _as_f32(t, device):
    return t.to(device=device, dtype=torch.float32)


def _global_rays(n_local: int, n_global: Optional[int], device) -> int:
    """The ray count the loss is the mean over.  An explicit n_global wins; with a multi-rank SUM gradient hook installed
    (dist.GradSync) the ranks' counts are all-reduced, so that a default call is still the gradient of the GLOBAL mean; a
    local-mean hook (GradSync(local_mean=True)) keeps the local count; a foreign multi-rank hook must be told."""
    if n_global:
        return int(n_global)
    if _grad_hook is not None and getattr(_grad_hook, "world", 1) > 1:
        if getattr(_grad_hook, "local_mean", False):
            return int(n_local)
        if hasattr(_grad_hook, "global_count"):
            return _grad_hook.global_count(int(n_local), device)
        raise AfxError("train_step_mse: a multi-rank gradient hook is installed; pass n_global (rays of the step over all ranks)")
    return int(n_local)


def train_step_mse(model, spec: RenderSpec, target: torch.Tensor, n_global: Optional[int] = None):
    """One fused training pass: render `spec`, L = mean over the (global) batch of (pixel - target)^2, backward.

    Replaces `pred = render(...); loss = mse_loss(pred, target); loss.backward()` (nerf/run_nerf_acc.py:287-306):
    the gradients are ACCUMULATED into `.grad` of the model's Linear parameters exactly as loss.backward() would,
    so `optimizer.zero_grad(); train_step_mse(...); optimizer.step()` is the training iteration.  The forward pass
    is the backward kernel's own forward (f16 / bf16 operands), nothing is rendered twice.  Returns (loss, pixels), detached.
    n_global: total rays of the step across all ranks - the mean is over that count.  Default: this batch; with a
    multi-rank gradient hook installed (dist.GradSync, a SUM all-reduce) the ranks' ray counts are all-reduced instead, so
    the default call stays the gradient of the global mean."""
    _check_model(model)
    if model.precision == "f32":
        raise NotImplementedError("train_step_mse needs a 16-bit precision (f16, bf16, bf16x3); with 'f32' use render + autograd")
    n = _global_rays(spec.n_rays, n_global, model.flat_params.device)
    flat_grad = torch.zeros(model.engine.param_count, dtype=torch.float32, device=model.flat_params.device)
    coef_grad = model._coef_grad_buffer()
    with model.engine.encoding_grad(model.flat_params, coef_grad):
        if model.engine.fused_step_available(spec.n_samples, model.precision):
            # one kernel per ray chunk - or, for rays that straddle workgroup tiles (the reference's 300 samples/ray, the 128 + 64 of
            # the hierarchical pass) at the default precision, its two halves with the per-ray reduction between them: no recompute
            pixel = model.engine.train_step_mse(model._prepared(), spec, target, 1.0 / n, flat_grad, model.precision)
        else:
            # such rays at the other precisions / with an input encoding: forward launch, then the backward kernel (which recomputes
            # the forward) with dL/dpixel = 2 (pixel - target) / n
            pixel, _, _ = model.engine.render_forward(model._prepared(), spec, model.precision)
            d_pixel = (pixel - _as_f32(target, pixel.device)) * (2.0 / n)
            model.engine.render_backward(model._prepared(), spec, pixel, d_pixel, flat_grad, model.precision)
    if _grad_hook is not None:
        _grad_hook(flat_grad)
        if coef_grad is not None:
            _grad_hook(coef_grad)
    for p, g in zip(model._fn_params(), model._fn_grads(flat_grad, coef_grad)):
        if p.grad is None:
            p.grad = g
        else:
            p.grad.add_(g)
    loss = torch.nn.functional.mse_loss(pixel, target) if n == spec.n_rays else ((pixel - target) ** 2).sum() / n
    return loss, pixel


def train_step_packed_mse(model, ray_origins, ray_directions, packed, target: torch.Tensor, n_global: Optional[int] = None):
    """The reference's iteration body behind the occupancy-grid march - positions, get_predictions, acc_render_volume_density,
    mse_loss, backward (nerf/run_nerf_acc.py:289-306) - as ONE fused pass over the march's packed samples (engine.PackedGroups, from
    `occupancy.ray_marching(..., return_packed=True)` or `engine.pack_groups`): forward half, per-ray transmittance product, backward
    half, weight gradients; the MLP is evaluated once (the operator sequence evaluates it in the forward and again inside backward).
    f16s8 precision (with or without an input encoding).  Gradients are ACCUMULATED into `.grad` as loss.backward() would.  Returns (loss, pixels[n_rays]);
    a ray without samples renders 1 (the empty product)."""
    _check_model(model)
    if model.precision != "f16s8":
        raise NotImplementedError("train_step_packed_mse: precision 'f16s8' (other precisions: the operator sequence "
                                  "get_predictions -> acc_render_volume_density -> mse_loss -> backward)")
    n = _global_rays(packed.n_rays, n_global, model.flat_params.device)
    flat_grad = torch.zeros(model.engine.param_count, dtype=torch.float32, device=model.flat_params.device)
    coef_grad = model._coef_grad_buffer()
    with model.engine.encoding_grad(model.flat_params, coef_grad):
        pixel = model.engine.train_step_packed_mse(model._prepared(), ray_origins, ray_directions, packed, target, 1.0 / n, flat_grad, model.precision)
    if _grad_hook is not None:
        _grad_hook(flat_grad)
        if coef_grad is not None:
            _grad_hook(coef_grad)
    for p, g in zip(model._fn_params(), model._fn_grads(flat_grad, coef_grad)):
        if p.grad is None:
            p.grad = g
        else:
            p.grad.add_(g)
    loss = torch.nn.functional.mse_loss(pixel, target) if n == packed.n_rays else ((pixel - target) ** 2).sum() / n      # (one launch, not four)
    return loss, pixel


def march_train_step_mse(model, grid, scene_aabb, ray_origins, ray_directions, depth_samples_per_ray: int, near_thresh: float, far_thresh: float,
                         early_stop_eps: float, alpha_thre: float, target: torch.Tensor, n_global: Optional[int] = None):
    """The reference's whole grid iteration - acc_ray_marching (march through the occupancy grid, alpha_fn pass, render_visibility) followed by the
    body (positions, get_predictions, acc_render_volume_density, mse_loss, backward; nerf/run_nerf_acc.py:284-306) - as ONE library call
    (afx_march_train_step_mse): what `nerf_helpers_acc.acc_ray_marching(..., return_packed=True)` + `train_step_packed_mse` do, entry point for
    entry point and bit for bit, without a Python round trip per launch.  `grid`: nerf.occupancy.OccupancyGrid or None.  f16s8.
    Gradients are ACCUMULATED into `.grad`.  Returns (loss, pixels[n_rays], n_kept) - (None, None, 0) when no sample survived the march (the
    reference then skips the optimizer step, :293)."""
    _check_model(model)
    if model.precision != "f16s8":
        raise NotImplementedError("march_train_step_mse: precision 'f16s8' (other precisions: acc_ray_marching + the operator sequence)")
    from .nerf.occupancy import _aabb_on_host
    n_rays = ray_origins.shape[0]
    n = _global_rays(n_rays, n_global, model.flat_params.device)
    flat_grad = torch.zeros(model.engine.param_count, dtype=torch.float32, device=model.flat_params.device)
    coef_grad = model._coef_grad_buffer()
    step = (float(far_thresh) - float(near_thresh)) / int(depth_samples_per_ray)
    with model.engine.encoding_grad(model.flat_params, coef_grad):
        pixel, _, n_kept = model.engine.march_train_step_mse(
            model._prepared(), ray_origins, ray_directions, target, 1.0 / n, flat_grad, model.precision,
            None if scene_aabb is None else _aabb_on_host(scene_aabb), near_thresh, far_thresh, step, early_stop_eps, alpha_thre,
            grid_bits=None if grid is None else grid.bits, grid_aabb=None if grid is None else grid._aabb_host,
            grid_res=None if grid is None else grid._res_host)
    if n_kept == 0:
        return None, None, 0
    if _grad_hook is not None:
        _grad_hook(flat_grad)
        if coef_grad is not None:
            _grad_hook(coef_grad)
    for p, g in zip(model._fn_params(), model._fn_grads(flat_grad, coef_grad)):
        if p.grad is None:
            p.grad = g
        else:
            p.grad.add_(g)
    loss = torch.nn.functional.mse_loss(pixel, target) if n == n_rays else ((pixel - target) ** 2).sum() / n
    return loss, pixel, n_kept


def hierarchical_train_step_mse(model, ray_origins, ray_directions, depth_values, depth_samples_per_ray_fine: int,
                                target: torch.Tensor, u: Optional[torch.Tensor] = None, n_global: Optional[int] = None,
                                fine_model=None, reuse_coarse: bool = True):
    """One hierarchical (coarse + fine) training pass on the fused kernels - the training step `fine_sampling`
    (nerf/nerf_helpers.py:178-195) belongs to, for the one-channel absorption model and an MSE loss on the fine render:

      coarse pass, no gradient (the reference detaches the samples, :186): ONE forward launch over the S coarse depths, which
        leaves the per-sample optical depths tau[R,S];
      afx_fine_depths_from_tau: weights = (1 - alpha + 1e-10) cumprod_exclusive(alpha) per ray, inverse-CDF sampling of
        `depth_samples_per_ray_fine` depths from weights[..., 1:-1] over the mid-point bins (sample_pdf, :197-222), merge with
        the coarse depths - nothing [R,S]-shaped is formed by torch operators in between;
      fine pass over the S + N_f per-ray depths with `fine_model or model`: train_step_mse (dense convention) - forward,
        compositing, MSE gradient and backward without recomputing the forward (128 + 64 = 192 samples straddle the 256-sample
        workgroup tiles: the split-phase step of afx_train_step_mse at the f16s8 precision).

    With ONE network (fine_model None, the reference's `fine_model or coarse_model`), f16s8 and no input encoding, the coarse depths are not evaluated
    twice (`reuse_coarse`, default on): the coarse pass IS the forward half of the training kernel over the coarse depths (stash, masks, sigma, tau), a
    second forward half evaluates only the N_f new depths, a per-ray kernel composites the merged list and hands every sample its finished dL/draw, and the
    backward halves + weight gradients of both sets follow (afx_hier_train_step_mse) - a third less MLP work than coarse forward + fine step.

    Gradients are accumulated into `.grad` of the fine network as train_step_mse does.  Returns (loss, fine pixels, merged depths)."""
    _check_model(model)
    from . import engine as _engine
    n_rays = ray_origins.shape[0]
    z = depth_values
    if (reuse_coarse and fine_model is None and model.precision == "f16s8" and model.engine.enc == "none"
            and int(depth_samples_per_ray_fine) >= 2 and 3 <= int(z.shape[-1]) <= 512):
        if u is None:
            u = torch.rand(n_rays, int(depth_samples_per_ray_fine), device=model.flat_params.device)
        n = _global_rays(n_rays, n_global, model.flat_params.device)
        flat_grad = torch.zeros(model.engine.param_count, dtype=torch.float32, device=model.flat_params.device)
        spec_c = RenderSpec(n_rays=n_rays, n_samples=int(z.shape[-1]), origins=ray_origins, dirs=ray_directions, mode="dense", z=z)
        pixel, z_all = model.engine.hier_train_step_mse(model._prepared(), spec_c, int(depth_samples_per_ray_fine), u, target, 1.0 / n, flat_grad,
                                                        model.precision)
        if _grad_hook is not None:
            _grad_hook(flat_grad)
        for p, g in zip(model._fn_params(), model._fn_grads(flat_grad, None)):
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        return ((pixel - target) ** 2).sum() / n, pixel, z_all
    with torch.no_grad():
        spec_c = RenderSpec(n_rays=n_rays, n_samples=int(z.shape[-1]), origins=ray_origins, dirs=ray_directions, mode="dense", z=z)
        _, _, tau = model.engine.render_forward(model._prepared(), spec_c, model.precision, want_tau=True)
        if u is None:
            u = torch.rand(n_rays, int(depth_samples_per_ray_fine), device=tau.device)
        z_all = _engine.fine_depths_from_tau(z, tau, u)
        del tau
    net = model if fine_model is None else fine_model
    spec_f = RenderSpec(n_rays=n_rays, n_samples=int(z_all.shape[-1]), origins=ray_origins, dirs=ray_directions, mode="dense", z=z_all)
    loss, pixel = train_step_mse(net, spec_f, target, n_global)
    return loss, pixel, z_all


def projection_spec(poses, width, height, focal, depth_samples_per_ray, near_thresh, far_thresh, ray_ids=None,
                    ray_id0=0, n_rays=None) -> RenderSpec:
    """RenderSpec for rays generated in-kernel from C-arm poses ('acc' convention)."""
    poses = poses[:, :3, :].contiguous()
    if n_rays is None:
        n_rays = ray_ids.numel() if ray_ids is not None else poses.shape[0] * width * height - ray_id0
    return RenderSpec(n_rays=int(n_rays), n_samples=int(depth_samples_per_ray), poses=poses, ray_ids=ray_ids,
                      ray_id0=int(ray_id0), width=int(width), height=int(height), focal=float(focal), mode="acc",
                      t_near=float(near_thresh), t_far=float(far_thresh))


GRID_PRECISION = "bf16x3"      # default arithmetic of density_grid: split bf16 - fp32-grade (the grid's 1e-4 bar) at 1/3 of the f16 rate


def grid_precision(model) -> str:
    """Arithmetic of a density-grid evaluation: the model's own precision when it is a strict one (f32, bf16x3), else split bf16."""
    if getattr(model, "_act_name", "relu") == "sine":
        return "f32"      # sin(w0 z) amplifies the first layer's operand rounding by w0: split bf16 measures 1e-3 at w0 = 15
    return model.precision if model.precision in ("f32", "bf16x3") else GRID_PRECISION


def density_grid(model, outside: float, n: int, precision: Optional[str] = None) -> torch.Tensor:
    """sigma on meshgrid(t,t,t), t = linspace(-outside, outside, n+1), numpy 'xy' indexing as upstream
    (visualization/visualization.py:100-102,209-229): grid[i,j,k] = sigma(t[j], t[i], t[k]).

    precision: arithmetic of the MLP evaluation.  Default: the model's precision if that is a strict one (f32, bf16x3),
    otherwise GRID_PRECISION (split bf16) whatever the model trains at: the reconstructed grid is held to 1e-4 relative L2 against the reference's fp32 path, which the training
    precisions miss (f16: ~1e-3 on sigmoid(raw), tests/test_gpu_round3.py) - pixels average that error over a ray, a
    grid cell does not.  201^3 points take ~25 ms in split bf16."""
    with torch.no_grad():
        _check_model(model)
    dev = model.flat_params.device
    t = torch.linspace(-outside, outside, n + 1, dtype=torch.float64, device=dev).float()
    gy, gx, gz = torch.meshgrid(t, t, t, indexing="ij")      # [i,j,k] -> (x=t[j], y=t[i], z=t[k])
    pts = torch.stack([gx, gy, gz], -1).reshape(-1, 3).contiguous()
    keep, model.precision = model.precision, (precision or grid_precision(model))
    try:
        with torch.no_grad():
            sig = model.engine.infer(model._prepared(), pts, model.precision, apply_sigmoid=True)
    finally:
        model.precision = keep
    return sig.reshape(n + 1, n + 1, n + 1)
