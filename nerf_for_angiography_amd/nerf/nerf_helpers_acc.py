"""Mirror of the reference's nerf/nerf_helpers_acc.py.  Upstream these wrap nerfacc 0.3.x and torch_scatter
(neither vendored nor pinned upstream, absent here); the dense no-grid march and the packed Beer-Lambert
product are restated and run in the HIP library (parity unpinned at that third-party boundary)."""
import torch

from .. import engine as _engine


def acc_ray_marching(radiance_field, grid, scene_aabb, ray_origins, ray_directions, depth_samples_per_ray,
                     near_thresh, far_thresh, early_stop_eps=1e-2, alpha_thre=1e-3, return_packed=False):
    """nerf/nerf_helpers_acc.py:10-31 -> (ray_indices[n], t_starts[n,1], t_ends[n,1]), packed and ray-sorted.

    grid=None and scene_aabb=None: every ray is marched with the fixed step (far-near)/depth_samples_per_ray from
    near_thresh, no pruning (the dense variant of model/nerf_helpers_acc.py:29).  Otherwise the occupancy-grid march
    of nerf/occupancy.py (nerfacc 0.3.x's published algorithm on HIP kernels; parity unpinned) with the reference's
    alpha_fn: sigmoid density at the interval mid-point, alpha = 1 - exp(-sigma * dt)."""
    render_step_size = (far_thresh - near_thresh) / depth_samples_per_ray
    if grid is None and scene_aabb is None:
        dev = ray_origins.device
        n_rays = ray_origins.shape[0]
        i = torch.arange(depth_samples_per_ray, dtype=torch.float32, device=dev)
        t_s = torch.tensor(near_thresh, dtype=torch.float32, device=dev) + i * torch.tensor(render_step_size, dtype=torch.float32, device=dev)
        t_e = t_s + torch.tensor(render_step_size, dtype=torch.float32, device=dev)
        ray_indices = torch.arange(n_rays, dtype=torch.int32, device=dev).repeat_interleave(depth_samples_per_ray)
        return ray_indices, t_s.repeat(n_rays)[:, None], t_e.repeat(n_rays)[:, None]

    # alpha_fn of the reference (nerf_helpers_acc.py:11-25): sigmoid density at the interval mid-point,
    # alpha = 1 - exp(-sigma * dt).  The march kernel emits the mid-points, the visibility kernel forms alpha from the
    # raw MLP output: only the fused MLP launch sits between the two.
    from .occupancy import ray_marching
    return ray_marching(ray_origins, ray_directions, scene_aabb=scene_aabb, grid=grid, raw_fn=radiance_field,
                        near_plane=near_thresh, far_plane=far_thresh, early_stop_eps=early_stop_eps,
                        alpha_thre=alpha_thre, render_step_size=render_step_size, return_packed=return_packed)


def get_ray_entropy(sigmas, rgb_map, threshold=0.4):
    weights_sum = torch.sum(sigmas, dim=-1)
    ray_density = sigmas / (weights_sum.unsqueeze(1) + 1e-10)
    ray_entropy = -torch.sum(ray_density * torch.log(ray_density + 1e-10), dim=-1)
    return ray_entropy * ((1 - rgb_map) > threshold).detach()


class _PackedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, ri, ts, te, n_rays):
        rgb = _engine.composite_packed(pred, ri, ts, te, n_rays)
        ctx.save_for_backward(pred, ri, ts, te, rgb)
        ctx.n_rays = n_rays
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        pred, ri, ts, te, rgb = ctx.saved_tensors
        return _engine.composite_packed_backward(pred, ri, ts, te, ctx.n_rays, rgb, d_rgb.contiguous()), None, None, None, None


def acc_render_volume_density(predictions, ray_indices, t_starts, t_ends, n_rays, depth_samples_per_ray, zero_idx=[]):
    """nerf/nerf_helpers_acc.py:45-63 -> (rgb_map[n_rays], entropy=None):
    rgb_map[r] = prod_{i in ray r} exp(-sigmoid(pred_i) * (t_end_i - t_start_i)); ray_indices sorted."""
    pred = predictions.reshape(-1).float()
    if len(zero_idx) > 0:       # sigma forced to 0  <=>  raw -> -inf
        pred = pred.clone()
        pred[zero_idx] = -float("inf")
    if not pred.is_cuda:
        from .._lib import AfxError
        raise AfxError("acc_render_volume_density: tensors must live on the GPU; there is no CPU fallback (tests compare with oracle/)")
    rgb = _PackedFn.apply(pred.contiguous(), ray_indices.to(torch.int32).contiguous(),
                          t_starts.reshape(-1).float().contiguous(), t_ends.reshape(-1).float().contiguous(), int(n_rays))
    return rgb, None


def acc_update_n_step(acc_grid, radiance_field, step, occ_thre=1e-2, inverse=False):
    """nerf/nerf_helpers_acc.py:65-78: refresh the occupancy grid every 16th step from sigmoid(MLP)."""
    if acc_grid is None:
        return acc_grid

    def occ_eval_fn(x):
        return torch.sigmoid(radiance_field(x))

    acc_grid.every_n_step(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre)
    return acc_grid
