"""Occupancy-grid acceleration — the part of nerfacc 0.3.x the reference uses (nerf/run_nerf_acc.py:196-198,284-287;
nerf/nerf_helpers_acc.py:10-31,65-78; visualization/visualization.py:214).

nerfacc is a third-party CUDA package that the reference neither vendors nor pins and that is absent here, so this
is a restatement of its PUBLISHED algorithm (0.3.x line: `OccupancyGrid`, `ContractionType.AABB`, `ray_marching`,
`render_visibility`) — PARITY UNPINNED: no reference output exists to compare with.  Semantics kept:

  OccupancyGrid(roi_aabb, resolution=128): `occs` float [res^3], `binary` bool [res,res,res];
    every_n_step(step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16): every n-th training step
    evaluate occupancy at one uniformly jittered point per selected cell (all cells during warm-up, afterwards
    num_cells/4 uniformly drawn cells plus num_cells/4 drawn among the occupied ones),
    occs[i] = max(occs[i] * ema_decay, occ), binary = occs > min(mean(occs), occ_thre).
  ray_marching: t range = ray/AABB intersection clipped to [near, far]; fixed-step lattice t_min + k*dt;
    a step is kept when the cell containing its mid-point is occupied; with alpha_fn, steps with
    alpha < alpha_thre or transmittance (exclusive product of 1-alpha over the ray's kept steps) < early_stop_eps
    are dropped.  Returns packed, ray-sorted (ray_indices, t_starts[n,1], t_ends[n,1]).

The MLP evaluations inside (occ_eval_fn / alpha_fn) run in the fused HIP MLP kernel through the model's forward."""
from __future__ import annotations

import torch


class ContractionType:
    AABB = 0


class OccupancyGrid(torch.nn.Module):
    NUM_DIM = 3

    def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB):
        super().__init__()
        if contraction_type != ContractionType.AABB:
            raise NotImplementedError("only ContractionType.AABB is used by the reference")
        res = [resolution] * 3 if isinstance(resolution, int) else list(resolution)
        self.register_buffer("_roi_aabb", torch.as_tensor(roi_aabb, dtype=torch.float32).flatten())
        self.register_buffer("resolution", torch.tensor(res, dtype=torch.int32))
        self.num_cells = int(res[0] * res[1] * res[2])
        self.register_buffer("occs", torch.zeros(self.num_cells))
        self.register_buffer("_binary", torch.zeros(res, dtype=torch.bool))
        g = torch.stack(torch.meshgrid([torch.arange(r) for r in res], indexing="ij"), -1).reshape(-1, 3)
        self.register_buffer("grid_coords", g)
        self.register_buffer("grid_indices", torch.arange(self.num_cells))
        self.contraction_type = contraction_type

    @property
    def roi_aabb(self):
        return self._roi_aabb

    @property
    def binary(self):
        return self._binary

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n):
        uniform = torch.randint(self.num_cells, (n,), device=self.occs.device)
        occupied = torch.nonzero(self._binary.flatten())[:, 0]
        if n < len(occupied):
            occupied = occupied[torch.randint(len(occupied), (n,), device=self.occs.device)]
        return torch.cat([uniform, occupied], dim=0)

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre=0.01, ema_decay=0.95, warmup_steps=256):
        if step < warmup_steps:
            indices = self.grid_indices
        else:
            indices = self._sample_uniform_and_occupied_cells(self.num_cells // 4)
        coords = self.grid_coords[indices]
        x = (coords + torch.rand_like(coords, dtype=torch.float32)) / self.resolution
        lo, hi = self._roi_aabb[:3], self._roi_aabb[3:]
        x = x * (hi - lo) + lo                                     # un-contract (AABB): unit cube -> world
        occ = occ_eval_fn(x).reshape(-1)
        self.occs[indices] = torch.maximum(self.occs[indices] * ema_decay, occ)
        self._binary = (self.occs > torch.clamp(self.occs.mean(), max=occ_thre)).view(self._binary.shape)

    @torch.no_grad()
    def every_n_step(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
        if not self.training:
            raise RuntimeError("every_n_step() is a training-time call; in eval mode use the grid as is")
        if step % n == 0:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)

    @torch.no_grad()
    def query_occ(self, samples):
        """Occupancy (0/1) at world points [P,3]; points outside the ROI are empty (visualization.py:214)."""
        idx, inside = _cell_index(samples, self._roi_aabb, self.resolution)
        out = torch.zeros(samples.shape[0], dtype=torch.bool, device=samples.device)
        out[inside] = self._binary.flatten()[idx[inside]]
        return out


def _cell_index(pts, aabb, resolution):
    lo, hi = aabb[:3], aabb[3:]
    u = (pts - lo) / (hi - lo)
    inside = ((u >= 0) & (u < 1)).all(-1)
    res = resolution.to(pts.device)
    ijk = torch.minimum((u * res).floor().long().clamp(min=0), (res - 1).long())
    return (ijk[:, 0] * res[1] + ijk[:, 1]) * res[2] + ijk[:, 2], inside


@torch.no_grad()
def ray_aabb_intersect(rays_o, rays_d, aabb):
    """Slab test -> (t_min, t_max); rays that miss get t_min = t_max = 1e10 (nerfacc's convention)."""
    inv = 1.0 / torch.where(rays_d == 0, torch.full_like(rays_d, 1e-12), rays_d)
    t0, t1 = (aabb[:3] - rays_o) * inv, (aabb[3:] - rays_o) * inv
    tmin = torch.minimum(t0, t1).amax(-1)
    tmax = torch.maximum(t0, t1).amin(-1)
    miss = tmax < torch.clamp(tmin, min=0)
    tmin, tmax = torch.clamp(tmin, min=0), tmax
    return torch.where(miss, torch.full_like(tmin, 1e10), tmin), torch.where(miss, torch.full_like(tmax, 1e10), tmax)


@torch.no_grad()
def ray_marching(rays_o, rays_d, scene_aabb=None, grid=None, alpha_fn=None, near_plane=None, far_plane=None,
                 early_stop_eps=1e-4, alpha_thre=0.0, render_step_size=1e-3):
    n_rays, dev = rays_o.shape[0], rays_o.device
    if scene_aabb is not None:
        t_min, t_max = ray_aabb_intersect(rays_o, rays_d, scene_aabb)
    else:
        t_min, t_max = torch.zeros(n_rays, device=dev), torch.full((n_rays,), 1e10, device=dev)
    if near_plane is not None:
        t_min = torch.clamp(t_min, min=near_plane)
    if far_plane is not None:
        t_max = torch.clamp(t_max, max=far_plane)
    dt = float(render_step_size)
    n_steps = torch.clamp(torch.ceil((t_max - t_min) / dt), min=0).long()
    n_steps = torch.where(t_min >= 1e10, torch.zeros_like(n_steps), n_steps)
    max_steps = int(n_steps.max()) if n_rays > 0 else 0
    k = torch.arange(max_steps, device=dev, dtype=torch.float32)
    t_s = t_min[:, None] + k[None, :] * dt                        # fixed-step lattice per ray
    t_e = t_s + dt
    keep = k[None, :] < n_steps[:, None]
    if grid is not None:
        mid = rays_o[:, None, :] + rays_d[:, None, :] * ((t_s + t_e) * 0.5)[..., None]
        idx, inside = _cell_index(mid.reshape(-1, 3), grid.roi_aabb, grid.resolution)
        occ = torch.zeros(idx.shape[0], dtype=torch.bool, device=dev)
        occ[inside] = grid.binary.flatten()[idx[inside]]
        keep &= occ.view(n_rays, max_steps)
    ray_indices = torch.arange(n_rays, device=dev)[:, None].expand(n_rays, max_steps)[keep]
    t_starts, t_ends = t_s[keep][:, None], t_e[keep][:, None]
    if alpha_fn is not None and ray_indices.numel() > 0:
        alphas = alpha_fn(t_starts, t_ends, ray_indices.long()).reshape(-1)
        # exclusive transmittance within each ray (packed, ray-sorted): cumulative sums of log(1-alpha) per segment
        logt = torch.log(torch.clamp(1 - alphas, min=1e-30))
        csum = torch.cumsum(logt, 0)
        first = torch.ones_like(ray_indices, dtype=torch.bool)
        first[1:] = ray_indices[1:] != ray_indices[:-1]
        start_off = torch.zeros_like(csum)
        seg_start = torch.nonzero(first)[:, 0]
        base = torch.where(seg_start > 0, csum[(seg_start - 1).clamp(min=0)], torch.zeros_like(csum[seg_start]))
        start_off = base[torch.cumsum(first.long(), 0) - 1]
        trans = torch.exp(csum - logt - start_off)
        vis = trans >= early_stop_eps
        if alpha_thre > 0:
            vis &= alphas >= alpha_thre
        ray_indices, t_starts, t_ends = ray_indices[vis], t_starts[vis], t_ends[vis]
    return ray_indices.to(torch.int32), t_starts, t_ends
