"""Occupancy-grid acceleration — the part of nerfacc 0.3.x the reference uses (nerf/run_nerf_acc.py:196-198,284-287;
nerf/nerf_helpers_acc.py:10-31,65-78; visualization/visualization.py:214), on the HIP kernels of
csrc/afx_kernels_grid.hip (include/afx.h: afx_grid_*, afx_march_*).

nerfacc is a third-party CUDA package that the reference neither vendors nor pins and that is absent here, so this is
its PUBLISHED algorithm (0.3.x line: `OccupancyGrid`, `ContractionType.AABB`, `ray_marching`, `render_visibility`) —
PARITY UNPINNED: no reference output exists to compare with; oracle/angio_oracle.py restates the same semantics on
the CPU for the tests.  Semantics kept:

  OccupancyGrid(roi_aabb, resolution=128): `occs` float [res^3], `binary` bool [res,res,res];
    every_n_step(step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16): every n-th training step
    evaluate occupancy at one uniformly jittered point per selected cell (all cells during warm-up, afterwards
    num_cells/4 uniformly drawn cells plus num_cells/4 drawn among the occupied ones),
    occs[i] = max(occs[i] * ema_decay, occ), binary = occs > min(mean(occs), occ_thre).
  ray_marching: t range = ray/AABB intersection clipped to [near, far]; fixed-step lattice t_min + k*dt;
    a step is kept when the cell containing its mid-point is occupied; with alpha_fn, render_visibility:
    steps with alpha < alpha_thre are skipped (they do not attenuate the transmittance), the ray ends once the
    transmittance over its kept steps falls below early_stop_eps.  Returns packed, ray-sorted
    (ray_indices, t_starts[n,1], t_ends[n,1]).

The grid lives on the GPU (there is no CPU fallback).  Randomness: which cells are refreshed is a torch draw (an index
list is data); the jitter inside a cell is drawn in the kernel from the counter-based Philox stream (seed, step) unless
the caller passes `jitter` (parity tests).

`ray_marching(..., return_packed=True)` additionally returns the kept samples group-aligned (engine.PackedGroups) for the fused
packed training step (`render.train_step_packed_mse`: the reference's positions -> get_predictions -> acc_render_volume_density -> mse ->
backward in one pass)."""
from __future__ import annotations

import weakref

import torch

from .. import engine as _engine
from .._lib import AfxError


class ContractionType:
    AABB = 0


class OccupancyGrid(torch.nn.Module):
    NUM_DIM = 3
    JITTER_STREAM_TAG = 0x47524944 << 32          # Philox stream ids of the in-cell jitter: tag | training step

    def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB, seed: int = 0):
        super().__init__()
        if contraction_type != ContractionType.AABB:
            raise NotImplementedError("only ContractionType.AABB is used by the reference")
        res = [resolution] * 3 if isinstance(resolution, int) else list(resolution)
        self.register_buffer("_roi_aabb", torch.as_tensor(roi_aabb, dtype=torch.float32).flatten().clone())
        self.register_buffer("resolution", torch.tensor(res, dtype=torch.int32))
        self.num_cells = int(res[0] * res[1] * res[2])
        self.register_buffer("occs", torch.zeros(self.num_cells))
        self.register_buffer("_binary_u8", torch.zeros(self.num_cells, dtype=torch.uint8))
        self.register_buffer("_bits", torch.zeros((self.num_cells + 31) // 32, dtype=torch.int32))      # packed bitfield for the march
        self.register_buffer("_scratch", torch.zeros(self.num_cells))
        self.register_buffer("_partial", torch.zeros(256, dtype=torch.float64))
        self._aabb_host = [float(x) for x in self._roi_aabb.tolist()]
        self._res_host = [int(x) for x in res]
        self.contraction_type = contraction_type
        self.seed = int(seed)

    @property
    def roi_aabb(self):
        return self._roi_aabb

    @property
    def binary(self):
        return self._binary_u8.view(*self._res_host).bool()

    @binary.setter
    def binary(self, mask):
        self.set_binary(mask)

    # the reference restores a trained grid with `acc_grid._binary = grid_occupancy` (visualization/visualization.py:162,
    # nerfacc 0.3.x keeps the mask in `_binary`): the same assignment works here and reaches the bitfield the march reads
    @property
    def _binary(self):
        return self.binary

    @_binary.setter
    def _binary(self, mask):
        self.set_binary(mask)

    @torch.no_grad()
    def set_binary(self, mask):
        """Install an occupancy mask [res,res,res] (or flat [num_cells]; bool or 0/1 numbers): the byte tensor `binary` views, and,
        once the grid lives on the GPU, the packed bitfield of the march (afx_grid_pack).  `occs` is left alone."""
        mask = torch.as_tensor(mask)
        if mask.numel() != self.num_cells:
            raise ValueError(f"binary: {mask.numel()} cells, expected {self.num_cells} = {self._res_host}")
        self._binary_u8.copy_((mask.reshape(-1) != 0).to(torch.uint8))
        self._bits_stale = not self._binary_u8.is_cuda      # a host-side grid is packed when it reaches the GPU
        if self._binary_u8.is_cuda:
            _engine.grid_pack(self._aabb_host, self._res_host, self._binary_u8, self._bits)

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if getattr(self, "_bits_stale", False) and self._binary_u8.is_cuda:
            _engine.grid_pack(self._aabb_host, self._res_host, self._binary_u8, self._bits)
            self._bits_stale = False
        return self

    @property
    def bits(self):
        if getattr(self, "_bits_stale", False):
            raise AfxError("OccupancyGrid: the mask was set on the host; move the grid to the GPU (.to(device)) before marching")
        return self._bits

    def _require_gpu(self):
        if not self.occs.is_cuda:
            raise AfxError("OccupancyGrid: move the grid to the GPU (.to(device)); there is no CPU fallback")

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n):
        uniform = torch.randint(self.num_cells, (n,), device=self.occs.device)
        occupied = torch.nonzero(self._binary_u8)[:, 0]
        if n < len(occupied):
            occupied = occupied[torch.randint(len(occupied), (n,), device=self.occs.device)]
        return torch.cat([uniform, occupied], dim=0)

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre=0.01, ema_decay=0.95, warmup_steps=256, jitter=None):
        self._require_gpu()
        if step < warmup_steps:
            indices, n = None, self.num_cells
        else:
            indices = self._sample_uniform_and_occupied_cells(self.num_cells // 4).to(torch.int32).contiguous()
            n = indices.numel()
        # (stream tag "GRID" in the high word: the ray sampler draws from (seed, step) with the same seed, nerf/run_nerf_acc.py)
        x = _engine.grid_points(self._aabb_host, self._res_host, indices, n, jitter=jitter, seed=self.seed,
                                stream_id=self.JITTER_STREAM_TAG | int(step), device=self.occs.device)
        occ = occ_eval_fn(x).reshape(-1).float().contiguous()
        _engine.grid_update(self._aabb_host, self._res_host, self.occs, indices, occ, ema_decay, self._scratch)
        _engine.grid_binarize(self._aabb_host, self._res_host, self.occs, occ_thre, self._binary_u8, self._bits, self._partial)

    @torch.no_grad()
    def every_n_step(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
        if not self.training:
            raise RuntimeError("every_n_step() is a training-time call; in eval mode use the grid as is")
        if step % n == 0:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)

    @torch.no_grad()
    def query_occ(self, samples):
        """Occupancy (0/1) at world points [P,3]; points outside the ROI are empty (visualization.py:214)."""
        lo, hi = self._roi_aabb[:3], self._roi_aabb[3:]
        u = (samples - lo) / (hi - lo)
        inside = ((u >= 0) & (u < 1)).all(-1)
        res = self.resolution.to(samples.device)
        ijk = torch.minimum((u * res).floor().long().clamp(min=0), (res - 1).long())
        idx = (ijk[:, 0] * res[1] + ijk[:, 1]) * res[2] + ijk[:, 2]
        out = torch.zeros(samples.shape[0], dtype=torch.bool, device=samples.device)
        out[inside] = self._binary_u8[idx[inside]].bool()
        return out


_AABB_HOST = []      # [(weakref to the tensor, version, six floats)]


def _aabb_on_host(scene_aabb):
    """The six floats of a scene box as host numbers.  The reference passes the same DEVICE tensor on every iteration (run_nerf_acc.py:196,288);
    reading it back each time is a host synchronisation per iteration, so the read is remembered for that tensor OBJECT (weak reference) at
    that version - a different tensor, or the same one after an in-place write, is read again."""
    if not torch.is_tensor(scene_aabb) or not scene_aabb.is_cuda:
        return [float(x) for x in torch.as_tensor(scene_aabb).flatten().tolist()]
    for ref, version, vals in _AABB_HOST:
        if ref() is scene_aabb and version == scene_aabb._version:
            return vals
    vals = [float(x) for x in scene_aabb.flatten().tolist()]
    _AABB_HOST[:] = [e for e in _AABB_HOST if e[0]() is not None and e[0]() is not scene_aabb][-7:] + [(weakref.ref(scene_aabb), scene_aabb._version, vals)]
    return vals


@torch.no_grad()
def ray_marching(rays_o, rays_d, scene_aabb=None, grid=None, alpha_fn=None, near_plane=None, far_plane=None,
                 early_stop_eps=1e-4, alpha_thre=0.0, render_step_size=1e-3, raw_fn=None, return_packed=False):
    """nerfacc.ray_marching on the HIP kernels.  `alpha_fn(t_starts[n,1], t_ends[n,1], ray_indices[n]) -> alpha[n,1]` is the
    upstream callback; `raw_fn(points[n,3]) -> raw[n,1]` is the short-cut the reference's alpha_fn reduces to (sigmoid
    density at the interval mid-point): the mid-points come out of the march kernel and alpha is formed in the visibility
    kernel, so nothing but the MLP launch sits between the two."""
    if not rays_o.is_cuda:
        raise AfxError("ray_marching: rays must live on a GPU; there is no CPU fallback")
    aabb = None if scene_aabb is None else _aabb_on_host(scene_aabb)
    bits = grid.bits if grid is not None else None
    ri, ts, te, pts, offsets = _engine.march(rays_o, rays_d, aabb, near_plane, far_plane, render_step_size, grid_bits=bits,
                                             grid_aabb=None if grid is None else grid._aabb_host,
                                             grid_res=None if grid is None else grid._res_host, want_points=raw_fn is not None)
    if (alpha_fn is None and raw_fn is None) or ri.numel() == 0:
        if return_packed:
            return ri, ts[:, None], te[:, None], _engine.pack_groups(offsets, ts, te)
        return ri, ts[:, None], te[:, None]
    if raw_fn is not None:
        vals, is_alpha = raw_fn(pts).reshape(-1).float(), False
    else:
        vals, is_alpha = alpha_fn(ts[:, None], te[:, None], ri.long()).reshape(-1).float(), True
    if return_packed:      # + the group-aligned copy render.train_step_packed_mse takes (its size comes with the same host sync)
        ri2, ts2, te2, off2, goff, ng = _engine.march_visibility(vals, ts, te, offsets, early_stop_eps, alpha_thre, is_alpha=is_alpha,
                                                                 return_offsets="groups")
        return ri2, ts2[:, None], te2[:, None], _engine.pack_groups(off2, ts2, te2, n_groups=ng, group_offsets=goff)
    ri2, ts2, te2 = _engine.march_visibility(vals, ts, te, offsets, early_stop_eps, alpha_thre, is_alpha=is_alpha)
    return ri2, ts2[:, None], te2[:, None]
