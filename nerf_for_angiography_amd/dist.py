"""Ray-batch data parallelism: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm, over xGMI).

The reference is single-device (nerf/run_nerf_acc.py:23).  Rays are independent, the only coupling is the
shared MLP, so the path shards by rays with ONE collective per step: a SUM all-reduce of the flat fp32
gradient buffer (527 621 floats = 2.1 MB at 8x256), issued on the flat buffer before it is split into
per-parameter views.  At that size the collective is latency-bound; one call is the whole cost."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import render as _render


def init_from_env(backend: str | None = None):
    """Initialise the default process group from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* and pick the device."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    # rehearsal knobs (several ranks on one GPU over gloo): AFX_DEVICE_INDEX pins the device, AFX_DIST_BACKEND the backend
    if os.environ.get("AFX_DEVICE_INDEX") is not None:
        local = int(os.environ["AFX_DEVICE_INDEX"])
    backend = backend or os.environ.get("AFX_DIST_BACKEND")
    if use_cuda:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    device = torch.device(f"cuda:{local}" if use_cuda else "cpu")
    return rank, world, device


def shard(n_items: int, rank: int, world: int):
    """Contiguous shard [start, start+count) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def broadcast_parameters(model, src: int = 0):
    """Every rank starts from rank `src`'s weights (one broadcast of the flat buffer)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(model.flat_params, src)


class GradSync:
    """SUM the flat parameter gradient over ranks with one all-reduce.

    Contract (one, everywhere): the loss of a step is the mean over the GLOBAL ray batch, so every rank seeds its
    backward with dL/dpixel = 2 (pixel - target) / n_global and the per-rank gradients ADD UP to the single-GPU gradient
    of the union batch.  That is exact for unequal shards too (a mean of per-rank means is not).
      * `render.train_step_mse(model, spec, target)` with this hook installed and no `n_global` derives it itself: one
        all-reduce of the rank's ray count (`global_count`), so the default call is correct on any number of ranks;
      * the autograd path (`render_*` -> your loss -> `.backward()`) must divide by the global count as well:
        `dist.global_mse(pred, target)` does; a loss that is a LOCAL mean (`F.mse_loss(pred, target)`) needs
        `GradSync(local_mean=True)`, which averages instead of summing (exact for equal shards only - the reason the
        global-count form is the default)."""

    def __init__(self, group=None, local_mean: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.local_mean = bool(local_mean)
        self.on_call = None            # optional (before, after) hooks, e.g. event records for bench.py

    def __call__(self, flat_grad: torch.Tensor):
        if self.world > 1:
            if self.on_call is not None:
                self.on_call[0]()
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            if self.local_mean:
                flat_grad.div_(self.world)
            if self.on_call is not None:
                self.on_call[1]()

    def global_count(self, n_local: int, device) -> int:
        """Rays of this step over all ranks (one tiny all-reduce; the ranks' batches may differ in size)."""
        if self.world == 1:
            return int(n_local)
        t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def install(self):
        _render._grad_hook = self
        return self

    @staticmethod
    def uninstall():
        _render._grad_hook = None


def global_mse(pred: torch.Tensor, target: torch.Tensor, group=None) -> torch.Tensor:
    """sum_r (pred_r - target_r)^2 / n_global: the loss whose per-rank gradients a SUM all-reduce (GradSync) turns into the
    gradient of the mean over the union batch (nerf/run_nerf_acc.py:298 on one device).  The returned value is this rank's
    SHARE of the global loss (add the ranks' values for logging)."""
    n = pred.numel()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.tensor([n], dtype=torch.int64, device=pred.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        n = int(t.item())
    return ((pred - target) ** 2).sum() / n


def density_grid_sharded(model, outside: float, n: int, group=None) -> torch.Tensor:
    """The reconstructed 3-D density grid (visualization/visualization.py:100-102,209-229; layout as render.density_grid:
    grid[i,j,k] = sigma(t[j], t[i], t[k]), t = linspace(-outside, outside, n+1)) computed by all ranks: each evaluates the
    MLP on its contiguous range of the (n+1)^3 points, one all-gather assembles the full grid on every rank.
    Points are generated from their flat index, so no rank materialises the whole point list."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    m = n + 1
    total = m * m * m
    start, count = shard(total, rank, world)
    dev = model.flat_params.device if getattr(model, "flat_params", None) is not None else next(model.parameters()).device
    t = torch.linspace(-outside, outside, m, dtype=torch.float64, device=dev).float()
    idx = torch.arange(start, start + count, device=dev)
    i, j, k = idx // (m * m), (idx // m) % m, idx % m
    pts = torch.stack([t[j], t[i], t[k]], -1).contiguous()
    # (evaluated at render.grid_precision: the training precisions miss the grid's 1e-4 bar, see render.density_grid)
    fused_gpu = getattr(model, "fused", False) and dev.type == "cuda"
    keep = getattr(model, "precision", None)
    if fused_gpu:
        model.precision = _render.grid_precision(model)
    try:
        with torch.no_grad():
            sig = torch.sigmoid(model(pts)).reshape(-1) if count > 0 else torch.zeros(0, device=dev)
    finally:
        if fused_gpu:
            model.precision = keep
    if world == 1:
        return sig.reshape(m, m, m)
    per = (total + world - 1) // world + 1                      # equal-size slots for the collective
    slot = torch.zeros(per, device=dev)
    slot[:count] = sig
    gathered = torch.empty(world * per, device=dev)
    dist.all_gather_into_tensor(gathered, slot, group=group)
    parts = [gathered[r * per:r * per + shard(total, r, world)[1]] for r in range(world)]
    return torch.cat(parts).reshape(m, m, m)
