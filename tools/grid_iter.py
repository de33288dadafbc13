#!/usr/bin/env python3
"""The reference's grid-march training iteration (nerf/run_nerf_acc.py:284-307) in isolation: 5 625 rays x 300 steps, 4x128 (or `layers width`),
occupancy grid shaped like a trained vessel tree (a few % of the cells), fused packed step vs the operator sequence.  Prints ms / iteration;
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  usage: grid_iter.py [layers width [iters [ops|fused|one [fill]]]]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import train_step_packed_mse, march_train_step_mse
from nerf_for_angiography_amd.engine import sample_rays, RayBatchSampler
from nerf_for_angiography_amd.nerf.nerf_helpers import get_predictions
from nerf_for_angiography_amd.nerf.nerf_helpers_acc import acc_ray_marching, acc_render_volume_density
from nerf_for_angiography_amd.nerf.occupancy import OccupancyGrid
from nerf_for_angiography_amd.phantomdata.helpers import capsule_tree, capsule_mu

dev = torch.device("cuda:0")
layers, width = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 128)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
mode = sys.argv[4] if len(sys.argv) > 4 else "fused"
fill = sys.argv[5] if len(sys.argv) > 5 else "vessels"
torch.manual_seed(0)
NT = 90 * 100 * 100
tab_o = torch.randn(NT, 3, device=dev) * 3 + torch.tensor([0, 0, 1500.0], device=dev)
tab_d = torch.nn.functional.normalize(torch.randn(NT, 3, device=dev) * 0.03 + torch.tensor([0, 0, -1.0], device=dev), dim=-1)
tab_p, tab_w = torch.rand(NT, device=dev), torch.rand(NT, device=dev) + 0.05
md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3, num_output_channels=1, num_input_channels_views=0,
          use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5, num_img=1, device=dev, precision="f16s8")
m = CPPN(md).to(dev)
with torch.no_grad():
    m.output_linear[0].bias.fill_(-3.0)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=os.environ.get("ADAM_FUSED", "1") != "0")      # (one multi-tensor kernel instead of the foreach sequence)
aabb = torch.tensor([-100.0, -100, -100, 100, 100, 100], device=dev)
grid = OccupancyGrid(roi_aabb=aabb, resolution=128).to(dev)
res = 128
c = (torch.stack(torch.meshgrid(*[torch.arange(res, device=dev)] * 3, indexing="ij"), -1).float() + 0.5) / res * 200 - 100
if fill == "vessels":      # cells within 4 units of the capsule tree: what a trained grid looks like
    caps = capsule_tree(levels=5, seed=0)
    caps[:, 6] += 4.0
    mask = torch.cat([capsule_mu(c[i:i + 8].reshape(-1, 3), caps) > 0 for i in range(0, res, 8)]).reshape(res, res, res)
else:
    mask = torch.ones(res, res, res, dtype=torch.bool, device=dev)
grid._binary = mask
print(f"occupied cells: {float(mask.float().mean()) * 100:.1f} %")
R, S, near, far = 5625, 300, 1400.0, 1600.0
n, tot = [0], [0]
batches = RayBatchSampler(tab_o, tab_d, tab_p, tab_w, R, seed=0, prefetch=int(os.environ.get("PREFETCH", 16))) if os.environ.get("PREFETCH", "16") != "0" else None
def it():
    n[0] += 1
    o, d, tgt, _ = batches.draw(n[0]) if batches else sample_rays(tab_o, tab_d, tab_p, tab_w, R, seed=0, stream_id=n[0])
    opt.zero_grad(set_to_none=True)
    if mode == "one":      # the whole iteration body in one library call
        _, _, kept = march_train_step_mse(m, grid, aabb, o, d, S, near, far, 1e-2, 1e-4, tgt)
        tot[0] += kept
        if kept:
            opt.step()
        return
    with torch.no_grad():
        out = acc_ray_marching(m, grid, aabb, o, d, S, near, far, 1e-2, 1e-4, return_packed=(mode == "fused"))
    ri, ts, te = out[:3]
    tot[0] += ri.numel()
    if ri.numel() == 0:
        return
    if mode == "fused":
        train_step_packed_mse(m, o, d, out[3], tgt)
    else:
        pos = o[ri.long()] + d[ri.long()] * (ts + te) / 2.0
        pred, _ = acc_render_volume_density(get_predictions(m, pos, 131072), ri, ts, te, R, S)
        torch.nn.functional.mse_loss(pred, tgt).backward()
    opt.step()
for _ in range(10): it()
tot[0] = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): it()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
print(f"{layers}x{width} {mode} ({fill}): {t * 1e3:.3f} ms/iteration, {1 / t:.0f} it/s, {tot[0] / iters:.0f} kept samples/iteration")
