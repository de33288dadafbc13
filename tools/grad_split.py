#!/usr/bin/env python3
"""Per-tensor relative L2 of the full-size (512^2 x 128, 8x256) fused-step weight gradient against the exact-fp32 kernels, for the
16-bit and 8-bit stash precisions, on bench.py's phantom targets and on random targets."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import train_step_mse, projection_spec, render_projection
from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values, capsule_tree, capsule_mu, VoxelVolume
from nerf_for_angiography_amd.engine import project_volume
dev = torch.device("cuda:0")
W, S = 512, 128
focal, near, far = 13.0 * W, 1400.0, 1600.0
def model(prec):
    torch.manual_seed(0)
    md = dict(num_early_layers=8, num_late_layers=0, num_filters=256, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc=os.environ.get("ENC", "none"), pos_enc_basis=5, act_func="relu", fourier_sigma=5,
              num_img=1, device=dev, precision=prec)
    m = CPPN(md).to(dev)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-5.0)
    if md["pos_enc"] == "barf":
        m.update_barf_alpha(2.5, "pts")
    m.engine.max_workspace_bytes = 48 << 30
    return m
_, _, m44, _, _ = get_ray_values(0.0, 0.0, 0.0, np.array([0, 0, 1500.0]), 2, 2, focal, "cpu")
pose = torch.from_numpy(m44[None]).to(dev)
ax = np.linspace(-100.0, 100.0, 192)
tx = torch.from_numpy(ax).float().to(dev)
caps = capsule_tree(levels=5, seed=0)
with torch.no_grad():
    gx, gy, gz = torch.meshgrid(tx, tx, tx, indexing="ij")
    mu = torch.cat([capsule_mu(torch.stack([gx[i:i + 16], gy[i:i + 16], gz[i:i + 16]], -1).reshape(-1, 3), caps) for i in range(0, 192, 16)]).reshape(192, 192, 192)
    vol = VoxelVolume(ax, ax, ax, mu.cpu().numpy(), fill_value=0.0, device=dev)
    z_gt = torch.linspace(0., 1., 160, device=dev) * (far - near) + near
    phantom = project_volume(vol.values, vol.origin, vol.spacing, vol.fill_value, z_gt, poses=pose, width=W, height=W, focal=focal, type_ct=True).reshape(-1).contiguous()
targets = {"phantom": phantom, "random": torch.rand(W * W, device=dev)}
spec = projection_spec(pose, W, W, focal, S, near, far)
for tname, tgt in targets.items():
    m32 = model("f32")
    out = render_projection(m32, pose, W, W, focal, S, near, far)
    (((out.rgb_map - tgt) ** 2).sum() / (W * W)).backward()
    ref = {k: p.grad.double() for k, p in m32.named_parameters() if p.grad is not None}
    del m32, out
    for prec in ("f16", "f16s8"):
        m = model(prec)
        train_step_mse(m, spec, tgt)
        got = {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}
        tot = float(torch.sqrt(sum(((got[k] - ref[k]) ** 2).sum() for k in ref)) / torch.sqrt(sum((ref[k] ** 2).sum() for k in ref)))
        per = {k.replace("early_pts_layers.", "L").replace("output_linear.0", "out"): f"{float((got[k] - ref[k]).norm() / ref[k].norm()):.1e}" for k in ref}
        print(tname, prec, f"total {tot:.2e}", per, flush=True)
        del m
