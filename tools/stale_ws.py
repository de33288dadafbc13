#!/usr/bin/env python3
"""Do results depend on what the backward workspace held before the call?  (They must not: every byte a kernel reads has to be
written by the same call.)  Same step with the workspace pre-filled with zeros, 0xFF bytes and random bytes; several ray chunks."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import render_rays, train_step_mse
from nerf_for_angiography_amd.engine import RenderSpec
dev = torch.device("cuda:0")
L, F = int(os.environ.get("L", 8)), int(os.environ.get("F", 256))
WS = int(os.environ.get("WS_MB", 512)) << 20
def run(fill, mode, prec, enc, R, S):
    torch.manual_seed(0)
    md = dict(num_early_layers=L, num_late_layers=0, num_filters=F, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc=enc, pos_enc_basis=5, act_func="relu", fourier_sigma=5,
              num_img=1, device=dev, precision=prec)
    m = CPPN(md).to(dev)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-4.0)
    m.engine.max_workspace_bytes = WS
    ws = m.engine._workspace(WS, dev)
    if fill == "zero": ws.zero_()
    elif fill == "ff": ws.fill_(255)
    else: ws.random_(0, 256)
    g = torch.Generator().manual_seed(1)
    o = (torch.tensor([[0.0, 0.0, 1.5]]).repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.01).to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.2 + torch.tensor([0, 0, -1.0]), dim=-1).to(dev)
    tgt = torch.rand(R, generator=g).to(dev)
    if mode == "dense":
        z = (torch.sort(torch.rand(R, S, generator=g), dim=-1).values * 2.0 + 0.5).to(dev)
        out = render_rays(m, o, d, mode="dense", z=z)
        torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
        pix = out.rgb_map.detach()
    elif mode == "acc":
        out = render_rays(m, o, d, S, 0.5, 2.5, mode="acc")
        torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
        pix = out.rgb_map.detach()
    else:
        _, pix = train_step_mse(m, RenderSpec(n_rays=R, n_samples=S, origins=o, dirs=d, mode="acc", t_near=0.5, t_far=2.5), tgt)
    return pix.cpu(), torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).cpu()
bad = 0
for prec in os.environ.get("PRECS", "f16s8,f16,bf16,f32").split(","):
    for mode, S in (("dense", 192), ("acc", 192), ("fused", 128)):
        if mode == "fused" and prec == "f32": continue
        for enc in ("none", "barf"):
            R = int(os.environ.get("RAYS", 6000))
            ref = run("zero", mode, prec, enc, R, S)
            for fill in ("ff", "rand"):
                got = run(fill, mode, prec, enc, R, S)
                same = torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
                bad += not same
                print(prec, mode, enc, fill, "identical" if same else f"DIFFERS: NaNs {int(torch.isnan(got[1]).sum())}, max |dg| {float((got[1] - ref[1]).abs().nan_to_num(1e30).max()):.3g}, ref |g|max {float(ref[1].abs().max()):.3g}", flush=True)
print("FAILED" if bad else "OK", bad)
