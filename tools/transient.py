#!/usr/bin/env python3
"""Step time as a function of what the chip did just before (C3 measured 187-189 ms as the 4th configuration of
tools/measure_configs.py - 3 timed steps right after an idle gap of model set-up - and 213-219 ms in a loop of its own).
For C4 (fused 512^2 x 128 step) and C3 (hierarchical step): idle for `gap` seconds, then 14 steps, each timed on its own
(host clock around a synchronised step).  The first steps after an idle gap run at the boost clock the power manager allows
until its averaging window fills; the steady state is what BASELINE.md quotes."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import render_rays, train_step_mse, projection_spec
from nerf_for_angiography_amd.nerf.nerf_helpers import fine_sampling
from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
dev = torch.device("cuda:0")
torch.manual_seed(0)
md = dict(num_early_layers=8, num_late_layers=0, num_filters=256, num_input_channels=3, num_output_channels=1,
          num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5,
          num_img=1, device=dev, precision="f16s8")
m = CPPN(md).to(dev)
with torch.no_grad():
    m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-5.0)
m.engine.max_workspace_bytes = 128 << 30
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
W, SC, NF = 512, 128, 64
o, d, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, dev)
pose = torch.from_numpy(m44[None]).to(dev)
o, d = o.reshape(-1, 3).float().contiguous(), d.reshape(-1, 3).float().contiguous()
tgt = torch.rand(W * W, device=dev); z = torch.linspace(1400.0, 1600.0, SC, device=dev)
spec = projection_spec(pose, W, W, 13.0 * W, SC, 1400.0, 1600.0)
def c4():
    opt.zero_grad(set_to_none=True); train_step_mse(m, spec, tgt); opt.step()
def c3():
    opt.zero_grad(set_to_none=True)
    with torch.no_grad():
        coarse = render_rays(m, o, d, mode="dense", z=z, want_aux=True)
    rgb, dep, ent = fine_sampling(z, coarse.weights, o, d, m, None, NF, 131072)
    torch.nn.functional.mse_loss(rgb, tgt).backward(); opt.step()
out = {}
for name, fn in (("C4", c4), ("C3", c3)):
    fn(); fn(); torch.cuda.synchronize()
    for gap in (0.0, 4.0):
        for _ in range(12): fn()           # loaded state
        torch.cuda.synchronize()
        time.sleep(gap)
        ts = []
        for _ in range(14):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
        out[f"{name} after {gap:.0f} s idle"] = ts
        print(name, f"after {gap:.0f} s idle:", ts, flush=True)
print(json.dumps(out))
