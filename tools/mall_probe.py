#!/usr/bin/env python3
"""Does the stash round trip get cheaper when a chunk's stash fits the 256 MiB Infinity Cache?  The 512^2 x 128 fused train step (f16s8) with the
workspace capped so that a ray chunk holds fewer and fewer tiles: kernel time of k_chain<bwd> and k_wgrad_s8 per step (HIP events around the
launches: the launch gaps between the many small chunks are NOT in these numbers) and wall time per step."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import train_step_mse, projection_spec
from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values

dev = torch.device("cuda:0")
torch.manual_seed(0)
md = dict(num_early_layers=8, num_late_layers=0, num_filters=256, num_input_channels=3, num_output_channels=1, num_input_channels_views=0,
          use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5, num_img=1, device=dev, precision="f16s8")
m = CPPN(md).to(dev)
with torch.no_grad():
    m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-5.0)
W, S = 512, 128
_, _, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, dev)
pose = torch.from_numpy(m44[None]).to(dev); tgt = torch.rand(W * W, device=dev)
spec = projection_spec(pose, W, W, 13.0 * W, S, 1400.0, 1600.0)
m.engine.profile(True)
for gib in [float(x) for x in (sys.argv[1:] or ["200", "8", "2", "1", "0.6", "0.45", "0.36"])]:
    m.engine.max_workspace_bytes = int(gib * (1 << 30))
    m.engine._ws = None
    def step():
        m.zero_grad(set_to_none=True); train_step_mse(m, spec, tgt)
    step(); torch.cuda.synchronize(); m.engine.profile_read("chain_bwd"); m.engine.profile_read("wgrad")
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    cb, nb = m.engine.profile_read("chain_bwd"); wg, nw = m.engine.profile_read("wgrad")
    print(f"workspace {gib:6.2f} GiB: {nb // 3:5d} chunks/step, chain {cb / 3:7.2f} ms, wgrad {wg / 3:7.2f} ms, wall {t * 1e3:7.2f} ms/step", flush=True)
