#!/usr/bin/env python3
"""Is k_chain<bwd> power-limited?  The same 512^2 x 128 fused train step (f16s8) on (a) the seeded weights, (b) all-zero
weights and biases (every MFMA operand zero: same instruction stream, minimal switching activity), (c) weights scaled so that
most ReLUs are dead.  Equal times = instruction-bound; a much shorter (b) = the clock under real data is what bounds (a)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import train_step_mse, projection_spec
from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values

dev = torch.device("cuda:0")
VARIANT = os.environ.get("AFX_VARIANT", "")     # "gaps": libafx_gaps.so (build.py --variant=gaps)
def model():
    torch.manual_seed(0)
    md = dict(num_early_layers=8, num_late_layers=0, num_filters=256, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5,
              num_img=1, device=dev, precision="f16s8")
    m = CPPN(md).to(dev)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-5.0)
    if VARIANT:
        from nerf_for_angiography_amd.engine import Engine
        m._engine = Engine(256, 8, "none", 0, variant=VARIANT)
    m.engine.max_workspace_bytes = 200 << 30
    return m

W, S = 512, 128
o, d, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, dev)
pose = torch.from_numpy(m44[None]).to(dev); tgt = torch.rand(W * W, device=dev)
spec = projection_spec(pose, W, W, 13.0 * W, S, 1400.0, 1600.0)
out = {}
for name in ("seeded", "zero", "seeded_again"):
    m = model()
    if name == "zero":
        with torch.no_grad():
            for p in m.parameters(): p.zero_()
        m.invalidate()
    m.engine.profile(True)
    def step():
        m.zero_grad(set_to_none=True); train_step_mse(m, spec, tgt)
    for _ in range(2): step()
    torch.cuda.synchronize(); m.engine.profile_read("chain_bwd"); m.engine.profile_read("wgrad")
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    cb, nb = m.engine.profile_read("chain_bwd"); wg, nw = m.engine.profile_read("wgrad")
    out[name] = dict(ms_per_step=round(t * 1e3, 2), chain_bwd_ms=round(cb / 5, 2), wgrad_ms=round(wg / 5, 2))
    print(name, out[name], flush=True)
print(json.dumps(out))
