#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations on one MI355X (fills BASELINE.md section 3).
C2: 256^2 x 64, 8x256; C3: 512^2 x 128 coarse + 64 fine (hierarchical, dense convention), 8x256;
C4 (1 GPU): 512^2 x 128; C5 (1 GPU): 1024^2 x 256.  fwd+bwd+Adam, f16s8 (the training precision), one projection per step."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import train_step_mse, projection_spec, render_projection, render_rays
from nerf_for_angiography_amd.nerf.nerf_helpers import fine_sampling
from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values

dev = torch.device("cuda:0")
def model(prec="f16s8", enc="none"):
    torch.manual_seed(0)
    md = dict(num_early_layers=8, num_late_layers=0, num_filters=256, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc=enc, pos_enc_basis=5, act_func="relu", fourier_sigma=5,
              num_img=1, device=dev, precision=prec)
    m = CPPN(md).to(dev)
    if enc == "barf":
        m.update_barf_alpha(2.5, "pts")
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-5.0)
    m.engine.max_workspace_bytes = 128 << 30      # as bench.py: 288 GB of HBM per GPU, few large ray chunks
    return m

def timeit(fn, warm=1, steps=3, min_time=1.0):
    """Mean time of `fn` over at least `steps` calls and at least `min_time` seconds (a 12 ms step timed over 3 calls right after set-up reads 8 % high)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    steps = max(steps, int(min_time / max(t1, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

out = {}
for name, (W, S) in {"C2 256^2x64": (256, 64), "C4 512^2x128 (1 GPU)": (512, 128), "C5 1024^2x256 (1 GPU)": (1024, 256)}.items():
    m = model(); opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
    o, d, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, dev)
    pose = torch.from_numpy(m44[None]).to(dev); tgt = torch.rand(W * W, device=dev)
    spec = projection_spec(pose, W, W, 13.0 * W, S, 1400.0, 1600.0)
    def step():
        opt.zero_grad(set_to_none=True); train_step_mse(m, spec, tgt); opt.step()
    t = timeit(step)
    out[name] = dict(ms_per_step=round(t * 1e3, 2), ray_samples_per_s=round(W * W * S / t / 1e6, 1))
    print(name, out[name], flush=True)
# C4 with positional encodings (33 encoded inputs, L = 5): BARF (fixed weights) and fourier with TRAINABLE coefficients
for enc in ("barf", "fourier"):
    W, S = 512, 128
    m = opt = None; import gc; gc.collect(); torch.cuda.empty_cache()       # the previous model's 128 GiB workspace
    m = model(enc=enc); opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
    o, d, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, dev)
    pose = torch.from_numpy(m44[None]).to(dev); tgt = torch.rand(W * W, device=dev)
    spec = projection_spec(pose, W, W, 13.0 * W, S, 1400.0, 1600.0)
    def step_e():
        opt.zero_grad(set_to_none=True); train_step_mse(m, spec, tgt); opt.step()
    t = timeit(step_e)
    out[f"C4 512^2x128 {enc} L=5"] = dict(ms_per_step=round(t * 1e3, 2), ray_samples_per_s=round(W * W * S / t / 1e6, 1))
    print(enc, out[f"C4 512^2x128 {enc} L=5"], flush=True)
# C3: hierarchical coarse (128) + fine (128 + 64), dense convention with per-ray depths, autograd backward
W, SC, NF = 512, 128, 64
m = opt = None; gc.collect(); torch.cuda.empty_cache()
m = model(); opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
with torch.no_grad():      # dense convention (last interval 1e10): only sigma(far) < 1e-9 leaves a non-zero pixel and non-zero gradients in the step
    m.output_linear[0].weight.div_(4.0); m.output_linear[0].bias.fill_(-26.0)
m.invalidate()
o, d, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, dev)
o, d = o.reshape(-1, 3).float().contiguous(), d.reshape(-1, 3).float().contiguous()
tgt = torch.rand(W * W, device=dev); z = torch.linspace(1400.0, 1600.0, SC, device=dev)
from nerf_for_angiography_amd.render import hierarchical_train_step_mse
def step3_fused():      # the hierarchical step with the coarse depths evaluated once (afx_hier_train_step_mse)
    opt.zero_grad(set_to_none=True); hierarchical_train_step_mse(m, o, d, z, NF, tgt); opt.step()
tf = timeit(step3_fused)
out["C3 512^2x(128 coarse + 64 new), coarse re-use"] = dict(ms_per_step=round(tf * 1e3, 2), ray_samples_per_s_reference_count=round(W * W * (SC + SC + NF) / tf / 1e6, 1))
print("C3 fused", out["C3 512^2x(128 coarse + 64 new), coarse re-use"], flush=True)
def step3():            # operator by operator: render with aux -> fine_sampling -> mse -> backward (recomputes the fine forward)
    opt.zero_grad(set_to_none=True)
    with torch.no_grad():
        coarse = render_rays(m, o, d, mode="dense", z=z, want_aux=True)
    rgb, dep, ent = fine_sampling(z, coarse.weights, o, d, m, None, NF, 131072)
    torch.nn.functional.mse_loss(rgb, tgt).backward(); opt.step()
t = timeit(step3)
m.engine.profile(True)
t_again = timeit(step3, warm=0, steps=3, min_time=0.0)
kern = {k: round(m.engine.profile_read(k)[0] / 3, 2) for k in ("chain_fwd", "chain_bwd", "wgrad")}
m.engine.profile(False)
print("C3 again (profiled):", round(t_again * 1e3, 2), kern, "workspace at 0x%x, %.1f GiB" % (m.engine._ws.data_ptr(), m.engine._ws.numel() / 2**30), flush=True)
out["C3 512^2x(128 coarse + 192 fine)"] = dict(ms_per_step=round(t * 1e3, 2), ray_samples_per_s=round(W * W * (SC + SC + NF) / t / 1e6, 1))
print("C3", out["C3 512^2x(128 coarse + 192 fine)"], flush=True)
# forward-only renders (evaluation): split-bf16 and bf16
for prec in ("bf16x3", "f16"):
    m = model(prec)
    pose = torch.from_numpy(get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), 512, 512, 13.0 * 512, dev)[2][None]).to(dev)
    with torch.no_grad():
        t = timeit(lambda: render_projection(m, pose, 512, 512, 13.0 * 512, 128, 1400.0, 1600.0))
    out[f"forward 512^2x128 {prec}"] = dict(ms=round(t * 1e3, 2), ray_samples_per_s=round(512 * 512 * 128 / t / 1e6, 1))
    print(prec, out[f"forward 512^2x128 {prec}"], flush=True)
# the reference's own training iteration (run_nerf_acc.py:142-155,263-307): 75^2 = 5 625 rays x 300 samples, 4x128 MLP (and 8x256),
# END TO END: the batch is drawn on the device from a resident table of 90 projections x 100 x 100 rays (weighted sampling
# without replacement, engine.sample_rays) - no pandas, no host round trip - then the train step + Adam
from nerf_for_angiography_amd.engine import RenderSpec, sample_rays
NT = 90 * 100 * 100
tab_o = torch.randn(NT, 3, device=dev) * 3 + torch.tensor([0, 0, 1500.0], device=dev)
tab_d = torch.nn.functional.normalize(torch.randn(NT, 3, device=dev) * 0.03 + torch.tensor([0, 0, -1.0], device=dev), dim=-1)
tab_p, tab_w = torch.rand(NT, device=dev), torch.rand(NT, device=dev) + 0.05
for layers, width in ((4, 128), (8, 256)):
    torch.manual_seed(0)
    md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5,
              num_img=1, device=dev, precision="f16s8")
    m = CPPN(md).to(dev); opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
    R, S = 5625, 300
    step_no = [0]
    def it():
        step_no[0] += 1
        o, d, tgt, _ = sample_rays(tab_o, tab_d, tab_p, tab_w, R, seed=0, stream_id=step_no[0])
        spec = RenderSpec(n_rays=R, n_samples=S, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)
        opt.zero_grad(set_to_none=True); train_step_mse(m, spec, tgt); opt.step()
    t = timeit(it, warm=5, steps=50, min_time=0.0)
    def sample_only():
        step_no[0] += 1
        sample_rays(tab_o, tab_d, tab_p, tab_w, R, seed=0, stream_id=step_no[0])
    ts = timeit(sample_only, warm=2, steps=50, min_time=0.0)
    out[f"reference training iteration 5625x300 {layers}x{width}"] = dict(ms_per_iter=round(t * 1e3, 3), iters_per_s=round(1 / t, 1),
                                                                          ray_samples_per_s=round(R * S / t / 1e6, 1),
                                                                          device_sampler_ms=round(ts * 1e3, 3))
    print(f"train-iter {layers}x{width}", out[f"reference training iteration 5625x300 {layers}x{width}"], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "configs.json"), "w"), indent=1)
