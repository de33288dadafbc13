#!/usr/bin/env python3
"""The reference's own training iteration (nerf/run_nerf_acc.py:142-155,263-307): 75^2 = 5 625 rays x 300 samples/ray, 4x128 MLP,
batch drawn on the device from a resident 900 000-ray table, fused train step, Adam.  Prints ms/iteration; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel breakdown.  usage: ref_iter.py [layers width [iters]]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import train_step_mse
from nerf_for_angiography_amd.engine import RenderSpec, sample_rays, RayBatchSampler

dev = torch.device("cuda:0")
layers, width = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 128)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
NT = 90 * 100 * 100
torch.manual_seed(0)
tab_o = torch.randn(NT, 3, device=dev) * 3 + torch.tensor([0, 0, 1500.0], device=dev)
tab_d = torch.nn.functional.normalize(torch.randn(NT, 3, device=dev) * 0.03 + torch.tensor([0, 0, -1.0], device=dev), dim=-1)
tab_p, tab_w = torch.rand(NT, device=dev), torch.rand(NT, device=dev) + 0.05
md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3, num_output_channels=1,
          num_input_channels_views=0, use_bias=True, pos_enc=os.environ.get("ENC", "none"), pos_enc_basis=5, act_func="relu", fourier_sigma=5,
          num_img=1, device=dev, precision=os.environ.get("PREC", "f16s8"))
m = CPPN(md).to(dev)
if os.environ.get("AFX_VARIANT"):
    from nerf_for_angiography_amd.engine import Engine
    m._engine = Engine(width, layers, md["pos_enc"], 5 if md["pos_enc"] != "none" else 0, variant=os.environ["AFX_VARIANT"])
if md["pos_enc"] == "barf":
    m.update_barf_alpha(2.5, "pts")
# the driver's --out_bias_init (default -5).  NOT cosmetic: with the default bias 0 the 300 samples of a ray absorb everything (pixel = e^-100),
# every gradient is exactly zero, and the backward half then runs TWICE as fast as on real gradients (round 3 measured this tool in that
# state until the driver's own rate, 646 it/s against the tool's 890, gave it away) - OBIAS=0 reproduces the degenerate state
with torch.no_grad():
    m.output_linear[0].bias.fill_(float(os.environ.get("OBIAS", "-5")))
if os.environ.get("TARGET") == "const":
    tab_p = torch.full_like(tab_p, 0.9)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=os.environ.get("ADAM_FUSED", "1") != "0")      # (one multi-tensor kernel instead of the foreach sequence)
R, S = 5625, 300
n = [0]
batches = RayBatchSampler(tab_o, tab_d, tab_p, tab_w, R, seed=0, prefetch=int(os.environ.get("PREFETCH", 16))) if os.environ.get("PREFETCH", "16") != "0" else None
def it():
    n[0] += 1
    o, d, tgt, _ = batches.draw(n[0]) if batches else sample_rays(tab_o, tab_d, tab_p, tab_w, R, seed=0, stream_id=n[0])
    spec = RenderSpec(n_rays=R, n_samples=S, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)
    opt.zero_grad(set_to_none=True); train_step_mse(m, spec, tgt); opt.step()
for _ in range(10): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): it()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
print(f"{layers}x{width}: {t * 1e3:.3f} ms/iteration, {1 / t:.0f} it/s, {R * S / t / 1e6:.0f} M ray-samples/s")
