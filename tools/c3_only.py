import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import render_rays
from nerf_for_angiography_amd.nerf.nerf_helpers import fine_sampling
from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
dev = torch.device("cuda:0")
torch.manual_seed(0)
md = dict(num_early_layers=8, num_late_layers=0, num_filters=256, num_input_channels=3, num_output_channels=1,
          num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5,
          num_img=1, device=dev, precision="f16s8")
if os.environ.get("PRE"):      # what tools/measure_configs.py does before C3: a 1024^2 x 256 train step with its own model, freed afterwards
    from nerf_for_angiography_amd.render import train_step_mse, projection_spec
    mp = CPPN(md).to(dev); mp.engine.max_workspace_bytes = 128 << 30
    Wp = int(os.environ["PRE"])
    _, _, m44p, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), Wp, Wp, 13.0 * Wp, dev)
    specp = projection_spec(torch.from_numpy(m44p[None]).to(dev), Wp, Wp, 13.0 * Wp, 256, 1400.0, 1600.0)
    train_step_mse(mp, specp, torch.rand(Wp * Wp, device=dev)); torch.cuda.synchronize()
    del mp, specp
m = CPPN(md).to(dev)
with torch.no_grad():
    # dense convention: the last interval is 1e10 long, so any sigma(far) > 1e-9 renders the pixel as exactly 0 and every gradient as exactly 0 - a bias of
    # -26 (sigma = 5e-12) keeps real gradients in the step (OBIAS=-5: the degenerate state this tool measured until the end of round 3)
    m.output_linear[0].bias.fill_(float(os.environ.get("OBIAS", "-26")))
m.engine.max_workspace_bytes = 128 << 30
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
if os.environ.get("WS_FIRST"):
    ws = m.engine._workspace(128 << 30, dev)      # allocate the backward workspace before anything else of size
    fill = os.environ.get("WS_FILL")
    if fill == "zero":
        ws.zero_()
    elif fill == "rand":      # random bytes, 8 GiB at a time
        for i in range(0, ws.numel(), 8 << 30):
            ws[i:i + (8 << 30)].random_(0, 256)
    del ws
W, SC, NF = 512, 128, 64
o, d, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, dev)
o, d = o.reshape(-1, 3).float().contiguous(), d.reshape(-1, 3).float().contiguous()
tgt = torch.rand(W * W, device=dev); z = torch.linspace(1400.0, 1600.0, SC, device=dev)
PROF = not os.environ.get("NOPROF")
m.engine.profile(PROF)
from nerf_for_angiography_amd.render import hierarchical_train_step_mse
def step3():
    opt.zero_grad(set_to_none=True)
    if os.environ.get("FUSED"):      # coarse forward -> in-kernel weights / sample_pdf / merge -> split-phase fused fine step
        hierarchical_train_step_mse(m, o, d, z, NF, tgt); opt.step()
        return
    with torch.no_grad():
        coarse = render_rays(m, o, d, mode="dense", z=z, want_aux=True)
    rgb, dep, ent = fine_sampling(z, coarse.weights, o, d, m, None, NF, 131072)
    torch.nn.functional.mse_loss(rgb, tgt).backward(); opt.step()
step3(); torch.cuda.synchronize()
ws_ = m.engine._ws
print(f"workspace: {ws_.numel() / 2**30:.2f} GiB at 0x{ws_.data_ptr():x} (offset in 2 MiB: {ws_.data_ptr() % (2 << 20)}, in 1 GiB: {ws_.data_ptr() % (1 << 30)}); "
      f"torch reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
gr = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])
print("after one step: grad |max|", float(gr.abs().max()), "NaNs", int(torch.isnan(gr).sum()), "weights |sum|", float(sum(p.double().abs().sum() for p in m.parameters())), flush=True)
for rep in range(int(os.environ.get("REPS", "3"))):
    step3(); torch.cuda.synchronize()
    if PROF:
        for k in ("chain_fwd", "chain_bwd", "wgrad"): m.engine.profile_read(k)
    t0 = time.perf_counter()
    for _ in range(3): step3()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    print(f"C3 {t*1e3:.1f} ms", {k: round(m.engine.profile_read(k)[0] / 3, 2) for k in ("chain_fwd", "chain_bwd", "wgrad")} if PROF else "", flush=True)
