#!/usr/bin/env python3
"""Dev probe (GPU): errors of each precision mode against the golden fixtures / CPU oracle, to set test tolerances."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_l2
from oracle import angio_oracle as orc
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import render_rays, train_step_mse
from nerf_for_angiography_amd.engine import RenderSpec
from nerf_for_angiography_amd.phantomdata.proj_helpers import source_matrix
DEV = "cuda:0"
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
def gold(n): return dict(np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")))
def make_model(layers, width, pos_enc="none", precision="f32", basis=5):
    md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc=pos_enc, pos_enc_basis=basis, act_func="relu", fourier_sigma=5,
              num_img=1, device=torch.device(DEV), precision=precision)
    return CPPN(md).to(DEV)
def load_sd(m, g, prefix="sd__"):
    m.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}, strict=False); return m
def grads(m): return {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}
precs = sys.argv[1:] or ["f16", "bf16", "bf16x3"]
for prec in precs:
    for name, L, W in [("none_relu_4x64", 4, 64), ("none_relu_4x128", 4, 128), ("none_relu_8x256", 8, 256)]:
        g = gold("g4_cppn_" + name); m = load_sd(make_model(L, W, precision=prec), g)
        with torch.no_grad(): y = m(T(g["x"]))
        print(prec, "mlp", name, f"{rel_l2(y.cpu().numpy(), g['y']):.2e}")
    g = gold("g4_cppn_barf_relu_4x64"); m = load_sd(make_model(4, 64, "barf", precision=prec), g)
    for a in (0.0, 2.5, 5.0):
        m.update_barf_alpha(a, "pts")
        with torch.no_grad(): y = m(T(g["x"]))
        print(prec, "mlp barf", a, f"{rel_l2(y.cpu().numpy(), g[f'y_alpha{a}']):.2e}")
    # C1 golden acc
    g = gold("g8_e2e_c1"); m = load_sd(make_model(4, 64, precision=prec), g, "init__")
    near, far, s = g["near_far_s"]; near, far, s = float(near), float(far), int(s)
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    out = render_rays(m, o, d, s, near, far, mode="acc"); torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
    ga = grads(m)
    print(prec, "c1 acc pix", f"{rel_l2(out.rgb_map.detach().cpu().numpy(), g['acc_rgb']):.2e}", "grad worst",
          f"{max(rel_l2(ga[k], g['acc_grad__' + k]) for k in ga):.2e}")
    m.zero_grad()
    spec = RenderSpec(n_rays=o.shape[0], n_samples=s, origins=o, dirs=d, mode="acc", t_near=near, t_far=far)
    loss_f, pix = train_step_mse(m, spec, tgt); gf = grads(m)
    print(prec, "c1 fused pix", f"{rel_l2(pix.cpu().numpy(), g['acc_rgb']):.2e}", "grad worst", f"{max(rel_l2(gf[k], g['acc_grad__' + k]) for k in gf):.2e}",
          "fused vs autograd", f"{max(rel_l2(gf[k], ga[k]) for k in gf):.2e}")
    # C2-scale vs oracle
    torch.manual_seed(3)
    m = make_model(8, 256, precision=prec)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-5.0)
    w = 256
    pose = source_matrix(np.array([0, 0, 1500.0]), 30.0, 12.0)
    o_all, d_all = orc.get_rays(pose, w, w, 13.0 * w)
    pick = torch.randperm(w * w)[:1003]
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    tgt = torch.rand(o.shape[0])
    cfg = dict(num_early_layers=8, num_filters=256)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for s in (64, 50):
        pix_c, loss_c, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=1400.0, far=1600.0, n_samples=s, convention="acc")
        m.zero_grad()
        out = render_rays(m, o.to(DEV), d.to(DEV), s, 1400.0, 1600.0, mode="acc")
        torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
        got = grads(m)
        errs = {k: rel_l2(got[k], v.numpy()) for k, v in grads_c.items()}
        print(prec, "c2", s, "pix", f"{rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()):.2e}", "grad worst", f"{max(errs.values()):.2e}",
              "per layer W:", " ".join(f"{errs[k]:.1e}" for k in sorted(errs) if k.endswith('weight')))
        if s == 64:
            m.zero_grad()
            spec = RenderSpec(n_rays=o.shape[0], n_samples=s, origins=o.to(DEV), dirs=d.to(DEV), mode="acc", t_near=1400.0, t_far=1600.0)
            _, pix = train_step_mse(m, spec, tgt.to(DEV)); gf = grads(m)
            print(prec, "c2 fused", "pix", f"{rel_l2(pix.cpu().numpy(), pix_c.numpy()):.2e}", "grad worst",
                  f"{max(rel_l2(gf[k], v.numpy()) for k, v in grads_c.items()):.2e}")
    # BARF-encoded inputs, forward + backward vs the oracle (the non-SG stash path)
    torch.manual_seed(11)
    m = make_model(4, 64, "barf", precision=prec)
    m.update_barf_alpha(3.5, "pts")
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-4.0)
    r, s = 300, 40
    o = torch.tensor([[0.0, 0.0, 1.5]]).repeat(r, 1) + torch.randn(r, 3) * 0.01
    d = torch.nn.functional.normalize(torch.randn(r, 3) * 0.2 + torch.tensor([0, 0, -1.0]), dim=-1)
    tgt = torch.rand(r)
    cfg = dict(num_early_layers=4, num_filters=64, pos_enc="barf", pos_enc_basis=5)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pix_c, _, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=0.5, far=2.5, n_samples=s, convention="acc")
    out = render_rays(m, o.to(DEV), d.to(DEV), s, 0.5, 2.5, mode="acc")
    torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
    got = grads(m)
    print(prec, "barf pix", f"{rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()):.2e}", "grad worst",
          f"{max(rel_l2(got[k], v.numpy()) for k, v in grads_c.items()):.2e}")
    # dense convention with per-ray z (1e10 tail), 4x128
    torch.manual_seed(7)
    m = make_model(4, 128, precision=prec)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-26.0)
    r, s = 257, 33
    pose = source_matrix(np.array([0, 0, 1500.0]), 40.0, -20.0)
    o_all, d_all = orc.get_rays(pose, 64, 64, 13.0 * 64)
    pick = torch.randperm(64 * 64)[:r]
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    z = torch.sort(torch.rand(r, s) * 200 + 1400, -1).values
    tgt = torch.rand(r)
    cfg = dict(num_early_layers=4, num_filters=128)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pix_c, _, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=0.0, far=0.0, n_samples=s, z=z, convention="dense")
    out = render_rays(m, o.to(DEV), d.to(DEV), mode="dense", z=z.to(DEV))
    torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
    got = grads(m)
    print(prec, "dense26 pix", f"{rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()):.2e}", "grad worst",
          f"{max(rel_l2(got[k], v.numpy()) for k, v in grads_c.items()):.2e}")
