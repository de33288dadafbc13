"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI via the Python
host layer, against golden vectors captured from the reference and against the CPU oracle.

Tolerances: fp32 mode ("f32", exact-fp32 MFMA): 1e-5 relative L2 on MLP outputs / pixels, 1e-4 on
gradients (north-star bar: 1e-4 on rendered projections and density grids)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def make_model(layers, width, pos_enc="none", precision="f32", basis=5):
    from nerf_for_angiography_amd.model.CPPN import CPPN
    md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3,
              num_output_channels=1, num_input_channels_views=0, use_bias=True, pos_enc=pos_enc,
              pos_enc_basis=basis, act_func="relu", fourier_sigma=5, num_img=1, device=torch.device(DEV),
              precision=precision)
    return CPPN(md).to(DEV)


def load_sd(model, g, prefix="sd__"):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    model.load_state_dict(sd, strict=False)
    return model


def test_native_library_is_what_runs():
    from nerf_for_angiography_amd import _lib
    lib = _lib.load()
    assert lib.afx_last_error is not None
    assert torch.cuda.is_available()


@pytest.mark.parametrize("name,layers,width", [("none_relu_4x64", 4, 64), ("none_relu_4x128", 4, 128),
                                               ("none_relu_8x256", 8, 256)])
def test_mlp_forward_golden(golden, name, layers, width):
    g = golden("g4_cppn_" + name)
    m = load_sd(make_model(layers, width), g)
    with torch.no_grad():
        y = m(T(g["x"]))
    assert y.shape == (g["x"].shape[0], 1)
    assert rel_l2(y.cpu().numpy(), g["y"]) < 1e-5


@pytest.mark.parametrize("name,layers,width,alphas", [("barf_relu_4x64", 4, 64, (0.0, 1.5, 2.5, 5.0)),
                                                      ("barf_relu_2x256", 2, 256, (2.5,))])
def test_mlp_forward_barf_golden(golden, name, layers, width, alphas):
    g = golden("g4_cppn_" + name)
    m = load_sd(make_model(layers, width, "barf"), g)
    for a in alphas:
        m.update_barf_alpha(a, "pts")
        assert np.array_equal(m.barf_weights.detach().cpu().numpy(), g[f"w_alpha{a}"])
        with torch.no_grad():
            y = m(T(g["x"]))
        assert rel_l2(y.cpu().numpy(), g[f"y_alpha{a}"]) < 2e-5, a


def test_mlp_forward_fourier_golden(golden):
    g = golden("g4_cppn_fourier_relu_4x64")
    m = load_sd(make_model(4, 64, "fourier"), g)
    with torch.no_grad():
        y = m(T(g["x"]))
    # sin/cos of arguments up to ~2*pi*100*15: fp32 argument rounding dominates; same bar as the oracle
    assert rel_l2(y.cpu().numpy(), g["y"]) < 5e-5


@pytest.mark.parametrize("enc", ["barf", "fourier"])
def test_encoding_large_coordinates(enc):
    """The kernels' own sin/cos (Cody-Waite reduction + minimax polynomials, csrc enc_sincos) at world-scale coordinates:
    |x| up to 2 000 puts the BARF arguments 2^k pi x at 1e5 rad.  Encoded first-layer inputs are compared through a linear probe:
    a 0-hidden-layer-equivalent check is not available, so the comparison is on the MLP output against the CPU oracle."""
    from oracle import angio_oracle as orc
    torch.manual_seed(21)
    m = make_model(2, 64, enc, precision="f32")
    if enc == "barf":
        m.update_barf_alpha(5.0, "pts")
    else:
        with torch.no_grad():
            m.fourier_coefficients.mul_(0.02)          # arguments 2 pi x coef ~ 1e3 rad
    pts = (torch.rand(8192, 3) * 2 - 1) * 2000.0
    cfg = dict(num_early_layers=2, num_filters=64, pos_enc=enc, pos_enc_basis=5)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = orc.cppn_forward(pts, cfg, params).reshape(-1).numpy()
    with torch.no_grad():
        got = m(pts.to(DEV)).reshape(-1).cpu().numpy()
    assert rel_l2(got, want) < 5e-5


# Tolerances (relative L2) per arithmetic mode.  "f32": exact-fp32 MFMA.  "bf16x3": split-bf16 forward
# (fp32-grade, meets the 1e-4 north-star bar on projections / density grids) with bf16 gradients.
# "f16": THE TRAINING / BENCHMARK PRECISION - f16 operands in the hidden layers (first layer split bf16), fp32
# accumulate; pixels of the 8x256 configurations sit at ~2e-5 (bar: 1e-4; asserted per test below), raw MLP outputs
# at 1.5e-4 ... 7.5e-4, gradients at <= 2e-3 (normalised f16 input-gradient chain).  The small 4x64 C1 fixture has
# fewer terms per sum to average the operand rounding over: its pixels sit at 1.3e-4, stated where it is used.
# "bf16": plain bf16 operands, fp32 accumulate (first layer split) - kept as the legacy throughput mode.
TOL = {"f32": dict(mlp=1e-5, pix=1e-5, grad=1e-4),
       "bf16x3": dict(mlp=5e-5, pix=1e-4, grad=3e-2),
       "f16": dict(mlp=1.5e-3, pix=1e-4, grad=1e-2),
       "f16s8": dict(mlp=1.5e-3, pix=1e-4, grad=1e-2),     # f16 arithmetic, bf8 backward stash (dZ' rounded stochastically): same pixels; whole-gradient error 2e-3 at full size
       "bf16": dict(mlp=3e-2, pix=1e-2, grad=6e-2)}
PIX_C1 = {"bf16x3": 1e-4, "f16": 2e-4, "f16s8": 2e-4, "bf16": 1e-2}      # the 4x64 / 32-sample C1 fixture (see above)


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "bf16"])
@pytest.mark.parametrize("name,layers,width", [("none_relu_4x64", 4, 64), ("none_relu_4x128", 4, 128),
                                               ("none_relu_8x256", 8, 256)])
def test_mlp_forward_golden_bf16(golden, name, layers, width, prec):
    g = golden("g4_cppn_" + name)
    m = load_sd(make_model(layers, width, precision=prec), g)
    with torch.no_grad():
        y = m(T(g["x"]))
    assert rel_l2(y.cpu().numpy(), g["y"]) < TOL[prec]["mlp"]


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "bf16"])
def test_mlp_forward_barf_golden_bf16(golden, prec):
    g = golden("g4_cppn_barf_relu_4x64")
    m = load_sd(make_model(4, 64, "barf", precision=prec), g)
    for a in (0.0, 2.5, 5.0):
        m.update_barf_alpha(a, "pts")
        with torch.no_grad():
            y = m(T(g["x"]))
        assert rel_l2(y.cpu().numpy(), g[f"y_alpha{a}"]) < 2 * TOL[prec]["mlp"], a


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "f16s8", "bf16"])
def test_fused_render_acc_golden_bf16(golden, prec):
    from nerf_for_angiography_amd.render import render_rays
    g, m, near, far, s = _c1(golden, prec)
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    out = render_rays(m, o, d, s, near, far, mode="acc")
    loss = torch.nn.functional.mse_loss(out.rgb_map, tgt)
    loss.backward()
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), g["acc_rgb"]) < PIX_C1[prec]
    got = _grads_by_name(m)
    assert set(got) == {k[len("acc_grad__"):] for k in g if k.startswith("acc_grad__")}
    for k in got:
        assert rel_l2(got[k], g["acc_grad__" + k]) < TOL[prec]["grad"], k


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "f16s8", "bf16"])
def test_c2_scale_vs_oracle_bf16(prec):
    """8x256 MLP, 64 / 50 samples per ray, ragged ray count, vs the CPU oracle; plus bit-identical re-runs."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd.phantomdata.proj_helpers import source_matrix
    torch.manual_seed(3)
    m = make_model(8, 256, precision=prec)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-5.0)
    w = 256
    pose = source_matrix(np.array([0, 0, 1500.0]), 30.0, 12.0)
    o_all, d_all = orc.get_rays(pose, w, w, 13.0 * w)
    pick = torch.randperm(w * w)[:1003]
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    tgt = torch.rand(o.shape[0])
    cfg = dict(num_early_layers=8, num_filters=256)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for s in (64, 50):
        pix_c, loss_c, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=1400.0, far=1600.0, n_samples=s,
                                                    convention="acc")
        m.zero_grad()
        out = render_rays(m, o.to(DEV), d.to(DEV), s, 1400.0, 1600.0, mode="acc")
        torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
        assert rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()) < TOL[prec]["pix"], s
        got = _grads_by_name(m)
        for k, v in grads_c.items():
            assert rel_l2(got[k], v.numpy()) < TOL[prec]["grad"], (s, k)
        g1 = {k: v.copy() for k, v in got.items()}
        m.zero_grad()
        out2 = render_rays(m, o.to(DEV), d.to(DEV), s, 1400.0, 1600.0, mode="acc")
        torch.nn.functional.mse_loss(out2.rgb_map, tgt.to(DEV)).backward()
        assert torch.equal(out.rgb_map, out2.rgb_map)
        for k, v in _grads_by_name(m).items():
            assert np.array_equal(v, g1[k]), k


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "f16s8", "bf16"])
def test_fused_train_step_matches_autograd_path(golden, prec):
    """afx_train_step_mse (in-kernel compositing + MSE gradient) == render -> mse_loss -> backward with the same
    bf16 backward kernel; and both sit within the bf16 gradient tolerance of the reference fixture."""
    from nerf_for_angiography_amd.render import render_rays, train_step_mse
    from nerf_for_angiography_amd.engine import RenderSpec
    g, m, near, far, s = _c1(golden, prec)
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    loss_a = torch.nn.functional.mse_loss(render_rays(m, o, d, s, near, far, mode="acc").rgb_map, tgt)
    loss_a.backward()
    ga = _grads_by_name(m)
    m.zero_grad()
    spec = RenderSpec(n_rays=o.shape[0], n_samples=s, origins=o, dirs=d, mode="acc", t_near=near, t_far=far)
    loss_f, pix = train_step_mse(m, spec, tgt)
    gf = _grads_by_name(m)
    btol = prec if prec in ("f16", "f16s8") else "bf16"       # bf16x3: the fused step's forward is the plain-bf16 backward kernel's
    assert rel_l2(pix.cpu().numpy(), g["acc_rgb"]) < PIX_C1[btol]
    np.testing.assert_allclose(float(loss_f), float(g["acc_loss"]), rtol=2e-2 if prec not in ("f16", "f16s8") else 5e-4)
    for k in gf:
        assert rel_l2(gf[k], g["acc_grad__" + k]) < TOL[btol]["grad"], k
        if prec in ("bf16", "f16", "f16s8"):      # same kernels, same pixels -> same gradients up to the fp32 loss-gradient rounding
            assert rel_l2(gf[k], ga[k]) < 1e-5, k
    # gradient accumulation semantics of loss.backward()
    train_step_mse(m, spec, tgt)
    for k, v in _grads_by_name(m).items():
        assert rel_l2(v, 2 * gf[k]) < 1e-6, k


def test_density_grid_bf16x3(golden):
    from nerf_for_angiography_amd.render import density_grid
    g = golden("g9_density_grid")
    m = load_sd(make_model(4, 64, precision="bf16x3"), g)
    assert rel_l2(density_grid(m, 100.0, 16).cpu().numpy(), g["sigma"]) < 1e-4


def _c1(golden, precision="f32"):
    g = golden("g8_e2e_c1")
    m = load_sd(make_model(4, 64, precision=precision), g, "init__")
    near, far, s = g["near_far_s"]
    return g, m, float(near), float(far), int(s)


def _grads_by_name(model):
    return {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}


def test_fused_render_acc_golden(golden):
    """Fused forward+backward, training convention (nerf_helpers_acc.py) vs the reference-built fixture."""
    from nerf_for_angiography_amd.render import render_rays
    g, m, near, far, s = _c1(golden)
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    out = render_rays(m, o, d, s, near, far, mode="acc")
    loss = torch.nn.functional.mse_loss(out.rgb_map, tgt)
    loss.backward()
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), g["acc_rgb"]) < 1e-5
    np.testing.assert_allclose(float(loss.detach()), float(g["acc_loss"]), rtol=1e-5)
    got = _grads_by_name(m)
    for k in got:
        assert rel_l2(got[k], g["acc_grad__" + k]) < 1e-4, k
    assert set(got) == {k[len("acc_grad__"):] for k in g if k.startswith("acc_grad__")}


def test_fused_render_dense_golden(golden):
    """Dense convention incl. the 1e10 tail: bias -26 fixture has non-trivial pixels/grads."""
    from nerf_for_angiography_amd.render import render_rays
    g, m, near, far, s = _c1(golden)
    o, d, tgt, z = T(g["o"]), T(g["d"]), T(g["target"]), T(g["z"])
    with torch.no_grad():
        pix = render_rays(m, o, d, mode="dense", z=z).rgb_map
    assert float(pix.abs().max()) == 0.0 and float(np.abs(g["dense_rgb"]).max()) == 0.0      # D3 literal
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-26.0)
    out = render_rays(m, o, d, mode="dense", z=z, want_aux=True)
    loss = torch.nn.functional.mse_loss(out.rgb_map, tgt)
    loss.backward()
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), g["dense26_rgb"]) < 1e-5
    assert rel_l2(out.weights.cpu().numpy(), g["dense26_weights"]) < 1e-4
    got = _grads_by_name(m)
    for k in got:
        assert rel_l2(got[k], g["dense26_grad__" + k]) < 1e-4, k


@pytest.mark.parametrize("fused", [False, True])
def test_adam_steps_golden(golden, fused):
    """10 optimizer steps (PyTorch Adam on the flat-buffer views; its default implementation and the fused multi-tensor kernel the training
    driver selects) reproduce the reference's weights."""
    from nerf_for_angiography_amd.render import render_rays
    g, m, near, far, s = _c1(golden)
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    opt = torch.optim.Adam(list(m.parameters()), lr=1e-4, fused=fused)
    for it in range(10):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(render_rays(m, o, d, s, near, far, mode="acc").rgb_map, tgt)
        loss.backward()
        opt.step()
        for pg in opt.param_groups:
            pg["lr"] = 1e-4 * (0.1 ** (it / 500000))
        if it in (0, 9):
            np.testing.assert_allclose(float(loss.detach()), float(g[f"acc_step{it + 1}_loss"]), rtol=2e-5)
            sd = m.state_dict()
            for k in sd:
                if k.startswith("early") or k.startswith("output"):
                    assert rel_l2(sd[k].cpu().numpy(), g[f"acc_step{it + 1}__{k}"]) < 5e-6, (it, k)


@pytest.mark.parametrize("prec", ["f32", "f16s8"])
def test_unfused_reference_flow(golden, prec):
    """The literal call sequence of run_nerf_acc.py:287-306 through the mirrored helpers:
    acc_ray_marching -> positions -> get_predictions -> acc_render_volume_density -> mse -> backward.
    (f16s8 in points mode = the f16 arithmetic with the 16-bit stash: the 8-bit stash is a rays-mode path.)"""
    from nerf_for_angiography_amd.nerf.nerf_helpers import get_predictions
    from nerf_for_angiography_amd.nerf.nerf_helpers_acc import acc_ray_marching, acc_render_volume_density
    g, m, near, far, s = _c1(golden, prec)
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    with torch.no_grad():
        ri, ts, te = acc_ray_marching(m, None, None, o, d, s, near, far)
    pos = o[ri.long()] + d[ri.long()] * (ts + te) / 2.0
    pred = get_predictions(m, pos, 8192)
    pix, ent = acc_render_volume_density(pred, ri, ts, te, o.shape[0], s)
    assert ent is None
    loss = torch.nn.functional.mse_loss(pix, tgt)
    loss.backward()
    assert rel_l2(pix.detach().cpu().numpy(), g["acc_rgb"]) < (1e-5 if prec == "f32" else PIX_C1[prec])
    got = _grads_by_name(m)
    for k in got:
        assert rel_l2(got[k], g["acc_grad__" + k]) < (1e-4 if prec == "f32" else TOL["f16"]["grad"]), k


def test_pose_raygen_matches_arrays(golden):
    """In-kernel get_ray_values (float64 pose -> fp32 rays) == rays supplied as arrays (fixture G2)."""
    from nerf_for_angiography_amd.render import render_rays, render_projection
    g2 = golden("g2_rays")
    g, m, near, far, s = _c1(golden)
    for tag in ("a", "b", "c"):
        w, h, f = g2[f"{tag}_whf"]
        w, h = int(w), int(h)
        pose = T(g2[f"{tag}_pose"][None])
        o = T(g2[f"{tag}_o"].reshape(-1, 3).astype(np.float32))
        d = T(g2[f"{tag}_d"].reshape(-1, 3).astype(np.float32))
        nf = (float(f) + 100.0, float(f) + 300.0)
        with torch.no_grad():
            a = render_rays(m, o, d, 32, nf[0], nf[1], mode="acc").rgb_map
            b = render_projection(m, pose, w, h, float(f), 32, nf[0], nf[1]).rgb_map
            ids = torch.arange(w * h - 1, -1, -1, dtype=torch.int32, device=DEV)
            c = render_projection(m, pose, w, h, float(f), 32, nf[0], nf[1], ray_ids=ids).rgb_map
        assert torch.equal(a, b), tag
        assert torch.equal(a, c.flip(0)), tag


def test_composite_dense_golden(golden):
    from nerf_for_angiography_amd.nerf.nerf_helpers import render_volume_density
    g = golden("g5_render")
    for rk in ("n", "m40", "m3", "tail"):
        for zk in ("z1", "z2"):
            rgb, dep, w, ent, (sig, _) = render_volume_density(T(g["raw_" + rk]), T(g["d"]), T(g[zk]))
            for name, got in (("rgb", rgb), ("depth", dep), ("weights", w), ("entropy", ent), ("sigma", sig)):
                np.testing.assert_allclose(got.cpu().numpy(), g[f"{rk}_{zk}_{name}"], rtol=2e-5, atol=1e-30,
                                           err_msg=f"{rk} {zk} {name}")


def test_composite_dense_backward_vs_oracle(golden):
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.nerf.nerf_helpers import render_volume_density
    g = golden("g5_render")
    raw_c = torch.from_numpy(g["raw_tail"]).clone().requires_grad_(True)
    rgb_c = orc.render_volume_density(raw_c, torch.from_numpy(g["d"]), torch.from_numpy(g["z2"]))[0]
    wgt = torch.linspace(0.5, 1.5, rgb_c.shape[0])
    (rgb_c * wgt).sum().backward()
    raw_g = T(g["raw_tail"]).clone().requires_grad_(True)
    rgb_g = render_volume_density(raw_g, T(g["d"]), T(g["z2"]))[0]
    (rgb_g * wgt.to(DEV)).sum().backward()
    assert rel_l2(raw_g.grad.cpu().numpy(), raw_c.grad.numpy()) < 1e-5


def test_fine_depths_vs_oracle(golden):
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd import engine
    g = golden("g7_sample_pdf")
    for tag, s in (("a", 32), ("b", 128)):
        r = g[f"{tag}_w"].shape[0]
        z = torch.linspace(0, 1, s) * 200 + 1400
        wc = torch.cat([torch.zeros(r, 1), torch.from_numpy(g[f"{tag}_w"]), torch.zeros(r, 1)], 1)
        u = torch.from_numpy(g[f"{tag}_u"])
        want = orc.fine_depths(z, wc, u, r)
        got = engine.fine_depths(z.to(DEV), wc.to(DEV), u.to(DEV)).cpu()
        assert got.shape == want.shape
        assert float((got[:, 1:] - got[:, :-1]).min()) >= 0.0           # sorted
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=8e-3)   # inverse CDF amplifies 1-ulp cdf differences
        assert rel_l2(got.numpy(), want.numpy()) < 1e-6
        # the reference's own sample_pdf output (fixture) is a subset of the merged depths
        samp = np.sort(g[f"{tag}_out"], -1)
        merged = got.numpy()
        for row in (0, 1, 2, r - 1):
            gap = np.abs(merged[row][None, :] - samp[row][:, None]).min(-1)
            assert float(gap.max()) < 8e-3, (tag, row)


def test_density_grid_golden(golden):
    from nerf_for_angiography_amd.render import density_grid
    g = golden("g9_density_grid")
    m = load_sd(make_model(4, 64), g)
    grid = density_grid(m, 100.0, 16)
    assert grid.shape == (17, 17, 17)
    assert rel_l2(grid.cpu().numpy(), g["sigma"]) < 1e-5


def test_c2_scale_vs_oracle_and_determinism():
    """256x256-class workload (config C2: 64 samples/ray, 8x256 MLP), rays sub-sampled so the CPU oracle
    finishes in seconds; ragged sizes (R not a multiple of the tile, S not a multiple of 32)."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd.phantomdata.proj_helpers import source_matrix
    torch.manual_seed(3)
    m = make_model(8, 256)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-5.0)
    w = 256
    pose = source_matrix(np.array([0, 0, 1500.0]), 30.0, 12.0)
    o_all, d_all = orc.get_rays(pose, w, w, 13.0 * w)
    pick = torch.randperm(w * w)[:1003]
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    tgt = torch.rand(o.shape[0])
    cfg = dict(num_early_layers=8, num_filters=256)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for s in (64, 50):
        pix_c, loss_c, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=1400.0, far=1600.0, n_samples=s,
                                                    convention="acc")
        m.zero_grad()
        out = render_rays(m, o.to(DEV), d.to(DEV), s, 1400.0, 1600.0, mode="acc")
        loss = torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV))
        loss.backward()
        assert rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()) < 1e-5, s
        got = _grads_by_name(m)
        for k, v in grads_c.items():
            assert rel_l2(got[k], v.numpy()) < 1e-4, (s, k)
        # bit-identical re-run (no float atomics anywhere)
        g1 = {k: v.copy() for k, v in got.items()}
        m.zero_grad()
        out2 = render_rays(m, o.to(DEV), d.to(DEV), s, 1400.0, 1600.0, mode="acc")
        torch.nn.functional.mse_loss(out2.rgb_map, tgt.to(DEV)).backward()
        assert torch.equal(out.rgb_map, out2.rgb_map)
        for k, v in _grads_by_name(m).items():
            assert np.array_equal(v, g1[k]), k


def test_backward_chunking_is_invisible():
    """A workspace that forces several backward chunks gives the same gradients as one chunk."""
    from nerf_for_angiography_amd.render import render_rays
    torch.manual_seed(5)
    m = make_model(4, 128)
    o = torch.randn(700, 3, device=DEV) * 5 + torch.tensor([0, 0, 1500.0], device=DEV)
    d = torch.nn.functional.normalize(torch.randn(700, 3, device=DEV) * 0.02 + torch.tensor([0, 0, -1.0], device=DEV), dim=-1)
    tgt = torch.rand(700, device=DEV)

    def grads(ws_bytes):
        m.engine.max_workspace_bytes = ws_bytes
        m.engine._ws = None
        m.zero_grad()
        torch.nn.functional.mse_loss(render_rays(m, o, d, 64, 1400.0, 1600.0).rgb_map, tgt).backward()
        return _grads_by_name(m)

    full = grads(8 << 30)
    lib = m.engine.lib
    small = int(lib.afx_query(m.engine.h, 4, 700, 0, 0)) + 40 * 128 * 4 * (2 * 5 * 128 + 5)
    part = grads(small)
    for k in full:
        assert rel_l2(part[k], full[k]) < 2e-6, k


def test_argument_validation():
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd._lib import AfxError
    m = make_model(4, 64)
    o = torch.zeros(8, 3, device=DEV)
    with pytest.raises(ValueError):
        render_rays(m, o, torch.zeros(7, 3, device=DEV), 32, 0.0, 1.0)
    with pytest.raises(ValueError):
        render_rays(m, o.double(), o, 32, 0.0, 1.0)
    with pytest.raises(AfxError):
        render_rays(m, o, o, 32, 1.0, 1.0)          # far <= near
    with pytest.raises(AfxError):
        render_rays(m, o, o, 1, 0.0, 1.0)           # too few samples
    # empty batch
    out = render_rays(m, o[:0], o[:0], 32, 0.0, 1.0)
    assert out.rgb_map.shape == (0,)
    # afx_set_encoding_grad: only the fourier encoding has trainable coefficients; params are required with a target
    buf = torch.zeros(15, device=DEV)
    with pytest.raises(AfxError):
        with m.engine.encoding_grad(m.flat_params, buf):
            pass
    mb = make_model(4, 64, "barf", precision="f16")
    with pytest.raises(AfxError):
        with mb.engine.encoding_grad(mb.flat_params, buf):
            pass
    mf = make_model(4, 64, "fourier", precision="f16")
    assert mf.engine.lib.afx_set_encoding_grad(mf.engine.h, None, buf.data_ptr()) != 0       # target without params
    with mf.engine.encoding_grad(mf.flat_params, buf):
        pass                                                                                   # accepted, and switched off again


# ------------------------------------------------------------------------------------------------
# Encoded inputs (BARF) through forward AND backward; hierarchical pipeline; full-size properties; graphs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "bf16x3", "f16", "f16s8", "bf16"])
def test_barf_backward_vs_oracle(prec):
    """K0 = 33 encoded inputs: first-layer MFMA k-steps, encoded-input stash (fp32 / 16-bit / bf8) and the first-layer weight
    gradient as an extra grid row of the weight-gradient kernels (k_wgrad_f32 / k_wgrad_bf16 / k_wgrad_s8)."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays
    torch.manual_seed(11)
    m = make_model(4, 64, "barf", precision=prec)
    m.update_barf_alpha(3.5, "pts")
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-4.0)
    r, s = 300, 40
    o = torch.tensor([[0.0, 0.0, 1.5]]).repeat(r, 1) + torch.randn(r, 3) * 0.01      # unit-scale scene: sin/cos args stay small
    d = torch.nn.functional.normalize(torch.randn(r, 3) * 0.2 + torch.tensor([0, 0, -1.0]), dim=-1)
    tgt = torch.rand(r)
    cfg = dict(num_early_layers=4, num_filters=64, pos_enc="barf", pos_enc_basis=5)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pix_c, _, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=0.5, far=2.5, n_samples=s, convention="acc")
    out = render_rays(m, o.to(DEV), d.to(DEV), s, 0.5, 2.5, mode="acc")
    torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()) < TOL[prec]["pix"]
    got = _grads_by_name(m)
    assert set(grads_c) <= set(got)
    for k, v in grads_c.items():
        assert rel_l2(got[k], v.numpy()) < TOL[prec]["grad"], k


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "f16s8", "bf16"])
def test_fourier_coefficients_train(prec):
    """model/CPPN.py:92 makes fourier_coefficients an nn.Parameter, so the reference's Adam updates them: their gradient
    (second pass of the first-layer kernel against d enc / d coef) vs the oracle's autograd, through the three backward
    entry points (render + autograd, fused train step, points mode)."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays, train_step_mse
    from nerf_for_angiography_amd.engine import RenderSpec
    torch.manual_seed(13)
    m = make_model(4, 64, "fourier", precision=prec)
    with torch.no_grad():
        m.fourier_coefficients.mul_(0.1)          # sigma 0.5: arguments of a few radians on a unit-scale scene
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-4.0)
    r, s = 256, 32
    o = torch.tensor([[0.0, 0.0, 1.5]]).repeat(r, 1) + torch.randn(r, 3) * 0.01
    d = torch.nn.functional.normalize(torch.randn(r, 3) * 0.2 + torch.tensor([0, 0, -1.0]), dim=-1)
    tgt = torch.rand(r)
    cfg = dict(num_early_layers=4, num_filters=64, pos_enc="fourier", pos_enc_basis=5)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pix_c, _, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=0.5, far=2.5, n_samples=s, convention="acc")
    assert "fourier_coefficients" in grads_c
    want = grads_c["fourier_coefficients"].numpy()
    tol = 2 * TOL[prec]["grad"]

    out = render_rays(m, o.to(DEV), d.to(DEV), s, 0.5, 2.5, mode="acc")
    torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()) < TOL[prec]["pix"]
    got = _grads_by_name(m)
    for k, v in grads_c.items():
        assert rel_l2(got[k], v.numpy()) < tol, k
    g_autograd = m.fourier_coefficients.grad.detach().cpu().numpy().copy()

    # the fused train step accumulates the same gradient into .grad
    m.zero_grad(set_to_none=True)
    spec = RenderSpec(n_rays=r, n_samples=s, origins=o.to(DEV), dirs=d.to(DEV), mode="acc", t_near=0.5, t_far=2.5)
    train_step_mse(m, spec, tgt.to(DEV))
    assert rel_l2(m.fourier_coefficients.grad.cpu().numpy(), want) < tol
    assert rel_l2(m.fourier_coefficients.grad.cpu().numpy(), g_autograd) < TOL[prec]["grad"]      # bf16x3: the train step renders with the backward kernel's own forward

    # points mode: d(sum c_p raw_p)/d coef
    m.zero_grad(set_to_none=True)
    pts = (torch.rand(1000, 3) * 2 - 1)
    c = torch.randn(1000, 1)
    (m(pts.to(DEV)) * c.to(DEV)).sum().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items() if v.dtype.is_floating_point}
    (orc.cppn_forward(pts, cfg, {**params, **leaves}) * c).sum().backward()
    # random-sign cotangents: the sum cancels to ~1/sqrt(P) of its terms, which magnifies the operand rounding (f16: 2.2e-2)
    assert rel_l2(m.fourier_coefficients.grad.cpu().numpy(), leaves["fourier_coefficients"].grad.numpy()) < 2 * tol

    # frozen coefficients: no gradient, and the f32 kernels refuse to train them
    m.fourier_coefficients.requires_grad_(False)
    m.zero_grad(set_to_none=True)
    render_rays(m, o.to(DEV), d.to(DEV), s, 0.5, 2.5, mode="acc").rgb_map.sum().backward()
    assert m.fourier_coefficients.grad is None
    m32 = make_model(4, 64, "fourier", precision="f32")
    with pytest.raises(NotImplementedError):
        render_rays(m32, o.to(DEV), d.to(DEV), s, 0.5, 2.5, mode="acc").rgb_map.sum().backward()


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_hierarchical_coarse_fine_vs_oracle(prec):
    """Config-C3 style pipeline: coarse dense render -> weights -> sample_pdf/merge (afx_fine_depths) -> fine render
    with per-ray depths, against the oracle's composition of the same reference pieces."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd.nerf.nerf_helpers import fine_sampling
    torch.manual_seed(13)
    m = make_model(4, 64, precision=prec)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(8.0)
        m.output_linear[0].bias.fill_(-6.0)
    r, sc, nf = 200, 32, 16
    o = torch.tensor([[0.0, 0.0, 1500.0]]).repeat(r, 1)
    d = torch.nn.functional.normalize(torch.randn(r, 3) * 0.03 + torch.tensor([0, 0, -1.0]), dim=-1) * 1.001
    z = orc.depth_values(1400.0, 1600.0, sc)
    u = torch.rand(r, nf)
    cfg = dict(num_early_layers=4, num_filters=64)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    fn = lambda p: orc.cppn_forward(p, cfg, params)
    raw_c = orc.get_predictions(fn, orc.points_dense(o, d, z).reshape(-1, 3), 8192).reshape(r, sc, 1)
    _, _, w_c, _, _ = orc.render_volume_density(raw_c, d, z)
    zf_c = orc.fine_depths(z, w_c, u, r)
    raw_f = orc.get_predictions(fn, orc.points_dense(o, d, zf_c).reshape(-1, 3), 8192).reshape(r, sc + nf, 1)
    _, dep_c, wf_c, ent_c, _ = orc.render_volume_density(raw_f, d, zf_c)
    with torch.no_grad():
        coarse = render_rays(m, o.to(DEV), d.to(DEV), mode="dense", z=z.to(DEV), want_aux=True)
        assert rel_l2(coarse.weights.cpu().numpy(), w_c.numpy()) < 10 * TOL[prec]["pix"]
        rgb_f, dep_f, ent_f = fine_sampling(z.to(DEV), coarse.weights, o.to(DEV), d.to(DEV), m, None, nf, 8192, u=u.to(DEV))
    assert float(rgb_f.abs().max()) == 0.0                      # D3: the 1e10 tail zeroes the projection
    assert rel_l2(dep_f.cpu().numpy(), dep_c.numpy()) < 1e-4
    assert rel_l2(ent_f.cpu().numpy(), ent_c.numpy()) < 1e-3


def _bench_model(precision, seed=0):
    torch.manual_seed(seed)
    m = make_model(8, 256, precision=precision)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-5.0)
    return m


@pytest.mark.parametrize("enc", ["barf", "fourier"])
def test_full_size_encoded_gradients(enc):
    """BASELINE size C4 with a positional encoding (33 encoded inputs) at the default precision: the fused f16s8 step (bf8 input
    stash, first layer and - fourier - the coefficient contraction as rows of k_wgrad_s8) against the fp32-mode backward over all
    33.5 M samples; pixels against the fp32 forward."""
    from nerf_for_angiography_amd.render import render_projection, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W, S = 512, 128
    _, _, m44, _, _ = get_ray_values(24.0, 8.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, "cpu")
    poses = torch.from_numpy(m44[None]).to(DEV)
    tgt = torch.rand(W * W, generator=torch.Generator().manual_seed(9)).to(DEV)

    def model(prec):
        torch.manual_seed(0)
        m = make_model(8, 256, enc, precision=prec)
        if enc == "barf":
            m.update_barf_alpha(2.5, "pts")
        else:
            with torch.no_grad():
                m.fourier_coefficients.mul_(0.002)         # 2 pi x coef of a few radians at |x| ~ 100
            m.fourier_coefficients.requires_grad_(prec != "f32")      # the fp32 kernels take them as constants
        with torch.no_grad():
            m.output_linear[0].weight.mul_(4.0)
            m.output_linear[0].bias.fill_(-5.0)
        return m

    m32 = model("f32")
    out = render_projection(m32, poses, W, W, 13.0 * W, S, 1400.0, 1600.0)
    (((out.rgb_map - tgt) ** 2).sum() / (W * W)).backward()
    ref = {k: p.grad.double() for k, p in m32.named_parameters() if p.grad is not None}
    pix32 = out.rgb_map.detach()
    del m32, out
    m = model("f16s8")
    _, pix = train_step_mse(m, projection_spec(poses, W, W, 13.0 * W, S, 1400.0, 1600.0), tgt)
    assert rel_l2(pix.cpu().numpy(), pix32.cpu().numpy()) < 1e-4
    got = {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}
    tot = float(torch.sqrt(sum(((got[k] - ref[k]) ** 2).sum() for k in ref)) / torch.sqrt(sum((ref[k] ** 2).sum() for k in ref)))
    assert tot < TOL["f16s8"]["grad"], tot
    for k in ref:
        assert float((got[k] - ref[k]).norm() / ref[k].norm()) < 2 * TOL["f16s8"]["grad"], k
    if enc == "fourier":      # the coefficient gradient: 8-bit path (k_wgrad_s8 row) against the 16-bit path (k_wgrad_bf16 row)
        g = m.fourier_coefficients.grad
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
        g8 = g.double().clone()
        del m
        m16 = model("f16")
        train_step_mse(m16, projection_spec(poses, W, W, 13.0 * W, S, 1400.0, 1600.0), tgt)
        g16 = m16.fourier_coefficients.grad.double()
        assert float((g8 - g16).norm() / g16.norm()) < 2 * TOL["f16s8"]["grad"]


def test_full_size_projection_properties():
    """BASELINE size C4 (512x512 rays x 128 samples, 8x256) at the TRAINING precision (f16), with no CPU oracle in reach:
    the f16 pixels against the exact-fp32 MFMA path on every pixel (relative L2 <= 1e-4 - the north-star bar - and max
    abs), in-kernel ray generation == array rays, the fused train step's pixels == the forward-only kernel's, its
    gradients against the fp32-mode backward over all 33.5 M samples, and independence of the workspace chunking."""
    from nerf_for_angiography_amd.render import render_projection, render_rays, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W = 512
    m = _bench_model("f16")
    o, d, m44, _, _ = get_ray_values(24.0, 8.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, DEV)
    poses = torch.from_numpy(m44[None]).to(DEV)
    tgt = torch.rand(W * W, device=DEV)
    with torch.no_grad():
        a = render_projection(m, poses, W, W, 13.0 * W, 128, 1400.0, 1600.0).rgb_map
        b = render_rays(m, o.reshape(-1, 3).float().contiguous(), d.reshape(-1, 3).float().contiguous(), 128, 1400.0, 1600.0).rgb_map
        m.precision = "bf16x3"
        x3 = render_projection(m, poses, W, W, 13.0 * W, 128, 1400.0, 1600.0).rgb_map
    assert a.shape == (W * W,) and torch.equal(a, b)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 and 1e-3 < float(a.std())
    # fp32 mode: pixels and the gradients of the same loss
    m.precision = "f32"
    m.zero_grad()
    out = render_projection(m, poses, W, W, 13.0 * W, 128, 1400.0, 1600.0)
    torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
    c = out.rgb_map.detach()
    g32 = {k: v.copy() for k, v in _grads_by_name(m).items()}
    assert rel_l2(a.cpu().numpy(), c.cpu().numpy()) < 1e-4
    assert float((a - c).abs().max()) < 2e-4
    assert rel_l2(x3.cpu().numpy(), c.cpu().numpy()) < 1e-5
    # training step: fused f16 step vs fp32 gradients, and chunking invariance (3 chunks vs 1)
    m.precision = "f16"
    spec = projection_spec(poses, W, W, 13.0 * W, 128, 1400.0, 1600.0)
    m.zero_grad()
    loss1, pix1 = train_step_mse(m, spec, tgt)
    g1 = {k: v.copy() for k, v in _grads_by_name(m).items()}
    assert float((pix1 - a).abs().max()) < 1e-6          # the step's forward IS the rendering arithmetic
    for k, v in g32.items():
        assert rel_l2(g1[k], v) < TOL["f16"]["grad"], k
    m.precision = "f16s8"                                # bf8 stash: same pixels, gradients within its own tolerance
    m.zero_grad()
    loss8, pix8 = train_step_mse(m, spec, tgt)
    g8 = _grads_by_name(m)
    assert torch.equal(pix8, pix1)
    for k, v in g32.items():
        assert rel_l2(g8[k], v) < TOL["f16s8"]["grad"], k
    m.precision = "f16"
    m.engine.max_workspace_bytes = 9 << 30
    m.engine._ws = None
    m.zero_grad()
    loss2, pix2 = train_step_mse(m, spec, tgt)
    g2 = _grads_by_name(m)
    assert torch.equal(pix1, pix2)
    for k in g1:
        assert rel_l2(g2[k], g1[k]) < 1e-5, k


@pytest.mark.parametrize("prec,enc", [("f16", "none"), ("f16s8", "none"), ("bf16", "none"), ("f16s8", "barf"), ("f16", "barf")])
def test_production_build_is_bit_identical_to_the_safe_waits_build(prec, enc):
    """Race detector for the hand-counted s_waitcnt vmcnt / lgkmcnt protocol of the chain kernels: libafx_safe.so is the
    same source with every counted wait replaced by a full one (-DAFX_SAFE_WAITS).  A deterministic under-wait would
    survive a run-to-run comparison; it cannot survive this one.  512x512 x 128, 8x256: pixels and every gradient."""
    from nerf_for_angiography_amd import build as afx_build
    from nerf_for_angiography_amd.engine import Engine
    from nerf_for_angiography_amd.render import render_projection, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    afx_build.build(variant="safe")
    W = 512
    _, _, m44, _, _ = get_ray_values(40.0, 3.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, DEV)
    poses = torch.from_numpy(m44[None]).to(DEV)
    tgt = torch.rand(W * W, device=DEV)
    spec = projection_spec(poses, W, W, 13.0 * W, 128, 1400.0, 1600.0)
    res = []
    for variant in ("", "safe"):
        torch.manual_seed(0)
        m = make_model(8, 256, enc, precision=prec)
        with torch.no_grad():
            m.output_linear[0].weight.mul_(4.0)
            m.output_linear[0].bias.fill_(-5.0)
        if enc == "barf":
            m.update_barf_alpha(2.5, "pts")
        m._engine = Engine(256, 8, enc, 5 if enc != "none" else 0, variant=variant)
        with torch.no_grad():
            fwd = render_projection(m, poses, W, W, 13.0 * W, 128, 1400.0, 1600.0).rgb_map
        loss, pix = train_step_mse(m, spec, tgt)
        torch.cuda.synchronize()
        res.append((fwd, pix, {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
        del m
    assert torch.equal(res[0][0], res[1][0]), int((res[0][0] != res[1][0]).sum())
    assert torch.equal(res[0][1], res[1][1]), int((res[0][1] != res[1][1]).sum())
    for k, g in res[0][2].items():
        assert torch.equal(g, res[1][2][k]), k


@pytest.mark.parametrize("prec", ["f16s8", "f16"])
def test_large_chunks_keep_32bit_stash_offsets_in_range(prec):
    """Regression: the 16-/8-bit chain kernels address a layer's stash with 32-bit offsets; a shallow model with a large
    workspace used to get chunks whose layer plane exceeded 4 GiB (rows beyond it wrapped around: wrong weight gradients,
    nothing else).  33.5 M samples, 1x256 MLP: with the default 24 GiB workspace the 4 stash planes of a chunk were 6 GiB (f16s8: one
    chunk; f16: two) before the fix; compared with 3 GiB of workspace (small chunks)."""
    from nerf_for_angiography_amd.render import train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W, S = 512, 128
    _, _, m44, _, _ = get_ray_values(20.0, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, "cpu")
    spec = projection_spec(torch.from_numpy(m44[None]).to(DEV), W, W, 13.0 * W, S, 1400.0, 1600.0)
    tgt = torch.rand(W * W, generator=torch.Generator().manual_seed(3)).to(DEV)
    grads = {}
    for ws_gib in (24, 3):
        torch.manual_seed(5)
        m = make_model(1, 256, precision=prec)
        with torch.no_grad():
            m.output_linear[0].weight.mul_(4.0)
            m.output_linear[0].bias.fill_(-5.0)
        m.engine.max_workspace_bytes = ws_gib << 30
        m.engine._workspace(ws_gib << 30, torch.device(DEV)).fill_(255)      # stale contents must not matter either
        train_step_mse(m, spec, tgt)
        grads[ws_gib] = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).cpu().numpy()
        assert np.isfinite(grads[ws_gib]).all()
        del m
        torch.cuda.empty_cache()
    assert rel_l2(grads[24], grads[3]) < 1e-5          # chunking only changes the order of fp32 partial sums


@pytest.mark.parametrize("prec", ["f16s8", "f16", "bf16", "f32"])
def test_results_do_not_depend_on_stale_workspace(prec):
    """Every byte a backward kernel reads must have been written by the same call: the same step with the workspace pre-filled
    with zeros, with 0xFF bytes (NaN patterns in every format) and with random bytes gives bit-identical pixels and gradients -
    several ray chunks, rays mode with and without the fused step, with and without an input encoding."""
    from nerf_for_angiography_amd.render import render_rays, train_step_mse
    from nerf_for_angiography_amd.engine import RenderSpec
    dev = torch.device(DEV)
    ws_bytes = 48 << 20

    def run(fill, mode, enc, n_samples):
        torch.manual_seed(0)
        m = make_model(3, 64, enc, precision=prec)
        with torch.no_grad():
            m.output_linear[0].weight.mul_(4.0)
            m.output_linear[0].bias.fill_(-4.0)
        m.engine.max_workspace_bytes = ws_bytes
        ws = m.engine._workspace(ws_bytes, dev)
        if fill == "zero":
            ws.zero_()
        elif fill == "ff":
            ws.fill_(255)
        else:
            ws.random_(0, 256)
        g = torch.Generator().manual_seed(1)
        r = 1500
        o = (torch.tensor([[0.0, 0.0, 1.5]]).repeat(r, 1) + torch.randn(r, 3, generator=g) * 0.01).to(dev)
        d = torch.nn.functional.normalize(torch.randn(r, 3, generator=g) * 0.2 + torch.tensor([0, 0, -1.0]), dim=-1).to(dev)
        tgt = torch.rand(r, generator=g).to(dev)
        if mode == "points":
            pts = (torch.rand(40000, 3, generator=g) * 2 - 1).to(dev)
            pix = m(pts)
            (pix * torch.randn(40000, 1, generator=g).to(dev)).sum().backward()
            pix = pix.detach()
        elif mode == "fused":
            _, pix = train_step_mse(m, RenderSpec(n_rays=r, n_samples=n_samples, origins=o, dirs=d, mode="acc", t_near=0.5, t_far=2.5), tgt)
        else:
            out = render_rays(m, o, d, n_samples, 0.5, 2.5, mode="acc")
            torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
            pix = out.rgb_map.detach()
        return pix.cpu(), torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).cpu()

    for mode, n_samples in (("autograd", 96), ("fused", 64), ("points", 0)):
        if mode == "fused" and prec == "f32":
            continue
        for enc in ("none", "barf"):
            ref = run("zero", mode, enc, n_samples)
            assert float(ref[1].abs().max()) > 0
            for fill in ("ff", "rand"):
                got = run(fill, mode, enc, n_samples)
                assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (mode, enc, fill)


def test_c3_full_size_hierarchical():
    """BASELINE config C3: 512x512 rays, 128 coarse + 64 fine samples, 8x256.  The dense convention's 1e10 tail makes
    rgb_map == 0 at ordinary weights, so the output bias is -26 as in the reference-captured dense26 fixture.
    Full size: f16 against the fp32 mode on every ray (coarse weights -> fine depths -> fine render); a sub-sample of rays
    against the CPU oracle's composition of the reference pieces."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_projection, render_rays
    from nerf_for_angiography_amd.engine import fine_depths
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W, SC, NF = 512, 128, 64
    m = _bench_model("f16", seed=5)
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-26.0)
    o, d, m44, _, _ = get_ray_values(100.0, -20.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, DEV)
    poses = torch.from_numpy(m44[None]).to(DEV)
    z = orc.depth_values(1400.0, 1600.0, SC).to(DEV)
    u = torch.rand(W * W, NF, generator=torch.Generator().manual_seed(7)).to(DEV)
    res = {}
    for prec in ("f16", "f32"):
        m.precision = prec
        with torch.no_grad():
            coarse = render_projection(m, poses, W, W, 13.0 * W, SC, 0.0, 0.0, mode="dense", z=z, want_aux=True)
            zf = fine_depths(z, coarse.weights, u)
            fine = render_projection(m, poses, W, W, 13.0 * W, SC + NF, 0.0, 0.0, mode="dense", z=zf, want_aux=True)
        res[prec] = (coarse.weights, zf, fine.rgb_map, fine.depth_map, fine.weights)
    assert torch.all(res["f16"][1][:, 1:] >= res["f16"][1][:, :-1])                      # merged depths are sorted
    assert float(res["f32"][2].max()) > 1e-3                                              # non-degenerate pixels
    assert rel_l2(res["f16"][0].cpu().numpy(), res["f32"][0].cpu().numpy()) < 1e-3       # coarse weights
    assert rel_l2(res["f16"][2].cpu().numpy(), res["f32"][2].cpu().numpy()) < 2e-3       # pixels: exp(-sigma 1e10 ||d||) amplifies
    assert rel_l2(res["f16"][3].cpu().numpy(), res["f32"][3].cpu().numpy()) < 1e-4       # depth map
    # sub-sampled rays vs the oracle (f32 mode: the strict parity path)
    pick = torch.randperm(W * W, generator=torch.Generator().manual_seed(3))[:384]
    oc, dc = o.reshape(-1, 3)[pick.to(DEV)].float().cpu(), d.reshape(-1, 3)[pick.to(DEV)].float().cpu()
    cfg = dict(num_early_layers=8, num_filters=256)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    fn = lambda p: orc.cppn_forward(p, cfg, params)
    zc = z.cpu()
    raw_c = orc.get_predictions(fn, orc.points_dense(oc, dc, zc).reshape(-1, 3), 65536).reshape(-1, SC, 1)
    _, _, w_c, _, _ = orc.render_volume_density(raw_c, dc, zc)
    zf_c = orc.fine_depths(zc, w_c, u.cpu()[pick], pick.numel())
    raw_f = orc.get_predictions(fn, orc.points_dense(oc, dc, zf_c).reshape(-1, 3), 65536).reshape(-1, SC + NF, 1)
    rgb_c, dep_c, wf_c, _, _ = orc.render_volume_density(raw_f, dc, zf_c)
    pk = pick.to(DEV)
    assert rel_l2(res["f32"][0][pk].cpu().numpy(), w_c.numpy()) < 1e-4
    same = (res["f32"][1][pk].cpu() - zf_c).abs().max(-1).values < 1e-2                   # rays whose inverse-CDF bins did not flip
    assert float(same.float().mean()) > 0.98
    assert rel_l2(res["f32"][3][pk].cpu()[same].numpy(), dep_c[same].numpy()) < 1e-4
    assert rel_l2(res["f32"][2][pk].cpu()[same].numpy(), rgb_c[same].numpy()) < 1e-3


def test_c5_full_size_and_graph_capture():
    """BASELINE config C5: 1024x1024 rays x 256 samples/ray, 8x256, hipGraph-captured train step.  268 M ray-samples:
    f16 pixels against the fp32 mode on every pixel, and a captured fused train step (17 ray chunks at a 128 GiB
    workspace) replayed bit-identically."""
    from nerf_for_angiography_amd.render import render_projection, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W, S = 1024, 256
    m = _bench_model("f16")
    _, _, m44, _, _ = get_ray_values(75.0, 5.0, 0.0, np.array([0, 0, 1500.0]), 8, 8, 13.0 * 8, DEV)
    poses = torch.from_numpy(m44[None]).to(DEV)
    with torch.no_grad():
        a = render_projection(m, poses, W, W, 13.0 * W, S, 1400.0, 1600.0).rgb_map
        m.precision = "f32"
        c = render_projection(m, poses, W, W, 13.0 * W, S, 1400.0, 1600.0).rgb_map
    assert a.shape == (W * W,) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert rel_l2(a.cpu().numpy(), c.cpu().numpy()) < 1e-4
    assert float((a - c).abs().max()) < 2e-4
    m.precision = "f16"
    eng = m.engine
    eng.max_workspace_bytes = 128 << 30
    spec = projection_spec(poses, W, W, 13.0 * W, S, 1400.0, 1600.0)
    tgt = torch.rand(W * W, device=DEV)
    prepared = m._prepared()
    grad_eager = torch.zeros(eng.param_count, device=DEV)
    pix_eager = eng.train_step_mse(prepared, spec, tgt, 1.0 / (W * W), grad_eager, "f16")      # also sizes the workspace
    torch.cuda.synchronize()
    assert float((pix_eager - a).abs().max()) < 1e-6
    grad_g = torch.zeros(eng.param_count, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            grad_g.zero_()
            pix_g = eng.train_step_mse(prepared, spec, tgt, 1.0 / (W * W), grad_g, "f16")
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(pix_g, pix_eager)
    assert torch.equal(grad_g, grad_eager)


def test_hip_graph_capture_of_the_render_and_train_step(golden):
    """The C-ABI allocates nothing and never synchronises: a fused forward and a fused train step can be
    captured into a HIP graph and replayed (BASELINE config 5 asks for a hipGraph-captured render step)."""
    from nerf_for_angiography_amd.engine import RenderSpec
    g, m, near, far, s = _c1(golden, "bf16")
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    spec = RenderSpec(n_rays=o.shape[0], n_samples=s, origins=o, dirs=d, mode="acc", t_near=near, t_far=far)
    eng = m.engine
    prepared = m._prepared()
    grad_eager = torch.zeros(eng.param_count, device=DEV)
    pix_eager = eng.train_step_mse(prepared, spec, tgt, 1.0 / o.shape[0], grad_eager, "bf16")      # also sizes the workspace
    fwd_eager, _, _ = eng.render_forward(prepared, spec, "bf16")
    torch.cuda.synchronize()
    grad_g = torch.zeros(eng.param_count, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            grad_g.zero_()
            pix_g = eng.train_step_mse(prepared, spec, tgt, 1.0 / o.shape[0], grad_g, "bf16")
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(pix_g, pix_eager)
    assert torch.equal(grad_g, grad_eager)
    assert rel_l2(fwd_eager.cpu().numpy(), pix_eager.cpu().numpy()) < 1e-6


def test_training_driver_reference_loop_with_grid(tmp_path):
    """--march grid: the reference's own iteration body (run_nerf_acc.py:284-306) on the HIP grid / march kernels, with the
    device ray sampler: loss goes down and the march skips most of the empty volume."""
    from nerf_for_angiography_amd.nerf.run_nerf_acc import main
    out = main(["--synthetic", "--img_size", "20", "--number_angles", "1", "--limited_size", "90", "--n_iters", "160",
                "--display_every", "80", "--sample_size", "16", "--depth_samples", "100", "--num_layers", "4",
                "--num_hidden_units", "64", "--sampling_strategy", "segmentation", "--march", "grid", "--precision", "f16",
                "--log_dir", str(tmp_path / "run")])
    h = out["history"]
    assert [r["iter"] for r in h] == [0, 80, 160]
    assert h[-1]["train_loss"] < h[0]["train_loss"] and all(np.isfinite(r["test_psnr"]) for r in h)
    assert 0 < h[-1]["marched_samples_per_iter"] < 256 * 100


def test_training_precision_tracks_fp32_convergence():
    """before a reduced precision is the training default, show it converges like fp32.  The same model, rays,
    targets and Adam settings are trained 150 iterations at f32 (render + autograd), f16 and f16s8 (fused train step): the
    loss curves agree to a fraction of a percent and the final PSNRs to 0.05 dB."""
    from nerf_for_angiography_amd.render import render_projection, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values, capsule_tree, capsule_mu, ray_tracing_fn
    W, S, near, far = 64, 64, 1400.0, 1600.0
    caps = capsule_tree(levels=4, seed=1)
    poses, targets = [], []
    z_gt = torch.linspace(near, far, 128, device=DEV)
    for th in (0.0, 45.0, 90.0, 135.0):
        o, d, m44, _, _ = get_ray_values(th, 0.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, DEV)
        poses.append(torch.from_numpy(m44[None]).to(DEV))
        targets.append(ray_tracing_fn(lambda p: capsule_mu(p, caps), o.reshape(-1, 3).float(), d.reshape(-1, 3).float(), z_gt).reshape(-1))
    curves = {}
    for prec in ("f32", "f16", "f16s8"):
        torch.manual_seed(21)
        m = make_model(4, 128, precision=prec)
        with torch.no_grad():
            m.output_linear[0].bias.fill_(-5.0)
        opt = torch.optim.Adam(list(m.parameters()), lr=5e-4)
        losses = []
        for it in range(150):
            i = it % 4
            opt.zero_grad()
            if prec == "f32":
                loss = torch.nn.functional.mse_loss(render_projection(m, poses[i], W, W, 13.0 * W, S, near, far).rgb_map, targets[i])
                loss.backward()
            else:
                loss, _ = train_step_mse(m, projection_spec(poses[i], W, W, 13.0 * W, S, near, far), targets[i])
            opt.step()
            losses.append(float(loss.detach()))
        curves[prec] = np.array(losses)
    assert curves["f32"][-8:].mean() < 0.5 * curves["f32"][:8].mean()            # it does train
    for prec in ("f16", "f16s8"):
        tail = slice(-20, None)
        rel = abs(curves[prec][tail].mean() - curves["f32"][tail].mean()) / curves["f32"][tail].mean()
        dpsnr = abs(10 * np.log10(curves[prec][tail].mean() / curves["f32"][tail].mean()))
        assert rel < 2e-2 and dpsnr < 0.1, (prec, rel, dpsnr)


def test_training_driver_runs_and_checkpoints(tmp_path):
    """Mirror of nerf/run_nerf_acc.py on a tiny synthetic dataset: loss goes down, the best checkpoint has the
    reference's dictionary layout and reloads into a fresh CPPN; 300 samples/ray (the reference's setting,
    not a divisor of the tile) exercises the two-launch train step."""
    from nerf_for_angiography_amd.nerf.run_nerf_acc import main
    from nerf_for_angiography_amd.model.CPPN import CPPN
    out = main(["--synthetic", "--img_size", "20", "--number_angles", "1", "--limited_size", "90", "--n_iters", "120",
                "--display_every", "60", "--sample_size", "16", "--depth_samples", "300", "--num_layers", "4",
                "--num_hidden_units", "64", "--sampling_strategy", "segmentation", "--log_dir", str(tmp_path / "run")])
    h = out["history"]
    assert [r["iter"] for r in h] == [0, 60, 120]
    assert h[-1]["train_loss"] < h[0]["train_loss"]
    assert all(np.isfinite(r["test_psnr"]) for r in h)
    ck = torch.load(str(tmp_path / "run" / "coarsemodel.pth"), weights_only=False)
    assert set(ck) == {"version", "parameters", "training_information", "model"}
    m2 = CPPN(ck["parameters"]).to(DEV)
    m2.load_state_dict(ck["model"])
    x = torch.randn(64, 3, device=DEV) * 50
    if ck["training_information"]["epochs"] == 120:
        with torch.no_grad():
            assert torch.equal(m2(x), out["model"](x))
    assert len((tmp_path / "run" / "train_log.jsonl").read_text().splitlines()) == 3


def test_training_driver_trains_fourier_coefficients(tmp_path):
    """--pos_enc fourier: the coefficients are an optimiser parameter as upstream (model/CPPN.py:92) and move during training."""
    from nerf_for_angiography_amd.nerf.run_nerf_acc import main
    out = main(["--synthetic", "--img_size", "20", "--number_angles", "1", "--limited_size", "90", "--n_iters", "40",
                "--display_every", "40", "--sample_size", "16", "--depth_samples", "64", "--num_layers", "4",
                "--num_hidden_units", "64", "--pos_enc", "fourier", "--log_dir", str(tmp_path / "run")])
    m = out["model"]
    assert m.fourier_coefficients.requires_grad
    ck = torch.load(str(tmp_path / "run" / "coarsemodel.pth"), weights_only=False)
    assert "fourier_coefficients" in ck["model"]
    assert any(p is m.fourier_coefficients for g in out["optimizer"].param_groups for p in g["params"])
    state = out["optimizer"].state[m.fourier_coefficients]
    assert int(state["step"]) >= 40 and float(state["exp_avg_sq"].sum()) > 0      # Adam has seen non-zero gradients
    assert all(np.isfinite(r["train_loss"]) for r in out["history"])


@pytest.mark.parametrize("type_ct", [True, False])
def test_voxel_projector_vs_scipy(type_ct):
    """afx_project_volume vs the reference's own projector pieces (scipy RegularGridInterpolator + exp/prod, float64)
    on a 41^3 voxelised capsule phantom, 64x48 image; rays leave the volume (fill value) on both ends."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.phantomdata.helpers import VoxelVolume, ray_tracing, get_ray_values, get_depth_values, capsule_tree, capsule_mu
    t = np.linspace(-80.0, 80.0, 41)
    gx, gy, gz = np.meshgrid(t, t, t, indexing="ij")
    vals = np.zeros_like(gx)
    for c, sg, amp in (((10.0, -5.0, 0.0), 18.0, 0.03), ((-30.0, 25.0, 20.0), 9.0, 0.08), ((35.0, 30.0, -25.0), 6.0, 0.15)):
        vals += amp * np.exp(-((gx - c[0]) ** 2 + (gy - c[1]) ** 2 + (gz - c[2]) ** 2) / (2 * sg ** 2))
    vals = (vals + 0.0005).astype(np.float32)
    vals[:2] = 0.0          # fill value = min = 0 outside, but non-zero background inside
    w, h = 64, 48
    o, d, _, ii, jj = get_ray_values(25.0, -15.0, 3.0, np.array([0, 0, 1500.0]), w, h, 13.0 * w, "cpu")
    z = get_depth_values(1380.0, 1620.0, 97, "cpu", stratified=False)
    if not type_ct:
        vals = vals * 0.1
    want = orc.project_volume_scipy((t, t, t), vals, o.reshape(-1, 3), d.reshape(-1, 3), z.double(), type_ct).reshape(h, w)
    vol = VoxelVolume(t, t, t, vals, device=DEV)
    got = ray_tracing(vol, [25.0, -15.0, 3.0], o, d, z, w, h, ii, jj, 100, DEV, None, type="ct" if type_ct else "other")
    assert got.shape == (h, w)
    assert 0.01 < float(want.std()) and float(want.min()) < 0.95
    assert rel_l2(got.cpu().numpy(), want.numpy()) < 1e-4
    # in-kernel ray generation from the pose gives the same image as ray arrays
    from nerf_for_angiography_amd.engine import project_volume
    from nerf_for_angiography_amd.phantomdata.proj_helpers import source_matrix
    pose = torch.from_numpy(source_matrix(np.array([0, 0, 1500.0]), 25.0, -15.0, 3.0)[None]).to(DEV)
    img2 = project_volume(vol.values, vol.origin, vol.spacing, vol.fill_value, z.to(DEV), poses=pose, width=w, height=h,
                          focal=13.0 * w, type_ct=type_ct).reshape(h, w)
    assert rel_l2(img2.cpu().numpy(), want.numpy()) < 2e-6          # float64 rays, as upstream
    with pytest.raises(ValueError):
        VoxelVolume(np.array([0.0, 1.0, 3.0]), t, t, np.zeros((3, 41, 41)), device=DEV)


def test_voxel_projector_vs_reference_fixture(golden):
    """afx_project_volume against the REFERENCE's ray_tracing output (fixture G10, captured with a frangi stand-in for the
    one import this container lacks): 41^3 voxelised sphere + capsule, 64 x 48 detector, two poses, both branches."""
    from nerf_for_angiography_amd.phantomdata.helpers import VoxelVolume, ray_tracing
    g = golden("g10_ray_tracing")
    w, h, f = (int(g["whf"][0]), int(g["whf"][1]), float(g["whf"][2]))
    vol = VoxelVolume(g["axis"], g["axis"], g["axis"], g["mu"], fill_value=float(g["fill"]), device=DEV)
    for tag in ("a", "b"):
        o, d, z = T(g[f"{tag}_o"]), T(g[f"{tag}_d"]), T(g[f"{tag}_z"])
        for kind in ("ct", "sdf"):
            got = ray_tracing(vol, list(g[f"{tag}_angles"]), o, d, z, w, h, None, None, 32, DEV, None, type=kind)
            assert rel_l2(got.cpu().numpy(), g[f"{tag}_img_{kind}"]) < 1e-5, (tag, kind)      # fp32 ray arrays into the kernel


def test_evaluation_sweep(golden):
    """visualization.py:277-454 as two launches: every view of the angle sweep rendered in one fused call and compared with
    the ground-truth projector's output over the same poses; PSNR / DOT 2D / DICE 2D per view against a per-view oracle."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.visualization.sweep import sweep_angles, evaluation_sweep, ground_truth_sweep
    from nerf_for_angiography_amd.phantomdata.helpers import VoxelVolume
    assert sweep_angles(180, 5).shape == (37 * 37, 2)
    g = golden("g10_ray_tracing")
    vol = VoxelVolume(g["axis"], g["axis"], g["axis"], g["mu"], fill_value=float(g["fill"]), device=DEV)
    angles = sweep_angles(40, 20)                       # 3 x 3 views
    w, h, s = 24, 20, 48
    near, far, src = 1400.0, 1600.0, np.array([0, 0, 1500.0])
    z = torch.linspace(near, far, 96)
    gt = ground_truth_sweep(vol, angles, w, h, 13.0 * w, src, z)
    torch.manual_seed(8)
    m = make_model(4, 64)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(8.0)
        m.output_linear[0].bias.fill_(-5.0)
    df, preds = evaluation_sweep(m, gt, angles, w, h, 13.0 * w, src, near, far, s, binary_targets=(gt > 0.9).float())
    assert list(df.columns[:9]) == ["image_id", "theta", "phi", "larm", "theta_360", "phi_360", "cam_pose_x", "cam_pose_y", "cam_pose_z"]
    assert len(df) == 9 and df["theta_360"].min() >= 0 and preds.shape == (9, h, w)
    cfg = dict(num_early_layers=4, num_filters=64)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for i, (th, ph) in enumerate(angles):
        pose = orc.source_matrix(src, th if th >= 0 else 360 + th, ph if ph >= 0 else 360 + ph)
        o, d = orc.get_rays(pose, w, h, 13.0 * w)
        want_gt = orc.project_volume_scipy((g["axis"],) * 3, g["mu"], o.reshape(-1, 3), d.reshape(-1, 3), z.double(), True, float(g["fill"]))
        assert rel_l2(gt[i].cpu().numpy().ravel(), want_gt.numpy()) < 1e-5
        pix = orc.render_rays(o.reshape(-1, 3).float(), d.reshape(-1, 3).float(), cfg, params, near=near, far=far, n_samples=s, convention="acc")
        assert rel_l2(preds[i].cpu().numpy().ravel(), pix.numpy()) < 1e-5
        psnr = float(-10 * torch.log10(torch.mean((pix - want_gt) ** 2)))
        assert abs(df["PSNR"][i] - psnr) < 1e-3
    assert 0.0 <= df["DICE 2D"].min() <= df["DICE 2D"].max() <= 1.0 and df["DOT 2D"].between(0, 1).all()


def test_two_stream_overlap_mode_matches_serial(monkeypatch):
    """AFX_OVERLAP=1 (weight-gradient kernels of chunk i on a side stream while the chain kernel of chunk i+1 runs,
    double-buffered stash, non-persistent chain grid) must give the same gradients as the serial schedule."""
    from nerf_for_angiography_amd.render import train_step_mse
    from nerf_for_angiography_amd.engine import RenderSpec
    torch.manual_seed(17)
    r, s = 3000, 64
    o = torch.randn(r, 3, device=DEV) * 3 + torch.tensor([0, 0, 1500.0], device=DEV)
    d = torch.nn.functional.normalize(torch.randn(r, 3, device=DEV) * 0.03 + torch.tensor([0, 0, -1.0], device=DEV), dim=-1)
    tgt = torch.rand(r, device=DEV)
    spec = RenderSpec(n_rays=r, n_samples=s, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("AFX_OVERLAP", mode)
        torch.manual_seed(3)
        m = make_model(4, 64, precision="bf16")           # the context reads AFX_OVERLAP when it is created
        with torch.no_grad():
            m.output_linear[0].bias.fill_(-5.0)
        lib = m.engine.lib
        # a workspace for ~1/3 of the tiles forces several chunks (and with overlap, two half-size stash buffers)
        fixed = int(lib.afx_query(m.engine.h, 4, 0, 0, 2)) - 32 * 256 * (2 * 5 * 64 * 2 + 64 + 4)
        m.engine.max_workspace_bytes = fixed + 260 * 256 * (2 * 5 * 64 * 2 + 64 + 4)
        loss, pix = train_step_mse(m, spec, tgt)
        torch.cuda.synchronize()
        results[mode] = (float(loss), pix.clone(), _grads_by_name(m))
    assert results["0"][0] == results["1"][0] and torch.equal(results["0"][1], results["1"][1])
    for k, v in results["0"][2].items():
        assert rel_l2(results["1"][2][k], v) < 1e-5, k


def test_reference_loop_with_occupancy_grid():
    """The training-iteration body of run_nerf_acc.py:284-306 with the occupancy grid enabled, through the mirrored
    call surface: grid update, grid-skipping march with early termination (restated nerfacc; unpinned), fused MLP on
    the packed positions, packed Beer-Lambert product, backward.  The packed result is checked against the CPU oracle
    on the very same packed samples."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.nerf.nerf_helpers import get_predictions
    from nerf_for_angiography_amd.nerf.nerf_helpers_acc import acc_ray_marching, acc_render_volume_density, acc_update_n_step
    from nerf_for_angiography_amd.nerf.occupancy import OccupancyGrid, ContractionType
    torch.manual_seed(23)
    m = make_model(4, 64)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(8.0)
        m.output_linear[0].bias.fill_(-4.0)
    outside = 100.0
    scene_aabb = torch.tensor([-outside] * 3 + [outside] * 3, dtype=torch.float32, device=DEV)
    grid = OccupancyGrid(roi_aabb=scene_aabb, resolution=32, contraction_type=ContractionType.AABB).to(DEV)
    r, s = 500, 100
    o = torch.tensor([[0.0, 0.0, 1500.0]], device=DEV).repeat(r, 1)
    d = torch.nn.functional.normalize(torch.randn(r, 3, device=DEV) * 0.03 + torch.tensor([0, 0, -1.0], device=DEV), dim=-1)
    tgt = torch.rand(r, device=DEV)
    with torch.no_grad():
        for n_iter in (0, 16):
            grid = acc_update_n_step(grid, m, n_iter, occ_thre=1e-4)
        ri, ts, te = acc_ray_marching(m, grid, scene_aabb, o, d, s, 1400.0, 1600.0, 1e-2, 1e-4)
    assert 0 < ri.numel() < r * s and bool(torch.all(ri[1:] >= ri[:-1]))
    pos = o[ri.long()] + d[ri.long()] * (ts + te) / 2.0
    pred = get_predictions(m, pos, 131072)
    pix, _ = acc_render_volume_density(pred, ri, ts, te, r, s)
    loss = torch.nn.functional.mse_loss(pix, tgt)
    loss.backward()
    cfg = dict(num_early_layers=4, num_filters=64)
    params = {k: v.detach().cpu().clone().requires_grad_(k.startswith(("early", "output"))) for k, v in m.state_dict().items()}
    pred_c = orc.cppn_forward(pos.cpu(), cfg, params)
    pix_c = orc.acc_render_volume_density(pred_c, ri.cpu(), ts.cpu(), te.cpu(), r)
    torch.nn.functional.mse_loss(pix_c, tgt.cpu()).backward()
    assert rel_l2(pix.detach().cpu().numpy(), pix_c.detach().numpy()) < 1e-5
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert rel_l2(p.grad.cpu().numpy(), params[k].grad.numpy()) < 1e-4, k


def test_grid_kernels_vs_oracle():
    """afx_grid_points / afx_grid_update / afx_grid_binarize / afx_march_count+write / afx_march_visibility+compact against
    the oracle's restatement of nerfacc 0.3.x: indices and cell decisions bit-exact, floats bit-exact where the same
    single-precision operations are applied (jitter, alphas supplied: parity mode)."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd import engine as eng

    def same(a, b, what, scale=None):      # floats: identical up to one unit in the last place of the operands' magnitude
        a, b = a.cpu(), b.cpu()
        assert a.shape == b.shape, what
        diff = (a.double() - b.double()).abs()
        tol = 1.2e-7 * (float(b.abs().max()) if scale is None else scale)
        assert bool((diff <= tol).all()), (what, int((diff > tol).sum()), float(diff.max()))

    g = torch.Generator().manual_seed(5)
    aabb = torch.tensor([-100.0, -80, -100, 100, 120, 90])
    res = (16, 12, 20)
    nc = res[0] * res[1] * res[2]
    # --- points of jittered cells, all cells and an index list with duplicates
    jit = torch.rand(nc, 3, generator=g)
    pts = eng.grid_points(aabb.tolist(), res, None, nc, jitter=jit.to(DEV), device=torch.device(DEV))
    same(pts, orc.grid_jittered_points(torch.arange(nc), jit, aabb, res), "points", 220.0)
    cells = torch.randint(nc, (1500,), generator=g)
    cells[100:200] = cells[0:100]                                          # duplicates
    pts2 = eng.grid_points(aabb.tolist(), res, cells.to(DEV, torch.int32), cells.numel(), jitter=jit[:1500].to(DEV))
    same(pts2, orc.grid_jittered_points(cells, jit[:1500], aabb, res), "points (index list)", 220.0)
    # --- update rule + binarisation
    occs0 = torch.rand(nc, generator=g) * (torch.rand(nc, generator=g) < 0.3)
    occ_new = torch.rand(1500, generator=g) * 0.8
    occs_d = occs0.clone().to(DEV)
    eng.grid_update(aabb.tolist(), res, occs_d, cells.to(DEV, torch.int32), occ_new.to(DEV), 0.95, torch.empty(nc, device=DEV))
    want, want_bin = orc.grid_update(occs0, cells, occ_new, 0.95, 1e-2)
    same(occs_d, want, "occs")
    b8, bits = torch.zeros(nc, dtype=torch.uint8, device=DEV), torch.zeros((nc + 31) // 32, dtype=torch.int32, device=DEV)
    eng.grid_binarize(aabb.tolist(), res, occs_d, 1e-2, b8, bits, torch.zeros(256, dtype=torch.float64, device=DEV))
    assert torch.equal(b8.cpu().bool(), want_bin)
    unpacked = ((bits.cpu().long()[:, None] >> torch.arange(32)) & 1).reshape(-1)[:nc].bool()
    assert torch.equal(unpacked, want_bin)
    # --- march: rays through, beside and missing the box; with and without the grid; near/far clipping
    r = 300
    o = torch.tensor([[0.0, 10.0, 1500.0]]).repeat(r, 1) + torch.randn(r, 3, generator=g) * 5
    d = torch.nn.functional.normalize(torch.randn(r, 3, generator=g) * 0.08 + torch.tensor([0, 0, -1.0]), dim=-1)
    d[7] = torch.tensor([0.0, 0.0, -1.0])                                   # a zero direction component
    d[8] = torch.tensor([0.7, 0.7, -0.1])                                   # misses
    binary = want_bin.view(*res)
    for use_grid, near, far in ((False, 1400.0, 1600.0), (True, 1400.0, 1600.0), (True, None, 1700.0), (True, 1450.0, None)):
        ri, ts, te, mid, off = eng.march(o.to(DEV), d.to(DEV), aabb.tolist(), near, far, 2.5, grid_bits=bits if use_grid else None,
                                         grid_aabb=aabb.tolist(), grid_res=res)
        ri_c, ts_c, te_c = orc.march_grid(o, d, aabb, near, far, 2.5, binary if use_grid else None, aabb)
        assert torch.equal(ri.cpu().long(), ri_c), (use_grid, near, far)
        same(ts, ts_c, "t_starts"); same(te, te_c, "t_ends")
        assert torch.equal(off.cpu(), torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(ri_c, minlength=r).cumsum(0)]))
        pos_c = o[ri_c] + d[ri_c] * (ts_c + te_c)[:, None] / 2.0
        same(mid, pos_c, "mid-points", 1700.0)
    # --- render_visibility on supplied alphas (parity mode) and through the raw path
    alphas = torch.rand(ri_c.numel(), generator=g) * (torch.rand(ri_c.numel(), generator=g) < 0.7) * 0.6
    for eps, thre in ((1e-2, 1e-3), (0.3, 0.0), (1e-4, 0.2)):
        ri2, ts2, te2 = eng.march_visibility(alphas.to(DEV), ts, te, off, eps, thre, is_alpha=True)
        keep = orc.render_visibility(alphas, ri_c, eps, thre)
        assert torch.equal(ri2.cpu().long(), ri_c[keep]); same(ts2, ts_c[keep], "kept t_starts"); same(te2, te_c[keep], "kept t_ends")
    raw = torch.randn(ri_c.numel(), generator=g) * 3
    a_raw = 1 - torch.exp(-torch.sigmoid(raw) * (te_c - ts_c))
    ri3, _, _ = eng.march_visibility(raw.to(DEV), ts, te, off, 1e-2, 1e-3)
    keep3 = orc.render_visibility(a_raw, ri_c, 1e-2, 1e-3)
    assert (ri3.cpu().long().numel() - int(keep3.sum())) in (-1, 0, 1)      # expf vs torch.exp at a threshold: at most one flip
    # empty input
    ri0, ts0, te0, _, off0 = eng.march(o[:0].to(DEV), d[:0].to(DEV), aabb.tolist(), 1400.0, 1600.0, 2.5)
    assert ri0.numel() == 0 and off0.tolist() == [0]


def test_device_ray_sampler_and_philox():
    """R13 on the device: a resident ray table, weighted sampling without replacement (Efraimidis-Spirakis keys + top-k) and
    the gather - no pandas, no host round trip per iteration; the Philox stream behind every perf-mode draw."""
    from nerf_for_angiography_amd import engine as eng
    dev = torch.device(DEV)
    u = eng.philox_uniform(7, 3, 100000, dev)
    assert torch.equal(u, eng.philox_uniform(7, 3, 100000, dev)) and not torch.equal(u[:1000], eng.philox_uniform(7, 4, 1000, dev))
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    assert torch.equal(u[:10], eng.philox_uniform(7, 3, 10, dev))            # counter-based: a prefix is a prefix
    n, k = 20000, 512
    g = torch.Generator().manual_seed(2)
    o, d, pix = torch.randn(n, 3, generator=g).to(DEV), torch.randn(n, 3, generator=g).to(DEV), torch.rand(n, generator=g).to(DEV)
    w = torch.rand(n, generator=g)
    w[:n // 2] *= 0.02                                                      # first half nearly never drawn
    w[5] = 0.0
    uu = torch.rand(n, generator=g).clamp(min=1e-7)
    so, sd, sp, idx = eng.sample_rays(o, d, pix, w.to(DEV), k, u=uu.to(DEV))
    keys = torch.where(w > 0, torch.log(uu) / w, torch.full_like(w, -float("inf")))
    assert set(idx.cpu().tolist()) == set(torch.topk(keys, k).indices.tolist())
    assert idx.unique().numel() == k and 5 not in idx.cpu().tolist()        # without replacement; zero weight never drawn
    assert torch.equal(so, o[idx]) and torch.equal(sd, d[idx]) and torch.equal(sp, pix[idx])
    frac_low = []
    for step in range(20):                                                  # perf mode: Philox keys, one stream per step
        _, _, _, idx = eng.sample_rays(o, d, pix, w.to(DEV), k, seed=11, stream_id=step)
        frac_low.append(float((idx < n // 2).float().mean()))
    assert 0.005 < float(np.mean(frac_low)) < 0.05                          # expected share of the down-weighted half ~ 2 %


def test_topk_radix_select():
    """afx_topk_indices (selection step of the ray sampler): bit-exact set equality with a sort, ascending output, ties by lowest
    index, -inf keys, k = n, tiny and ragged sizes, run-to-run determinism."""
    from nerf_for_angiography_amd import engine as eng
    g = torch.Generator().manual_seed(5)
    for n, k in ((1, 1), (7, 3), (1023, 1000), (1024, 1), (1025, 513), (300001, 5625), (900000, 5625), (4096, 4096)):
        keys = torch.log(torch.rand(n, generator=g).clamp(min=1e-7)) / (torch.rand(n, generator=g) + 0.05)
        if n > 100:
            keys[::97] = -float("inf")
        kd = keys.to(DEV)
        idx = eng.topk_indices(kd, k)
        assert idx.dtype == torch.int64 and idx.shape == (k,)
        ref = torch.sort(keys, descending=True, stable=True).indices[:k]
        assert torch.equal(idx.cpu(), torch.sort(ref).values), (n, k)
        assert torch.equal(idx, eng.topk_indices(kd, k))
    # ties: 5 distinct values over 10 000 keys; the k-th largest value is cut in index order
    keys = torch.randint(0, 5, (10000,), generator=g).float()
    for k in (1, 1999, 2000, 2001, 7777, 10000):
        idx = eng.topk_indices(keys.to(DEV), k).cpu()
        ref = torch.sort(torch.sort(keys, descending=True, stable=True).indices[:k]).values
        assert torch.equal(idx, ref), k
    # mixed signs, zeros and denormals order like floats
    keys = torch.tensor([0.0, -0.0, 1e-45, -1e-45, 3.5, -3.5, float("inf"), -float("inf"), 1.0, -1.0])
    idx = eng.topk_indices(keys.to(DEV), 4).cpu().tolist()
    assert idx == [2, 4, 6, 8]
    assert eng.topk_indices(keys.to(DEV), 0).numel() == 0
    with pytest.raises(ValueError):
        eng.topk_indices(keys.to(DEV), 11)


def test_stratified_depths_fused():
    """R4: (a) parity mode - the reference-captured stratified depths of fixture G3 fed through the fused kernel (dense
    convention, shared z) against the oracle; (b) perf mode - randomize_depth drawn IN the kernel from Philox equals the
    host formula on the same uniform numbers."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd import engine as eng
    from nerf_for_angiography_amd.render import render_rays, render_spec
    from nerf_for_angiography_amd.engine import RenderSpec
    from nerf_for_angiography_amd.phantomdata.proj_helpers import source_matrix
    g3 = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "g3_stratify.npz")))
    torch.manual_seed(4)
    m = make_model(4, 128)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-26.0)
    pose = source_matrix(np.array([0, 0, 1500.0]), 60.0, 15.0)
    o_all, d_all = orc.get_rays(pose, 48, 48, 13.0 * 48)
    pick = torch.randperm(48 * 48)[:700]
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    cfg = dict(num_early_layers=4, num_filters=128)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for s in (32, 128):
        z = torch.from_numpy(g3[f"out{s}"]) * 200.0 + 1400.0               # the reference's jittered depths, scaled to the scene
        assert torch.equal(orc.stratify(torch.from_numpy(g3[f"z{s}"]), torch.from_numpy(g3[f"u{s}"])), torch.from_numpy(g3[f"out{s}"]))
        pix_c = orc.render_rays(o, d, cfg, params, near=0.0, far=0.0, n_samples=s, z=z, convention="dense")
        with torch.no_grad():
            pix = render_rays(m, o.to(DEV), d.to(DEV), mode="dense", z=z.to(DEV)).rgb_map
        assert float(pix_c.max()) > 1e-3 and rel_l2(pix.cpu().numpy(), pix_c.numpy()) < 1e-5, s
    # perf mode
    s, near, far = 64, 1400.0, 1600.0
    u = eng.philox_uniform(99, 5, s, torch.device(DEV)).cpu()
    t = torch.arange(s, dtype=torch.float32) / (s - 1)
    z_lin = near * (1.0 - t) + far * t
    z_host = orc.stratify(z_lin, u)
    spec = RenderSpec(n_rays=o.shape[0], n_samples=s, origins=o.to(DEV), dirs=d.to(DEV), mode="stratified", t_near=near, t_far=far,
                      jitter_seed=99, jitter_stream=5)
    with torch.no_grad():
        a = render_spec(spec, m).rgb_map
        b = render_rays(m, o.to(DEV), d.to(DEV), mode="dense", z=z_host.to(DEV)).rgb_map
        spec.jitter_stream = 6
        c = render_spec(spec, m).rgb_map
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    assert rel_l2(c.cpu().numpy(), b.cpu().numpy()) > 1e-6                  # another stream, another jitter


@pytest.mark.parametrize("layers,width,n_samples,n_rays", [(1, 64, 2, 1), (2, 128, 33, 7), (12, 64, 32, 40), (3, 256, 5, 3),
                                                              (5, 128, 96, 257)])
@pytest.mark.parametrize("prec", ["f32", "bf16", "f16", "f16s8"])
def test_odd_shapes_vs_oracle(layers, width, n_samples, n_rays, prec):
    """Edge geometry: a single ray, 2 samples, sample counts that are not multiples of 32, 1 and 12 hidden layers, ray
    counts that leave most of the last workgroup tile empty."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays
    torch.manual_seed(layers * 1000 + width + n_samples)
    m = make_model(layers, width, precision=prec)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-3.0)
    o = torch.tensor([[0.0, 0.0, 1500.0]]).repeat(n_rays, 1) + torch.randn(n_rays, 3)
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3) * 0.03 + torch.tensor([0, 0, -1.0]), dim=-1)
    tgt = torch.rand(n_rays)
    cfg = dict(num_early_layers=layers, num_filters=width)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pix_c, _, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=1450.0, far=1550.0, n_samples=n_samples, convention="acc")
    out = render_rays(m, o.to(DEV), d.to(DEV), n_samples, 1450.0, 1550.0, mode="acc")
    torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
    assert out.rgb_map.shape == (n_rays,)
    # few, thick steps (dt up to 50) give the 16-bit modes no averaging over samples: their error bars are 4x wider here;
    # the bf8 stash rounds each operand of the weight-gradient sum to 2 significant bits (12 % per term, zero-mean):
    # with 2 ... 25 000 terms per sum instead of millions its bar is 0.25 - this case checks geometry, not precision
    loose = 1 if prec == "f32" else 4
    # (thick steps: optical depths of tens, so a pixel's RELATIVE error is its optical depth's ABSOLUTE error: the f16 modes'
    # 1e-4 of an optical depth of 40 is 4e-3 on the pixel; their bar for well-conditioned renders is asserted elsewhere)
    ptol = 1e-2 if prec in ("f16", "f16s8") else loose * TOL[prec]["pix"]
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()) < ptol
    got = _grads_by_name(m)
    gtol = 0.25 if prec == "f16s8" else 2 * loose * TOL[prec]["grad"]
    for k, v in grads_c.items():
        if float(v.abs().max()) > 0:
            assert rel_l2(got[k], v.numpy()) < gtol, k


def test_model_too_deep_for_lds_is_refused():
    """16 hidden layers of width 256 need more ReLU-mask LDS than a CU has: a clear error, not a crash."""
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd._lib import AfxError
    m = make_model(16, 256, precision="bf16")
    o = torch.zeros(8, 3, device=DEV)
    with torch.no_grad():
        assert render_rays(m, o, o + 1, 32, 0.0, 1.0).rgb_map.shape == (8,)          # forward needs no masks
    out = render_rays(m, o, o + 1, 32, 0.0, 1.0)
    with pytest.raises(AfxError, match="LDS"):
        out.rgb_map.sum().backward()


@pytest.mark.parametrize("case", ["acc_arrays", "acc_pose_fused", "dense_shared_z", "dense_per_ray_z"])
def test_in_kernel_small_gradients_match_stashed_path(monkeypatch, case):
    """bf16 backward in rays mode: the first-/output-layer gradients come from per-group sums formed inside the chain
    kernel (MFMA transposes of H_N / dZ_0, inputs affine in the ray parameter: DESIGN.md 3) instead of from stashed
    H_N, dZ_0 and encoded inputs.  AFX_SMALL_IN_KERNEL=0 selects the stashed path: every gradient must agree (both
    contract the same bf16 operands in fp32; only the summation order and the x = c + (t - t0) d form differ),
    for every depth mode and ray source, including ragged sample counts (padded groups) and several chunks."""
    from nerf_for_angiography_amd.render import render_rays, render_projection, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    torch.manual_seed(11)
    m = make_model(3, 128, precision="bf16")
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-4.0)
    r, s = 777, 75                      # 75 samples: s_pad = 96, the last group of every ray is partly padding
    o = torch.randn(r, 3, device=DEV) * 3 + torch.tensor([0, 0, 1500.0], device=DEV)
    d = torch.nn.functional.normalize(torch.randn(r, 3, device=DEV) * 0.03 + torch.tensor([0, 0, -1.0], device=DEV), dim=-1)
    tgt = torch.rand(r, device=DEV)
    w = 40
    _, _, m44, _, _ = get_ray_values(33.0, 5.0, 0.0, np.array([0, 0, 1500.0]), w, w, 13.0 * w, DEV)
    poses = torch.from_numpy(m44[None]).to(DEV)
    tgt_p = torch.rand(w * w, device=DEV)
    z1 = torch.linspace(1400.0, 1600.0, s, device=DEV)
    z2 = (z1[None, :] + torch.rand(r, s, device=DEV) * 2.0).sort(dim=-1).values.contiguous()

    def grads(flag):
        monkeypatch.setenv("AFX_SMALL_IN_KERNEL", flag)
        m.engine.max_workspace_bytes = 1 << 30
        m.engine._ws = None
        m.zero_grad()
        if case == "acc_arrays":
            torch.nn.functional.mse_loss(render_rays(m, o, d, s, 1400.0, 1600.0).rgb_map, tgt).backward()
        elif case == "acc_pose_fused":
            train_step_mse(m, projection_spec(poses, w, w, 13.0 * w, s, 1400.0, 1600.0), tgt_p)
        else:       # render_volume_density convention: the 1e10 tail (D3) needs a very negative raw to leave a signal
            with torch.no_grad():
                m.output_linear[0].bias.fill_(-25.0)
            pix = render_rays(m, o, d, mode="dense", z=z1 if case == "dense_shared_z" else z2).rgb_map
            torch.nn.functional.mse_loss(pix, tgt).backward()
        torch.cuda.synchronize()
        return _grads_by_name(m)

    on, off = grads("1"), grads("0")
    assert any(float(np.abs(v).max()) > 0 for v in off.values())
    for k in off:
        assert rel_l2(on[k], off[k]) < 2e-5, (k, rel_l2(on[k], off[k]))


def test_full_size_train_step_is_bit_reproducible():
    """512x512 x 128, 8x256, f16 fused train step, three runs: pixels and every gradient bit-identical.  The chain
    kernels overlap the weight LDS-DMA with in-flight stash stores through counted s_waitcnt vmcnt / lgkmcnt; a wait
    that is one short shows up as a handful of differing pixels per projection (it did once: DESIGN.md 3), and only at
    a size where every CU runs many tiles back to back."""
    from nerf_for_angiography_amd.render import train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W = 512
    m = _bench_model("f16")
    _, _, m44, _, _ = get_ray_values(40.0, 3.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, DEV)
    poses = torch.from_numpy(m44[None]).to(DEV)
    tgt = torch.rand(W * W, device=DEV)
    spec = projection_spec(poses, W, W, 13.0 * W, 128, 1400.0, 1600.0)
    runs = []
    for _ in range(3):
        m.zero_grad()
        loss, pix = train_step_mse(m, spec, tgt)
        torch.cuda.synchronize()
        runs.append((float(loss), pix.clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        assert torch.equal(r[1], runs[0][1]), int((r[1] != runs[0][1]).sum())
        for k, g in runs[0][2].items():
            assert torch.equal(r[2][k], g), k
