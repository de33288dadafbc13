"""Pin the CPU oracle against vectors captured from the reference itself
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import angio_oracle as orc
from conftest import rel_l2


def T(a):
    return torch.from_numpy(np.asarray(a))


def sd_of(g, prefix="sd__"):
    return {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}


def test_g1_pose(golden):
    g = golden("g1_pose")
    for a, m in zip(g["args"], g["mats"]):
        got = orc.source_matrix(g["src_pt"], a[0], a[1], a[2], a[3:6])
        np.testing.assert_allclose(got, m, rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g2_rays(golden, tag):
    g = golden("g2_rays")
    w, h, f = g[f"{tag}_whf"]
    o, d = orc.get_rays(g[f"{tag}_pose"], int(w), int(h), float(f))
    assert o.shape == (int(h), int(w), 3)
    assert np.array_equal(o.numpy(), g[f"{tag}_o"])          # bit-exact float64
    np.testing.assert_allclose(d.numpy(), g[f"{tag}_d"], rtol=0, atol=1e-15)
    z = orc.depth_values(float(f) + 100, float(f) + 300, 8)
    assert np.array_equal(z.numpy(), g[f"{tag}_z"])


@pytest.mark.parametrize("s", [32, 128])
def test_g3_stratify(golden, s):
    g = golden("g3_stratify")
    out = orc.stratify(T(g[f"z{s}"]), T(g[f"u{s}"]))
    assert np.array_equal(out.numpy(), g[f"out{s}"])


CASES = {
    "none_relu_4x64": dict(num_early_layers=4, num_filters=64),
    "none_tanh_4x64": dict(num_early_layers=4, num_filters=64, act_func="tanh"),
    "none_sine_4x64": dict(num_early_layers=4, num_filters=64, act_func="sine", sine_weights=15),
    "none_relu_4x64_late4": dict(num_early_layers=4, num_filters=64, num_late_layers=4),
    "fourier_relu_4x64": dict(num_early_layers=4, num_filters=64, pos_enc="fourier", pos_enc_basis=5),
    "none_relu_4x128": dict(num_early_layers=4, num_filters=128),
    "none_relu_8x256": dict(num_early_layers=8, num_filters=256),
}


@pytest.mark.parametrize("name", list(CASES))
def test_g4_cppn(golden, name):
    g = golden("g4_cppn_" + name)
    y = orc.cppn_forward(T(g["x"]), CASES[name], sd_of(g))
    assert rel_l2(y.numpy(), g["y"]) < 2e-6


@pytest.mark.parametrize("name,cfg,alphas", [
    ("barf_relu_4x64", dict(num_early_layers=4, num_filters=64, pos_enc="barf", pos_enc_basis=5),
     (0.0, 1.5, 2.5, 5.0)),
    ("barf_relu_2x256", dict(num_early_layers=2, num_filters=256, pos_enc="barf", pos_enc_basis=5), (2.5,)),
])
def test_g4_cppn_barf(golden, name, cfg, alphas):
    g = golden("g4_cppn_" + name)
    sd = sd_of(g)
    for a in alphas:
        w = orc.barf_weights(a, 5)
        assert np.array_equal(w.numpy(), g[f"w_alpha{a}"])      # D7 literal
        sd["barf_weights"] = w
        y = orc.cppn_forward(T(g["x"]), cfg, sd)
        assert rel_l2(y.numpy(), g[f"y_alpha{a}"]) < 2e-6


@pytest.mark.parametrize("rk", ["n", "m40", "m3", "tail", "c2", "c3"])
@pytest.mark.parametrize("zk", ["z1", "z2"])
def test_g5_render(golden, rk, zk):
    g = golden("g5_render")
    rgb, dep, w, ent, (sig, _) = orc.render_volume_density(T(g["raw_" + rk]), T(g["d"]), T(g[zk]))
    for name, got in (("rgb", rgb), ("depth", dep), ("weights", w), ("entropy", ent), ("sigma", sig)):
        np.testing.assert_allclose(got.numpy(), g[f"{rk}_{zk}_{name}"], rtol=2e-6, atol=1e-30, err_msg=name)
    if rk == "m3":      # the 1e10 tail makes the projection exactly 0
        assert float(rgb.abs().max()) == 0.0
    if rk == "m40":
        assert float(rgb.min()) > 0.999999


def test_g5_cumprod(golden):
    g = golden("g5_render")
    np.testing.assert_allclose(orc.cumprod_exclusive(T(g["cumprod_in"])).numpy(), g["cumprod_out"], rtol=1e-7)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g7_sample_pdf(golden, tag):
    g = golden("g7_sample_pdf")
    out = orc.sample_pdf(T(g[f"{tag}_bins"]), T(g[f"{tag}_w"]), T(g[f"{tag}_u"]))
    np.testing.assert_allclose(out.numpy(), g[f"{tag}_out"], rtol=0, atol=2e-4)
    assert rel_l2(out.numpy(), g[f"{tag}_out"]) < 1e-7


def _c1_cfg():
    return dict(num_early_layers=4, num_filters=64)


def test_g8_dense(golden):
    g = golden("g8_e2e_c1")
    near, far, s = g["near_far_s"]
    params = sd_of(g, "init__")
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    pix, loss, grads = orc.loss_and_grads(o, d, tgt, _c1_cfg(), params, near=float(near), far=float(far),
                                          n_samples=int(s), convention="dense", z=T(g["z"]))
    np.testing.assert_allclose(pix.numpy(), g["dense_rgb"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(float(loss), float(g["dense_loss"]), rtol=1e-6)
    for k, v in grads.items():
        assert rel_l2(v.numpy(), g["dense_grad__" + k]) < 1e-5, k


def test_g8_dense26(golden):
    """Output bias -26: the 1e10 tail term is neither 0 nor 1, grads non-trivial."""
    g = golden("g8_e2e_c1")
    near, far, s = g["near_far_s"]
    params = sd_of(g, "init__")
    params["output_linear.0.bias"] = torch.full((1,), -26.0)
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    pix, loss, grads = orc.loss_and_grads(o, d, tgt, _c1_cfg(), params, near=float(near), far=float(far),
                                          n_samples=int(s), convention="dense", z=T(g["z"]))
    assert rel_l2(pix.numpy(), g["dense26_rgb"]) < 1e-6
    np.testing.assert_allclose(float(loss), float(g["dense26_loss"]), rtol=1e-5)
    for k, v in grads.items():
        assert rel_l2(v.numpy(), g["dense26_grad__" + k]) < 1e-4, k


def test_g8_acc(golden):
    g = golden("g8_e2e_c1")
    near, far, s = g["near_far_s"]
    params = sd_of(g, "init__")
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    pix, loss, grads = orc.loss_and_grads(o, d, tgt, _c1_cfg(), params, near=float(near), far=float(far),
                                          n_samples=int(s), convention="acc")
    assert rel_l2(pix.numpy(), g["acc_rgb"]) < 1e-6
    np.testing.assert_allclose(float(loss), float(g["acc_loss"]), rtol=1e-6)
    assert 0.05 < float(pix.mean()) < 0.95          # a non-degenerate projection
    for k, v in grads.items():
        assert rel_l2(v.numpy(), g["acc_grad__" + k]) < 1e-5, k


def test_g8_adam_steps(golden):
    """10 Adam steps with the reference schedule reproduce the captured weights."""
    g = golden("g8_e2e_c1")
    near, far, s = g["near_far_s"]
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_of(g, "init__").items()}
    o, d, tgt = T(g["o"]), T(g["d"]), T(g["target"])
    opt = torch.optim.Adam([v for v in params.values() if v.requires_grad], lr=1e-4)
    for it in range(10):
        opt.zero_grad()
        pix = orc.render_rays(o, d, _c1_cfg(), params, near=float(near), far=float(far), n_samples=int(s),
                              convention="acc")
        torch.nn.functional.mse_loss(pix, tgt).backward()
        opt.step()
        for pg in opt.param_groups:
            pg["lr"] = orc.lr_at(it)
        if it in (0, 9):
            for k, v in params.items():
                if k.startswith("early") or k.startswith("output"):
                    assert rel_l2(v.detach().numpy(), g[f"acc_step{it + 1}__{k}"]) < 2e-6, (it, k)


def test_g9_density_grid(golden):
    g = golden("g9_density_grid")
    pts = orc.density_grid_points(100.0, 16)
    assert np.array_equal(pts.numpy(), g["points"])          # D9 layout
    sd = sd_of(g)
    grid = orc.density_grid(lambda p: orc.cppn_forward(p, _c1_cfg(), sd), 100.0, 16)
    assert rel_l2(grid.numpy(), g["sigma"]) < 1e-6


def test_g10_ray_tracing(golden):
    """R14: the reference's ray_tracing output on a voxelised analytic phantom (captured with a frangi stand-in for the one
    import the container lacks) vs the oracle's projector, both branches, two poses; and R2 once more through helpers.py."""
    g = golden("g10_ray_tracing")
    w, h, f = g["whf"]
    ax = g["axis"]
    for tag in ("a", "b"):
        th, ph, la = g[f"{tag}_angles"]
        pose = orc.source_matrix(np.array([0, 0, 1500.0]), th, ph, la)
        assert np.array_equal(pose, g[f"{tag}_pose"])
        o, d = orc.get_rays(pose, int(w), int(h), float(f))
        assert np.array_equal(o.float().numpy(), g[f"{tag}_o"]) and np.array_equal(d.float().numpy(), g[f"{tag}_d"])
        z = T(g[f"{tag}_z"])
        for kind, ct in (("ct", True), ("sdf", False)):
            img = orc.project_volume_scipy((ax, ax, ax), g["mu"], o.reshape(-1, 3), d.reshape(-1, 3), z, ct, fill_value=float(g["fill"]))
            # the fixture stores the volume in fp32, the reference interpolated the float64 original: 1e-6, not 1e-7
            assert rel_l2(img.reshape(int(h), int(w)).numpy(), g[f"{tag}_img_{kind}"]) < 2e-6, (tag, kind)


def test_g11_transfer_functions(golden):
    from nerf_for_angiography_amd.phantomdata import helpers as ph
    g = golden("g11_transfer")
    for binary, key in ((False, "tf"), (True, "tf_binary")):
        assert np.allclose(orc.transfer_func_ct(g["vals"], binary), g[key], rtol=0, atol=1e-15)
        assert np.allclose(ph.transfer_func_ct(g["vals"], binary=binary), g[key], rtol=0, atol=1e-12)
    assert np.array_equal(orc.rev_sigmoid(g["x"], 2.0), g["rev_sigmoid_c1_2"]) and np.array_equal(orc.rev_sigmoid(g["x"]), g["rev_sigmoid_default"])
    assert np.array_equal(ph.rev_sigmoid(g["x"], c1=2), g["rev_sigmoid_c1_2"])
