"""CPU-only tests: the C-ABI library loads and exports what include/afx.h declares, host-side logic
(argument validation, layouts, CPPN mirror, helper mirrors) and the 2-rank gloo data-parallel path.
No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_l2


def T(a):
    return torch.from_numpy(np.asarray(a))


def model_def(layers, width, pos_enc="none", act="relu", late=0, **kw):
    d = dict(num_early_layers=layers, num_late_layers=late, num_filters=width, num_input_channels=3,
             num_output_channels=1, num_input_channels_views=0, use_bias=True, pos_enc=pos_enc, pos_enc_basis=5,
             act_func=act, fourier_sigma=5, num_img=1, device=torch.device("cpu"))
    if act == "sine":
        d["sine_weights"] = 15
    d.update(kw)
    return d


# ---------------------------------------------------------------- C-ABI surface
def header_functions():
    txt = open(os.path.join(ROOT, "include", "afx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(afx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from nerf_for_angiography_amd import _lib
    declared = header_functions()
    assert len(declared) >= 17
    assert sorted(_lib.exported_symbols()) == declared          # ctypes table == header
    lib = _lib.load()                                            # raises if a symbol is missing
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", nm), name
        assert getattr(lib, name) is not None


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from nerf_for_angiography_amd import _lib
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AfxError, match="not built"):
        _lib.load()


def test_context_queries_and_validation_without_gpu():
    from nerf_for_angiography_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    d = _lib.ModelDesc(3, 0, 0, 256, 8)
    assert lib.afx_create(C.byref(d), C.byref(h)) == 0
    assert lib.afx_query(h, _lib.Q_PARAM_COUNT, 0, 0, 0) == 527617      # 8x256, 3 inputs (Linear layers only)
    assert lib.afx_query(h, _lib.Q_K0, 0, 0, 0) == 3
    wo, bo, r, c = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
    assert lib.afx_param_layout(h, 9, C.byref(wo), C.byref(bo), C.byref(r), C.byref(c)) == 0
    assert (r.value, c.value, bo.value - wo.value) == (1, 256, 256)
    assert lib.afx_param_layout(h, 10, None, None, None, None) < 0
    for prec in (0, 1, 2):
        assert lib.afx_query(h, _lib.Q_PREPARED_BYTES, prec, 0, 0) > 1 << 20
    full = lib.afx_query(h, _lib.Q_BWD_WORKSPACE_FULL, 4096, 128, 2)
    assert full > lib.afx_query(h, _lib.Q_BWD_WORKSPACE_MIN, 4096, 0, 2) > 0
    # 16-/8-bit kernels: a chunk's per-layer stash plane stops at 4 GiB (32-bit stash offsets), so the FULL size saturates there:
    # 8x256 -> 32 768 tiles of 256 samples at 2 bytes per element, 65 536 tiles at 1 byte (f16s8)
    for prec, tiles in ((_lib.PREC["f16"], 32768), (_lib.PREC["bf16"], 32768), (_lib.PREC["f16s8"], 65536)):
        at_cap = lib.afx_query(h, _lib.Q_BWD_WORKSPACE_FULL, 0, tiles * 256, prec)
        assert lib.afx_query(h, _lib.Q_BWD_WORKSPACE_FULL, 0, 4 * tiles * 256, prec) == at_cap
        assert lib.afx_query(h, _lib.Q_BWD_WORKSPACE_FULL, 0, tiles * 256 // 2, prec) < at_cap
    assert lib.afx_query(h, _lib.Q_BWD_WORKSPACE_FULL, 0, 1 << 30, _lib.PREC["f32"]) > lib.afx_query(h, _lib.Q_BWD_WORKSPACE_FULL, 0, 1 << 29, _lib.PREC["f32"])
    lib.afx_destroy(h)
    # bad descriptors are refused with a message
    for bad in (_lib.ModelDesc(2, 0, 0, 256, 8), _lib.ModelDesc(3, 0, 0, 100, 8), _lib.ModelDesc(3, 1, 0, 64, 4),
                _lib.ModelDesc(3, 0, 0, 64, 0), _lib.ModelDesc(3, 0, 0, 64, 4, 3, 1.0),          # unknown activation
                _lib.ModelDesc(3, 1, 5, 64, 4, _lib.ACT["tanh"], 1.0)):                            # tanh / sine: without an input encoding only
        assert lib.afx_create(C.byref(bad), C.byref(h)) == -1
        assert lib.afx_last_error()
    d = _lib.ModelDesc(3, 0, 0, 128, 4, _lib.ACT["sine"], 15.0)       # forward-only activation: accepted; its workspace queries answer as usual
    assert lib.afx_create(C.byref(d), C.byref(h)) == 0 and lib.afx_query(h, _lib.Q_PARAM_COUNT, 0, 0, 0) == 128 * 3 + 128 + 4 * (128 * 128 + 128) + 129
    lib.afx_destroy(h)
    d = _lib.ModelDesc(3, 1, 5, 64, 4)
    assert lib.afx_create(C.byref(d), C.byref(h)) == 0
    assert lib.afx_query(h, _lib.Q_K0, 0, 0, 0) == 33
    assert lib.afx_query(h, _lib.Q_PARAM_COUNT, 0, 0, 0) == 64 * 33 + 64 + 4 * (64 * 64 + 64) + 65
    # null / malformed call arguments come back as error codes, not crashes
    args = _lib.RenderArgs()
    assert lib.afx_render_forward(h, 0, None, C.byref(args), None) == 0          # zero rays: nothing to do
    args.n_rays, args.n_samples = 8, 32
    assert lib.afx_render_forward(h, 0, None, C.byref(args), None) == -1
    assert b"origins" in lib.afx_last_error()
    assert lib.afx_mlp_infer(h, 7, None, None, 4, None, 0, None) == -1
    lib.afx_destroy(h)
    # hierarchical step with coarse re-use: its workspace query grows with the rays up to the 4 GiB-plane chunk, and a workspace that cannot
    # hold 8 rays of both sample sets is refused (the chunk-sizing loop terminates: rays per chunk shrink in steps of at least 8)
    d = _lib.ModelDesc(3, 0, 0, 256, 8)
    assert lib.afx_create(C.byref(d), C.byref(h)) == 0
    w9, w1k, w1m = (lib.afx_hier_workspace_bytes(h, n, 128, 64) for n in (9, 1000, 1 << 20))
    assert 0 < w9 < w1k < w1m and lib.afx_hier_workspace_bytes(h, 1 << 22, 128, 64) - w1m < (1 << 22) * 4 * 400
    args = _lib.RenderArgs()
    args.n_rays, args.n_samples, args.ray_mode, args.depth_mode = 9, 35, _lib.RAYS_ARRAYS, _lib.DEPTH_SHARED_Z
    args.origins = args.dirs = args.z = args.pixel = args.workspace = 4096      # (never dereferenced: validation and sizing come first)
    fixed_only = lib.afx_hier_workspace_bytes(h, 9, 35, 7) // 3
    for ws_bytes in (fixed_only, 1 << 20):
        args.workspace_bytes = ws_bytes
        rc = lib.afx_hier_train_step_mse(h, _lib.PREC["f16s8"], 4096, C.byref(args), 7, 4096, 4096, 1.0, None, 4096, None)
        assert rc == -2 and b"workspace" in lib.afx_last_error(), (rc, lib.afx_last_error())
    assert lib.afx_hier_train_step_mse(h, _lib.PREC["f16"], 4096, C.byref(args), 7, 4096, 4096, 1.0, None, 4096, None) == -1
    # the one-call grid iteration and the batched ray draws validate before they touch a device
    mt = _lib.MarchTrainArgs()
    assert lib.afx_march_train_step_mse(h, _lib.PREC["f16s8"], 4096, None, None) == -1
    assert lib.afx_march_train_step_mse(h, _lib.PREC["f16s8"], 4096, C.byref(mt), None) == 0 and mt.n_kept == 0          # zero rays: nothing to do
    mt.march.n_rays = 8
    assert lib.afx_march_train_step_mse(h, _lib.PREC["f16s8"], 4096, C.byref(mt), None) == -1 and b"null" in lib.afx_last_error()
    mt.march.origins = mt.march.dirs = mt.target = mt.pixel = mt.grad_flat = mt.workspace = 4096
    assert lib.afx_march_train_step_mse(h, _lib.PREC["f16"], 4096, C.byref(mt), None) == -1 and b"F16S8" in lib.afx_last_error()
    assert lib.afx_sample_batches(4096, 100, 0, 0, 4, 101, 4096, 4096, 1 << 20, None) == -1                                # k > n
    assert lib.afx_sample_batches(4096, 100, 0, 0, 4, 10, 4096, 4096, 16, None) == -2                                      # workspace
    assert lib.afx_sample_batches_workspace_bytes(900000, 16) >= 16 * 900000 * 4
    assert lib.afx_ray_offsets(None, 5, None, None, None, None) == -1
    lib.afx_destroy(h)


def test_fused_path_refuses_cpu_tensors():
    from nerf_for_angiography_amd.model.CPPN import CPPN
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd._lib import AfxError
    m = CPPN(model_def(4, 64))
    with pytest.raises(AfxError, match="no CPU fallback"):
        render_rays(m, torch.zeros(4, 3), torch.zeros(4, 3), 32, 0.0, 1.0)
    md16 = dict(model_def(4, 64, act="tanh"), precision="f16")      # tanh / sine train in the exact-fp32 kernels only
    with pytest.raises(NotImplementedError):
        render_rays(CPPN(md16), torch.zeros(4, 3), torch.zeros(4, 3), 32, 0.0, 1.0)
    with pytest.raises(NotImplementedError):
        render_rays(CPPN(model_def(4, 64, late=4)), torch.zeros(4, 3), torch.zeros(4, 3), 32, 0.0, 1.0)


# ---------------------------------------------------------------- CPPN mirror
CASES = {"none_relu_4x64": model_def(4, 64), "none_tanh_4x64": model_def(4, 64, act="tanh"),
         "none_sine_4x64": model_def(4, 64, act="sine"), "none_relu_4x64_late4": model_def(4, 64, late=4),
         "fourier_relu_4x64": model_def(4, 64, pos_enc="fourier"), "none_relu_8x256": model_def(8, 256)}


@pytest.mark.parametrize("name", list(CASES))
def test_cppn_mirror_state_dict_and_forward(golden, name):
    """Same state-dict keys/shapes as the reference and, through PyTorch operators on the host, same outputs."""
    from nerf_for_angiography_amd.model.CPPN import CPPN
    g = golden("g4_cppn_" + name)
    ref_sd = {k[4:]: v for k, v in g.items() if k.startswith("sd__")}
    m = CPPN(CASES[name])
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    assert all(tuple(sd[k].shape) == ref_sd[k].shape for k in sd)
    m.load_state_dict({k: T(v) for k, v in ref_sd.items()})
    with torch.no_grad():
        y = m(T(g["x"]))
    assert rel_l2(y.numpy(), g["y"]) < 2e-6


def test_cppn_barf_schedule_is_literal(golden):
    from nerf_for_angiography_amd.model.CPPN import CPPN
    g = golden("g4_cppn_barf_relu_4x64")
    m = CPPN(model_def(4, 64, pos_enc="barf"))
    m.load_state_dict({k[4:]: T(v) for k, v in g.items() if k.startswith("sd__")})
    for a in (0.0, 1.5, 2.5, 5.0):
        m.update_barf_alpha(a, "pts")
        assert m.barf_alpha == a
        assert np.array_equal(m.barf_weights.detach().numpy(), g[f"w_alpha{a}"])
        with torch.no_grad():
            assert rel_l2(m(T(g["x"])).numpy(), g[f"y_alpha{a}"]) < 2e-6


def test_cppn_parameters_are_views_of_one_flat_buffer(tmp_path):
    from nerf_for_angiography_amd.model.CPPN import CPPN
    m = CPPN(model_def(4, 64))
    assert m.fused and m.flat_params.numel() == 3 * 64 + 64 + 4 * (64 * 64 + 64) + 65
    lins = [x for x in m.early_pts_layers if isinstance(x, torch.nn.Linear)] + [m.output_linear[0]]
    base = m.flat_params.data_ptr()
    off = 0
    for lin in lins:
        assert lin.weight.data_ptr() == base + 4 * off
        off += lin.weight.numel()
        assert lin.bias.data_ptr() == base + 4 * off
        off += lin.bias.numel()
    # an optimizer step and load_state_dict write through to the flat buffer and bump the cache key
    before = m.flat_params.clone()
    v0 = tuple(p._version for p in m._hip_params())
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    m(torch.randn(16, 3)).sum().backward()
    opt.step()
    assert not torch.equal(before, m.flat_params)
    assert tuple(p._version for p in m._hip_params()) != v0
    sd = {k: torch.zeros_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert float(m.flat_params.abs().max()) == 0.0 and lins[0].weight.data_ptr() == base
    # checkpoint dictionary layout of CPPN.save
    path = str(tmp_path / "ckpt.pth")
    m.save(path, {"epochs": 3})
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"version", "parameters", "training_information", "model"} and ck["version"] == "v0.00"
    assert list(ck["model"].keys()) == list(m.state_dict().keys())


# ---------------------------------------------------------------- helper mirrors on the host
def test_cppn_rebinding_data_is_detected():
    """`p.data = ...` detaches a parameter from the flat buffer the kernels read; the model re-flattens."""
    from nerf_for_angiography_amd.model.CPPN import CPPN
    torch.manual_seed(0)
    m = CPPN(model_def(4, 64))
    lin = m._linears()[1]
    new_w = torch.randn_like(lin.weight)
    lin.weight.data = new_w.clone()                      # what torch.nn.utils.vector_to_parameters does
    assert lin.weight.data_ptr() != m.flat_params.data_ptr() + m._layout()[0][1][0] * 4
    m._check_views()
    wo = m._layout()[0][1][0]
    assert lin.weight.data_ptr() == m.flat_params.data_ptr() + wo * 4
    assert torch.equal(m.flat_params[wo:wo + 64 * 64].view(64, 64), new_w)
    m.invalidate()                                        # no engine yet: a no-op, must not raise


def test_pose_and_ray_helpers_match_golden(golden):
    from nerf_for_angiography_amd.phantomdata import proj_helpers as ph
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values, get_depth_values
    g = golden("g1_pose")
    for a, mref in zip(g["args"][::7], g["mats"][::7]):
        np.testing.assert_allclose(ph.source_matrix(g["src_pt"], a[0], a[1], a[2], list(a[3:6])), mref, atol=1e-12)
    g2 = golden("g2_rays")
    w, h, f = g2["b_whf"]
    o, d, mat, ii, jj = get_ray_values(35.0, -20.0, 5.0, np.array([0, 0, f + 200.0]), int(w), int(h), float(f), "cpu",
                                       np.array([3.0, -2.0, 1.0]))
    assert np.array_equal(mat, g2["b_pose"]) and ii.shape == (int(h), int(w))
    np.testing.assert_allclose(d.numpy(), g2["b_d"], atol=1e-15)
    assert np.array_equal(get_depth_values(float(f) + 100, float(f) + 300, 8, "cpu", stratified=False).numpy(), g2["b_z"])


def test_nerf_helpers_host_paths_match_golden(golden):
    from nerf_for_angiography_amd.nerf import nerf_helpers as nh
    from nerf_for_angiography_amd._lib import AfxError
    g3 = golden("g3_stratify")
    torch.manual_seed(100 + 32)
    assert np.array_equal(nh.randomize_depth(T(g3["z32"]), "cpu").numpy(), g3["out32"])
    g5 = golden("g5_render")
    np.testing.assert_allclose(nh.cumprod_exclusive(T(g5["cumprod_in"])).numpy(), g5["cumprod_out"], rtol=1e-7)
    sig, rgb = T(g5["n_z2_sigma"]), T(g5["n_z2_rgb"])
    np.testing.assert_allclose(nh.get_ray_entropy(sig, rgb).numpy(), g5["n_z2_entropy"], rtol=2e-6, atol=1e-30)
    g7 = golden("g7_sample_pdf")
    out = nh.sample_pdf(T(g7["a_bins"]), T(g7["a_w"]), 16, "cpu", u=T(g7["a_u"]))
    assert rel_l2(out.numpy(), g7["a_out"]) < 1e-7
    assert [b.shape[0] for b in nh.get_minibatches(torch.zeros(10, 3), 4)] == [4, 4, 2]
    # compositing has no CPU fallback (the oracle holds the maths, tests/test_oracle_golden.py pins it to G5) and only the
    # one-channel configuration the reference trains
    with pytest.raises(AfxError, match="no CPU fallback"):
        nh.render_volume_density(T(g5["raw_n"]), T(g5["d"]), T(g5["z2"]))
    with pytest.raises(AfxError, match="no CPU fallback"):      # (2-channel / mean-relu branches: PyTorch-ROCm operators, GPU only)
        nh.render_volume_density(T(g5["raw_c2"]), T(g5["d"]), T(g5["z2"]))
    # ... whose arithmetic is the reference's (G5 pins the 2-channel branch): the operator route, called directly on host tensors
    from oracle import angio_oracle as orc
    got = nh._render_volume_density_ops(T(g5["raw_c2"]), T(g5["d"]), T(g5["z2"]))
    want = orc.render_volume_density(T(g5["raw_c2"]), T(g5["d"]), T(g5["z2"]))
    for a, b in zip(got[:4], want[:4]):
        assert rel_l2(a.numpy(), b.numpy()) < 1e-6
    raw3 = torch.randn(6, 9, 3)
    got = nh._render_volume_density_ops(raw3, T(g5["d"])[:6], T(g5["z2"])[:6, :9])
    want = orc.render_volume_density(raw3, T(g5["d"])[:6], T(g5["z2"])[:6, :9])
    for a, b in zip(got[:4], want[:4]):
        assert rel_l2(a.numpy(), b.numpy()) < 1e-6
    with pytest.raises(AfxError, match="no CPU fallback"):
        nh.fine_sampling(torch.linspace(0, 1, 8), torch.rand(3, 8), torch.zeros(3, 3), torch.ones(3, 3), None, None, 4, 64)


def test_acc_helpers_host_paths():
    from nerf_for_angiography_amd.nerf import nerf_helpers_acc as na
    from oracle import angio_oracle as orc
    o, d = torch.zeros(5, 3), torch.ones(5, 3)
    ri, ts, te = na.acc_ray_marching(None, None, None, o, d, 16, 1400.0, 1600.0)
    ri_o, ts_o, te_o = orc.march_uniform(1400.0, 1600.0, 16, 5)
    assert torch.equal(ri.long(), ri_o) and torch.equal(ts, ts_o) and torch.equal(te, te_o)
    from nerf_for_angiography_amd._lib import AfxError
    with pytest.raises(AfxError, match="no CPU fallback"):
        na.acc_render_volume_density(torch.randn(80, 1), ri, ts, te, 5, 16)
    assert na.acc_update_n_step(None, None, 0) is None


def test_occupancy_grid_march_oracle_properties():
    """The oracle's restatement of nerfacc 0.3.x (parity unpinned): grid update rule, grid skipping on the fixed-step
    lattice, render_visibility (alpha threshold skips without attenuating, early termination), packed ray-sorted
    output.  The product runs these on HIP kernels (tests/test_gpu_parity.py compares the two)."""
    from oracle import angio_oracle as orc
    torch.manual_seed(0)
    aabb = torch.tensor([-100.0, -100, -100, 100, 100, 100])
    res = (16, 16, 16)
    occs = torch.zeros(16 ** 3)
    ball = lambda x: (x.norm(dim=-1) < 40).float() * 0.5          # dense ball of radius 40
    for step in range(4):
        cells = torch.arange(16 ** 3)
        pts = orc.grid_jittered_points(cells, torch.rand(cells.numel(), 3), aabb, res)
        occs, binary = orc.grid_update(occs, cells, ball(pts), 0.95, 1e-2)
    assert 0.03 < binary.float().mean() < 0.25
    idx, inside = orc.grid_cell_index(torch.tensor([[0.0, 0, 0], [90.0, 90, 90], [500.0, 0, 0]]), aabb, res)
    assert bool(binary[idx[0]]) and not bool(binary[idx[1]]) and not bool(inside[2])
    # a cell drawn twice takes the max over its draws, from the decayed OLD value
    o2, _ = orc.grid_update(torch.tensor([1.0, 0.2]), torch.tensor([0, 0, 1]), torch.tensor([0.3, 0.99, 0.1]), 0.5, 1e-2)
    assert torch.allclose(o2, torch.tensor([0.99, 0.1]))
    o = torch.tensor([[0.0, 0.0, 1500.0]]).repeat(3, 1)
    d = torch.tensor([[0.0, 0.0, -1.0], [0.02, 0.0, -1.0], [0.5, 0.5, -1.0]])
    tmin, tmax = orc.ray_aabb(o, d, aabb)
    assert abs(float(tmin[0]) - 1400) < 1e-3 and abs(float(tmax[0]) - 1600) < 1e-3 and float(tmin[2]) == 1e10
    ri, ts, te = orc.march_grid(o, d, aabb, 1400.0, 1600.0, 2.0)
    assert int((ri == 0).sum()) == 100 and int((ri == 2).sum()) == 0            # dense inside the box, miss -> nothing
    ri, ts, te = orc.march_grid(o, d, aabb, 1400.0, 1600.0, 2.0, binary.view(*res), aabb)
    mid0 = ((ts + te) / 2)[ri == 0]
    assert 30 <= mid0.numel() <= 60 and float(mid0.min()) > 1440 and float(mid0.max()) < 1560   # only the ball's cells
    assert torch.all(ri[1:] >= ri[:-1]) and torch.allclose(te - ts, torch.full_like(ts, 2.0))
    # render_visibility: T = 1, 0.1, 0.01 (< eps): two steps kept; thin samples are skipped and do not attenuate
    ri1 = torch.zeros(6, dtype=torch.int64)
    assert orc.render_visibility(torch.full((6,), 0.9), ri1, 2e-2, 0.0).tolist() == [True, True, False, False, False, False]
    assert orc.render_visibility(torch.full((6,), 1e-4), ri1, 1e-4, 1e-3).sum() == 0
    mixed = torch.tensor([1e-4, 0.9, 1e-4, 0.9, 0.9, 0.9])
    assert orc.render_visibility(mixed, ri1, 2e-2, 1e-3).tolist() == [False, True, False, True, False, False]
    # two rays in one packed list: T restarts
    assert orc.render_visibility(torch.full((4,), 0.95), torch.tensor([0, 0, 1, 1]), 0.1, 0.0).tolist() == [True, False, True, False]


def test_occupancy_product_refuses_cpu():
    from nerf_for_angiography_amd.nerf.occupancy import OccupancyGrid, ray_marching
    from nerf_for_angiography_amd.nerf import nerf_helpers_acc as na
    from nerf_for_angiography_amd._lib import AfxError
    grid = OccupancyGrid(roi_aabb=torch.tensor([-1.0, -1, -1, 1, 1, 1]), resolution=8)
    assert grid.binary.shape == (8, 8, 8) and grid.bits.numel() == 16
    with pytest.raises(AfxError, match="no CPU fallback"):
        grid.every_n_step(0, lambda x: x[:, :1])
    with pytest.raises(AfxError, match="no CPU fallback"):
        ray_marching(torch.zeros(2, 3), torch.ones(2, 3), far_plane=1.0)
    grid.eval()
    with pytest.raises(RuntimeError):
        grid.every_n_step(0, lambda x: x[:, :1])
    # restoring a trained grid the reference's way (visualization/visualization.py:162): the assignment reaches the module's mask;
    # the packed bitfield the march reads is made on the GPU, so a host-side grid refuses to hand out stale bits
    mask = torch.zeros(8, 8, 8, dtype=torch.bool)
    mask[2:5, 3, 1] = True
    grid._binary = mask
    assert torch.equal(grid.binary, mask) and torch.equal(grid._binary, mask)
    assert grid.query_occ(torch.tensor([[-0.3, -0.2, -0.7], [0.9, 0.9, 0.9]])).tolist() == [True, False]
    with pytest.raises(AfxError, match="move the grid to the GPU"):
        grid.bits
    with pytest.raises(ValueError):
        grid.binary = torch.zeros(7)
    assert na.acc_update_n_step(None, None, 0) is None


def test_bench_gpus_n_starts_torchrun_as_a_child(monkeypatch):
    """`python bench.py --gpus N` outside torchrun: the parent only spawns `python -m torch.distributed.run ... bench.py <flags>`
    (a child process, rendezvous on 127.0.0.1) and returns its exit code."""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = {}

    class FakeProc:
        pid = 0

        def __init__(self, cmd, **kw):
            seen["cmd"], seen["kw"] = cmd, kw

        def wait(self):
            return 7
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[-7]) == "bench.py" and seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_flop_model():
    sys.path.insert(0, ROOT)
    import bench
    fwd, dgrad, wgrad = bench.flops_per_sample(256, 8)
    assert (fwd, fwd + dgrad + wgrad) == (1050624, 3150336)
    fwd, dgrad, wgrad = bench.flops_per_sample(64, 4)
    assert fwd + dgrad + wgrad == 99456


# ---------------------------------------------------------------- data-parallel path on 2 gloo ranks
def _dp_problem():
    """A seeded global ray batch (37 rays: the two shards are unequal) for the data-parallel contract test."""
    g = torch.Generator().manual_seed(99)
    n = 37
    o = torch.tensor([[0.0, 0.0, 1500.0]]).repeat(n, 1)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g) * 0.03 + torch.tensor([0, 0, -1.0]), dim=-1)
    tgt = torch.rand(n, generator=g)
    return o, d, tgt


def _dp_flat_grad(m, o, d, tgt, n_global):
    """Flat gradient of sum_r (pixel_r - target_r)^2 / n_global over the given rays, through the module's
    torch-operator path (CPU) and the reference's acc compositing - what a rank's fused kernels produce on a GPU."""
    from nerf_for_angiography_amd.nerf.nerf_helpers_acc import acc_ray_marching
    from oracle import angio_oracle as orc
    s = 16
    ri, ts, te = acc_ray_marching(m, None, None, o, d, s, 1400.0, 1600.0)
    pos = o[ri.long()] + d[ri.long()] * (ts + te) / 2.0
    pix = orc.acc_render_volume_density(m(pos), ri, ts, te, o.shape[0])
    loss = ((pix - tgt) ** 2).sum() / n_global
    m.zero_grad()
    loss.backward()
    flat = torch.zeros(m.flat_params.numel())
    for p, gv in zip(m._hip_params(), m._split_grad(flat)):
        gv.copy_(p.grad)
    return flat


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from nerf_for_angiography_amd import dist as afx_dist
    from nerf_for_angiography_amd import render
    from nerf_for_angiography_amd.model.CPPN import CPPN
    r, w, dev = afx_dist.init_from_env("gloo")
    torch.manual_seed(rank)                       # different weights per rank until the broadcast
    m = CPPN(model_def(4, 64))
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-4.0)
    afx_dist.broadcast_parameters(m)
    flat0 = m.flat_params.clone()
    sync = afx_dist.GradSync().install()
    assert render._grad_hook is sync
    o, d, tgt = _dp_problem()
    n = o.shape[0]
    start, count = afx_dist.shard(n, r, w)
    g = _dp_flat_grad(m, o[start:start + count], d[start:start + count], tgt[start:start + count], n)
    render._grad_hook(g)                          # what _RenderFn.backward / train_step_mse call: SUM over ranks
    grid = afx_dist.density_grid_sharded(m, 100.0, 6)      # 7^3 = 343 points: unequal point ranges, one all-gather
    g_union = _dp_flat_grad(m, o, d, tgt, n) if rank == 0 else None      # the 1-rank gradient of the union batch
    # the DEFAULT train_step_mse call (no n_global) under this hook: the ranks' ray counts are all-reduced (19 + 18 = 37)
    n_default = render._global_rays(count, None, torch.device("cpu"))
    # autograd path: dist.global_mse divides by the all-reduced count; a local-mean loss needs GradSync(local_mean=True)
    gm = afx_dist.global_mse(tgt[start:start + count] * 0.5, tgt[start:start + count])
    want_gm = float((((tgt[start:start + count] * 0.5) - tgt[start:start + count]) ** 2).sum() / n)
    lm = afx_dist.GradSync(local_mean=True)
    v = torch.full((4,), float(rank + 1))
    lm(v)                                         # mean over ranks of (1, 2) = 1.5
    render._grad_hook = lm
    n_local_mean = render._global_rays(count, None, torch.device("cpu"))
    render._grad_hook = sync
    extra = (n_default, abs(float(gm) - want_gm) < 1e-7, v.tolist(), n_local_mean)
    with torch.no_grad():
        t = torch.linspace(-100.0, 100.0, 7, dtype=torch.float64).float()
        gy, gx, gz = torch.meshgrid(t, t, t, indexing="ij")
        want = torch.sigmoid(m(torch.stack([gx, gy, gz], -1).reshape(-1, 3))).reshape(7, 7, 7)
    q.put((rank, flat0[:8].tolist(), g.tolist(), None if g_union is None else g_union.tolist(), start, count,
           bool(torch.allclose(grid, want, atol=1e-6)), extra))
    afx_dist.GradSync.uninstall()
    dist.destroy_process_group()


def test_two_rank_gloo_grad_sync_and_sharding():
    """2 ranks, UNEQUAL ray shards (19 + 18 rays), real gradients: the all-reduced gradient equals the 1-rank
    gradient of the union batch (loss = mean over the global batch, SUM all-reduce, no division)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, w0, g0, u0, s0, c0, d0, x0), (r1, w1, g1, u1, s1, c1, d1, x1) = res
    assert x0 == (37, True, [1.5] * 4, 19) and x1 == (37, True, [1.5] * 4, 18)      # default n_global / global_mse / local-mean mode
    assert d0 and d1                               # the sharded density grid equals the single-process grid on both ranks
    assert w0 == w1                                # rank 0's weights everywhere
    assert g0 == g1                                # identical synced gradients on both ranks
    assert (s0, c0, s1, c1) == (0, 19, 19, 18)
    g0, u0 = np.asarray(g0), np.asarray(u0)
    assert np.linalg.norm(u0) > 0
    assert rel_l2(g0, u0) < 1e-5                   # N-rank == 1-rank up to fp32 summation order
    from nerf_for_angiography_amd.dist import shard
    assert [shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]


# ---------------------------------------------------------------- dataset wire format / load_data
def test_dataset_wire_format_round_trip(tmp_path):
    from nerf_for_angiography_amd.phantomdata import dataset as ds
    from nerf_for_angiography_amd.nerf.nerf_helpers import sample_pixel_rays
    angles = ds.angle_grid(90.0, 1, (90, 0))
    assert len(angles) == 5
    proj_df, ray_df = ds.make_synthetic_dataset(angles[:2] + angles[-1:], img_size=12, depth_samples_per_ray=40)
    assert list(proj_df.columns) == ds.PROJ_COLUMNS and list(ray_df.columns) == ds.RAY_COLUMNS
    assert len(ray_df) == 3 * 144 and 0.0 <= ray_df["pixel_value"].min() and ray_df["pixel_value"].max() <= 1.0
    folder = str(tmp_path / "ct")
    name = "background-90.0-1.0-[90, 0]"
    p, r = ds.save_dataset(proj_df, ray_df, folder, name, binary=False)
    assert open(p).readline().count(";") == len(ds.PROJ_COLUMNS)      # ';'-separated, index column first
    proj2, ray2, store, unseen = ds.load_data("ct", name, False, False, 12, 90.0, data_root=str(tmp_path))
    assert store == folder and unseen is None
    np.testing.assert_allclose(ray2.to_numpy(), ray_df.to_numpy(), rtol=1e-12)
    assert proj2["tform_cam2world"].iloc[1] == proj_df["tform_cam2world"].iloc[1]
    np.testing.assert_allclose(np.array(proj2["image_data"].iloc[0]), np.array(proj_df["image_data"].iloc[0]), rtol=1e-12)
    with pytest.raises(FileNotFoundError):
        ds.load_data("ct", "nope", False, False, 12, 90.0, data_root=str(tmp_path))
    # the frames feed the reference's weighted pixel sampler unchanged
    ray2["ray_origins"] = ray2[["ray_origins_x", "ray_origins_y", "ray_origins_z"]].to_numpy().tolist()
    ray2["ray_directions"] = ray2[["ray_directions_x", "ray_directions_y", "ray_directions_z"]].to_numpy().tolist()
    o, d, pix = sample_pixel_rays(ray2, 50, "cpu", weights="distance_pixel_value")
    assert o.shape == (50, 3) and d.shape == (50, 3) and pix.shape == (50,) and pix.dtype == torch.float32
    w = ds.sampling_weights(np.array(proj_df["image_data"].iloc[0]), "segmentation")
    assert w.shape == (12, 12) and w.min() > 0
    with pytest.raises(NotImplementedError):
        ds.sampling_weights(w, "frangi")


def test_training_driver_flags_match_reference():
    from nerf_for_angiography_amd.nerf.run_nerf_acc import build_parser
    ns = build_parser().parse_args(["--limited_size", "90", "--number_angles", "2", "--center_point", "[90,0]", "--binary", "True",
                                    "--sampling_strategy", "random", "--data_name", "ct", "--num_layers", "8",
                                    "--num_hidden_units", "256"])
    assert (ns.limited_size, ns.num_layers, ns.binary) == ("90", "8", "True")     # strings, cast later as upstream
