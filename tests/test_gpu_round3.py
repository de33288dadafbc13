"""GPU parity tests added in round 3 (run with -m gpu on an MI355X): the legs VERDICT r2 found unpinned -
the hierarchical (C3) fine pass WITH gradients, the density grid at every inference precision and at the
reference's 201^3 size, training from a dataset on disk through the device sampler - and the restored
occupancy grid, the operator-route compositing branches and the data-parallel default call."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from test_gpu_parity import DEV, T, TOL, make_model, load_sd, _bench_model, _grads_by_name

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ C3 with gradients
def _c3_setup(n_rays, seed=3):
    """Rays of one 512^2 projection (sub-sampled), the 8x256 model of the C3 tests (output bias -26: with the dense convention's
    1e10 tail the projection is exactly 0 at ordinary weights), coarse depths, uniforms, random targets."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W, SC, NF = 512, 128, 64
    o, d, m44, _, _ = get_ray_values(100.0, -20.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, "cpu")
    g = torch.Generator().manual_seed(seed)
    pick = torch.randperm(W * W, generator=g)[:n_rays] if n_rays < W * W else torch.arange(W * W)
    o, d = o.reshape(-1, 3)[pick].float().contiguous(), d.reshape(-1, 3)[pick].float().contiguous()
    z = orc.depth_values(1400.0, 1600.0, SC)
    u = torch.rand(o.shape[0], NF, generator=g)
    tgt = torch.rand(o.shape[0], generator=g)
    return o, d, z, u, tgt, SC, NF


def test_c3_hierarchical_backward_vs_oracle():
    """Config C3's training step - coarse dense render (no grad) -> weights -> sample_pdf / merge -> fine render with PER-RAY
    depths -> MSE -> backward (nerf/nerf_helpers.py:178-195 as tools/c3_only.py composes it) - with GRADIENTS against the CPU
    oracle's autograd through the same composition: 8x256, 128 + 64 samples, 320 rays of a 512^2 projection, exact-fp32 kernels.
    The fine pass is given the oracle's merged depths so that an inverse-CDF bin that flips in the last bit does not
    enter the gradient comparison (the depths themselves are compared first)."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd.engine import fine_depths
    o, d, z, u, tgt, SC, NF = _c3_setup(320)
    m = _bench_model("f32", seed=5)
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-26.0)
    cfg = dict(num_early_layers=8, num_filters=256)
    params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    leaves = {k: v.requires_grad_(True) for k, v in params.items() if k.startswith(("early", "output"))}
    fn = lambda p: orc.cppn_forward(p, cfg, params)
    with torch.no_grad():
        raw_c = orc.get_predictions(fn, orc.points_dense(o, d, z).reshape(-1, 3), 65536).reshape(-1, SC, 1)
        _, _, w_c, _, _ = orc.render_volume_density(raw_c, d, z)
        zf_c = orc.fine_depths(z, w_c, u, o.shape[0])
    raw_f = orc.get_predictions(fn, orc.points_dense(o, d, zf_c).reshape(-1, 3), 65536).reshape(-1, SC + NF, 1)
    rgb_c, _, _, _, _ = orc.render_volume_density(raw_f, d, zf_c)
    loss_c = torch.nn.functional.mse_loss(rgb_c, tgt)
    loss_c.backward()
    assert float(rgb_c.detach().max()) > 1e-3 and float(loss_c) > 0
    # GPU: the same step
    od, dd = o.to(DEV), d.to(DEV)
    with torch.no_grad():
        coarse = render_rays(m, od, dd, mode="dense", z=z.to(DEV), want_aux=True)
        zf = fine_depths(z.to(DEV), coarse.weights, u.to(DEV))
    assert rel_l2(coarse.weights.cpu().numpy(), w_c.numpy()) < 1e-4
    same = (zf.cpu() - zf_c).abs().max(-1).values < 1e-2
    assert float(same.float().mean()) > 0.98
    fine = render_rays(m, od, dd, mode="dense", z=zf_c.to(DEV))
    loss = torch.nn.functional.mse_loss(fine.rgb_map, tgt.to(DEV))
    loss.backward()
    assert rel_l2(fine.rgb_map.detach().cpu().numpy(), rgb_c.detach().numpy()) < 1e-3      # exp(-sigma 1e10 ||d||) amplifies raw's 1e-6
    assert abs(float(loss) - float(loss_c)) < 1e-4 * float(loss_c) + 1e-9
    got = _grads_by_name(m)
    num = sum(float(((torch.from_numpy(got[k]).double() - leaves[k].grad.double()) ** 2).sum()) for k in leaves)
    den = sum(float((leaves[k].grad.double() ** 2).sum()) for k in leaves)
    assert den > 0 and (num / den) ** 0.5 < 1e-3, (num / den) ** 0.5
    for k in leaves:
        assert rel_l2(got[k], leaves[k].grad.numpy()) < 3e-3, k


@pytest.mark.parametrize("prec", ["f16", "f16s8"])
def test_c3_full_size_hierarchical_backward(prec):
    """C3 at FULL size (512^2 rays, 128 coarse + 64 fine = 192 per-ray depths, 8x256): weight gradients of the fine pass at the
    training precisions against the exact-fp32 kernels over all 50 M samples (same merged depths for both, taken from the
    fp32 coarse pass), pixels included.  This is the backward BASELINE.md's C3 row times."""
    from nerf_for_angiography_amd.render import render_rays
    from nerf_for_angiography_amd.engine import fine_depths
    o, d, z, u, tgt, SC, NF = _c3_setup(512 * 512)
    od, dd, zd, td = o.to(DEV), d.to(DEV), z.to(DEV), tgt.to(DEV)
    m = _bench_model("f32", seed=5)
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-26.0)
        coarse = render_rays(m, od, dd, mode="dense", z=zd, want_aux=True)
        zf = fine_depths(zd, coarse.weights, u.to(DEV))
        del coarse
    assert torch.all(zf[:, 1:] >= zf[:, :-1])
    out = render_rays(m, od, dd, mode="dense", z=zf)
    torch.nn.functional.mse_loss(out.rgb_map, td).backward()
    pix32 = out.rgb_map.detach().clone()
    g32 = {k: torch.from_numpy(v).double() for k, v in _grads_by_name(m).items()}
    del out
    m.zero_grad(set_to_none=True)
    m.precision = prec
    out = render_rays(m, od, dd, mode="dense", z=zf)
    torch.nn.functional.mse_loss(out.rgb_map, td).backward()
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), pix32.cpu().numpy()) < 2e-3        # the 1e10 tail amplifies (as the forward-only test)
    g = {k: torch.from_numpy(v).double() for k, v in _grads_by_name(m).items()}
    tot = float(torch.sqrt(sum(((g[k] - g32[k]) ** 2).sum() for k in g32)) / torch.sqrt(sum((g32[k] ** 2).sum() for k in g32)))
    assert tot < TOL[prec]["grad"], tot
    for k in g32:
        assert float((g[k] - g32[k]).norm() / g32[k].norm()) < 3 * TOL[prec]["grad"], k


# ------------------------------------------------------------------------------------------------ density grid (R15)
GRID_BAR = 1e-4      # north-star bar on the reconstructed 3-D density grid


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "f16", "bf16"])
def test_density_grid_golden_every_precision(golden, prec):
    """G9 (17^3 grid of the reference's CPPN, layout D9) at every inference precision: the strict ones meet the 1e-4 bar;
    f16 / bf16 do NOT (raw MLP outputs at 1e-3 / 1e-2) - which is why render.density_grid() evaluates in split-bf16 whatever
    the model's training precision is.  The reduced precisions are still bounded here so a regression shows."""
    from nerf_for_angiography_amd.render import density_grid
    g = golden("g9_density_grid")
    m = load_sd(make_model(4, 64, precision="f16s8"), g)
    err = rel_l2(density_grid(m, 100.0, 16, precision=prec).cpu().numpy(), g["sigma"])
    bar = {"f32": 1e-5, "bf16x3": GRID_BAR, "f16": 5e-3, "bf16": 5e-2}[prec]
    assert err < bar, (prec, err)
    if prec == "f32":      # the default: a training precision is upgraded to split bf16 for the grid, and model.precision is left alone
        assert m.precision == "f16s8"
        assert rel_l2(density_grid(m, 100.0, 16).cpu().numpy(), g["sigma"]) < GRID_BAR and m.precision == "f16s8"


def test_density_grid_reference_size_201():
    """The reference's own grid size: t = linspace(-100, 100, 201) -> 201^3 = 8.1 M points (visualization/visualization.py:100-102,209),
    8x256 model.  Default precision of density_grid (split bf16) against the exact-fp32 kernels on EVERY point and against the CPU
    oracle on 65 536 sub-sampled points, both under the 1e-4 bar; f16 is measured beside it (and misses the bar: not the default)."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import density_grid
    m = _bench_model("f16s8", seed=2)
    n = 200
    g_def = density_grid(m, 100.0, n)
    g_32 = density_grid(m, 100.0, n, precision="f32")
    g_16 = density_grid(m, 100.0, n, precision="f16")
    assert g_def.shape == (201, 201, 201)
    e_def = rel_l2(g_def.cpu().numpy(), g_32.cpu().numpy())
    e_16 = rel_l2(g_16.cpu().numpy(), g_32.cpu().numpy())
    assert e_def < GRID_BAR, e_def
    assert e_16 < 5e-3, e_16
    pts = orc.density_grid_points(100.0, n)
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(8))[:65536]
    cfg = dict(num_early_layers=8, num_filters=256)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = torch.sigmoid(orc.cppn_forward(pts[pick], cfg, params)).reshape(-1)
    assert rel_l2(g_def.reshape(-1)[pick.to(DEV)].cpu().numpy(), want.numpy()) < GRID_BAR
    assert rel_l2(g_32.reshape(-1)[pick.to(DEV)].cpu().numpy(), want.numpy()) < 1e-5
    print(f"density grid 201^3 vs fp32 kernels: default (bf16x3) {e_def:.2e}, f16 {e_16:.2e}")


# ------------------------------------------------------------------------------------------------ f-1: CSV on disk -> GPU training
def test_training_from_csv_dataset_with_device_sampler(tmp_path):
    """SURVEY 8f-1 on the GPU path: a dataset written in the reference's wire format (save_dataset: the two ';'-separated CSVs of
    phantomdata/cttoray.py:271-308), read back by load_data (the call at nerf/run_nerf_acc.py:82), turned into the device-resident
    ray table, sampled on the device and trained - no --synthetic.  The device draw on SUPPLIED uniforms equals the host's
    Efraimidis-Spirakis selection computed from the CSV's own columns (keys log(u)/w, k largest), row for row."""
    from nerf_for_angiography_amd.phantomdata import dataset as ds
    from nerf_for_angiography_amd.nerf.run_nerf_acc import main
    from nerf_for_angiography_amd import engine as eng
    angles = ds.angle_grid(90.0, 1, (90, 0))
    proj_df, ray_df = ds.make_synthetic_dataset(angles, img_size=20, depth_samples_per_ray=80)
    name = "background-90.0-1.0-[90, 0]"
    ds.save_dataset(proj_df, ray_df, str(tmp_path / "data" / "ct"), name, binary=False)
    del proj_df, ray_df
    # (1) the device sampler on the table built from the CSV == the host selection on the same uniforms
    p2, r2, _, _ = ds.load_data("ct", name, False, False, 20, 90.0, data_root=str(tmp_path / "data"))
    train = r2[r2["image_id"] != p2.index[-1]]
    # (a frame's multi-column to_numpy() is column-major and .float().to(device) keeps those strides: the engine re-lays such a table out)
    col = lambda stem: torch.from_numpy(train[[f"{stem}_x", f"{stem}_y", f"{stem}_z"]].to_numpy()).float()
    assert not col("ray_origins").is_contiguous()
    tab_o, tab_d = col("ray_origins"), col("ray_directions")
    tab_p = torch.from_numpy(train["pixel_value"].to_numpy()).float()
    tab_w = torch.from_numpy(train["distance_pixel_value"].to_numpy()).float()
    n, k = tab_o.shape[0], 256
    u = torch.rand(n, generator=torch.Generator().manual_seed(17)).clamp_min(1e-12)
    o, d, p, idx = eng.sample_rays(tab_o.to(DEV), tab_d.to(DEV), tab_p.to(DEV), tab_w.to(DEV), k, u=u.to(DEV))
    keys = np.log(u.double().numpy()) / tab_w.double().numpy()
    want = np.sort(np.argsort(-keys, kind="stable")[:k])
    got = idx.cpu().numpy()
    assert len(set(got.tolist())) == k
    # fp32 keys on the device vs float64 on the host: the sets may differ only where two keys tie within fp32 rounding at the cut
    assert len(set(got.tolist()) ^ set(want.tolist())) <= 2
    assert torch.equal(o.cpu(), tab_o[idx.cpu()]) and torch.equal(d.cpu(), tab_d[idx.cpu()]) and torch.equal(p.cpu(), tab_p[idx.cpu()])
    # (2) the driver trains from the files
    out = main(["--data_name", "ct", "--data_root", str(tmp_path / "data"), "--limited_size", "90", "--number_angles", "1",
                "--center_point", "[90, 0]", "--binary", "False", "--img_size", "20", "--n_iters", "120", "--display_every", "60",
                "--sample_size", "16", "--depth_samples", "64", "--num_layers", "4", "--num_hidden_units", "64",
                "--log_dir", str(tmp_path / "run")])
    h = out["history"]
    assert [r["iter"] for r in h] == [0, 60, 120] and h[-1]["train_loss"] < h[0]["train_loss"]
    assert all(np.isfinite(r["test_psnr"]) for r in h)
    assert os.path.exists(str(tmp_path / "run" / "coarsemodel.pth"))


# ------------------------------------------------------------------------------------------------ restored occupancy grid
def test_march_after_restoring_a_grid_matches_oracle():
    """The reference restores a trained grid with `acc_grid._binary = grid_occupancy` (visualization/visualization.py:162) and then
    marches: the assignment must reach the packed bitfield the HIP march reads (afx_grid_pack), also when it is made on the host
    before .to(device).  Indices bit-exact against the oracle's march on the same mask (nerfacc semantics: parity unpinned)."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.nerf.occupancy import OccupancyGrid, ray_marching
    torch.manual_seed(4)
    aabb = torch.tensor([-100.0, -100, -100, 100, 100, 100])
    res = 16
    c = (torch.stack(torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij"), -1).float() + 0.5) / res * 200 - 100
    mask = (c.norm(dim=-1) < 45) & (torch.rand(res, res, res) < 0.8)
    o = torch.tensor([[0.0, 0.0, 1500.0]]).repeat(40, 1)
    d = torch.nn.functional.normalize(torch.randn(40, 3) * 0.03 + torch.tensor([0, 0, -1.0]), dim=-1) * 1.0007
    ri_o, ts_o, te_o = orc.march_grid(o, d, aabb, 1400.0, 1600.0, 200.0 / 300, mask, aabb)
    assert ri_o.numel() > 100
    for where in ("device", "host"):
        grid = OccupancyGrid(roi_aabb=aabb, resolution=res)
        if where == "host":
            grid._binary = mask            # set before the grid reaches the GPU: packed by .to()
            grid = grid.to(DEV)
        else:
            grid = grid.to(DEV)
            grid._binary = mask.to(DEV)
        assert torch.equal(grid.binary.cpu(), mask)
        ri, ts, te = ray_marching(o.to(DEV), d.to(DEV), scene_aabb=aabb, grid=grid, near_plane=1400.0, far_plane=1600.0,
                                  render_step_size=200.0 / 300)
        assert torch.equal(ri.long().cpu(), ri_o) and torch.equal(ts.reshape(-1).cpu(), ts_o) and torch.equal(te.reshape(-1).cpu(), te_o), where
        q = grid.query_occ(c.reshape(-1, 3).to(DEV))
        assert torch.equal(q.cpu(), mask.reshape(-1))
    # the march keeps a step while its MID-POINT lies before t_max: a ray clipped at 1600 by the far plane, 300 steps of 2/3
    ri, ts, te = ray_marching(o[:1].to(DEV), torch.tensor([[0.0, 0.0, -1.0]], device=DEV), scene_aabb=None, grid=None, near_plane=1400.0,
                              far_plane=1600.2, render_step_size=200.0 / 300)
    assert ri.numel() == 300 and float(((ts + te) / 2).max()) < 1600.2      # step 300 starts at 1600.0 < far, but its mid-point is beyond


# ------------------------------------------------------------------------------------------------ operator-route branches
def test_render_volume_density_other_channel_counts_on_gpu(golden):
    """render_volume_density for 2 and > 2 output channels (nerf/nerf_helpers.py:67-87; branches the reference's training path never
    takes) run on PyTorch-ROCm operators on the device; the 2-channel branch is pinned by the reference's own output (G5)."""
    from nerf_for_angiography_amd.nerf import nerf_helpers as nh
    g5 = golden("g5_render")
    rgb, depth, w, ent, (sig, col) = nh.render_volume_density(T(g5["raw_c2"]), T(g5["d"]), T(g5["z2"]))
    assert rgb.is_cuda
    for got, key in ((rgb, "c2_z2_rgb"), (depth, "c2_z2_depth"), (w, "c2_z2_weights"), (ent, "c2_z2_entropy"), (sig, "c2_z2_sigma")):
        assert rel_l2(got.cpu().numpy(), g5[key]) < 1e-5, key
    # > 2 channels: relu(mean) density, then the absorption formulas - pinned by the reference's output as well (G5 raw_c3)
    for zk in ("z1", "z2"):
        rgb, depth, w, ent, (sig, col) = nh.render_volume_density(T(g5["raw_c3"]), T(g5["d"]), T(g5[zk]))
        for got, key in ((rgb, "rgb"), (depth, "depth"), (w, "weights"), (ent, "entropy"), (sig, "sigma")):
            assert rel_l2(got.cpu().numpy(), g5[f"c3_{zk}_{key}"]) < 1e-5, (zk, key)
    rgb, depth, w, ent, _ = nh.render_volume_density(T(g5["raw_c2"]), T(g5["d"]), T(g5["z1"]))
    assert rel_l2(rgb.cpu().numpy(), g5["c2_z1_rgb"]) < 1e-5 and rel_l2(w.cpu().numpy(), g5["c2_z1_weights"]) < 1e-5


def test_fine_sampling_with_a_model_outside_the_fused_kernels():
    """fine_sampling (nerf/nerf_helpers.py:178-195) with a tanh CPPN: not a fused configuration, so the fine pass takes the
    reference's own sequence on PyTorch-ROCm operators (points -> get_predictions -> render_volume_density); vs the oracle."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.model.CPPN import CPPN
    from nerf_for_angiography_amd.nerf.nerf_helpers import fine_sampling
    torch.manual_seed(6)
    md = dict(num_early_layers=3, num_late_layers=0, num_filters=64, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="tanh", fourier_sigma=5,
              num_img=1, device=torch.device(DEV), precision="bf16x3")      # (tanh at "f32" trains in the kernels: a fused configuration)
    m = CPPN(md).to(DEV)
    assert not m.fused
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-26.0)
    r, sc, nf = 64, 32, 16
    o = torch.tensor([[0.0, 0.0, 1500.0]]).repeat(r, 1)
    d = torch.nn.functional.normalize(torch.randn(r, 3) * 0.03 + torch.tensor([0, 0, -1.0]), dim=-1) * 1.001
    z = orc.depth_values(1400.0, 1600.0, sc)
    u, w_c = torch.rand(r, nf), torch.rand(r, sc)
    cfg = dict(num_early_layers=3, num_filters=64, act_func="tanh")
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    zf = orc.fine_depths(z, w_c, u, r)
    with torch.no_grad():
        raw = orc.cppn_forward(orc.points_dense(o, d, zf).reshape(-1, 3), cfg, params).reshape(r, sc + nf, 1)
        rgb_c, dep_c, _, ent_c, _ = orc.render_volume_density(raw, d, zf)
        rgb, dep, ent = fine_sampling(z.to(DEV), w_c.to(DEV), o.to(DEV), d.to(DEV), m, None, nf, 4096, u=u.to(DEV))
    assert rel_l2(dep.cpu().numpy(), dep_c.numpy()) < 1e-4
    assert rel_l2(rgb.cpu().numpy(), rgb_c.numpy()) < 1e-2          # exp(-sigma 1e10): see the C3 tests
    assert rel_l2(ent.cpu().numpy(), ent_c.numpy()) < 1e-3


def test_full_size_gpu_tests_at_the_bench_workspace_setting():
    """The full-size GPU tests run at the engine's 24 GiB default workspace (3+ ray chunks per 512^2 x 128 projection); bench.py
    ships 128 GiB (2 chunks of exactly 4 GiB stash planes).  The same fused step at BOTH settings against the exact-fp32 kernels,
    and against each other (chunking changes only the fp32 summation order of the partial sums)."""
    from nerf_for_angiography_amd.render import render_projection, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W, S = 512, 128
    _, _, m44, _, _ = get_ray_values(24.0, 8.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, "cpu")
    poses = torch.from_numpy(m44[None]).to(DEV)
    tgt = torch.rand(W * W, generator=torch.Generator().manual_seed(9)).to(DEV)
    m32 = _bench_model("f32")
    out = render_projection(m32, poses, W, W, 13.0 * W, S, 1400.0, 1600.0)
    (((out.rgb_map - tgt) ** 2).sum() / (W * W)).backward()
    g32 = torch.cat([p.grad.reshape(-1) for p in m32._hip_params()]).double()
    del m32, out
    flat = {}
    for gib in (24, 128):
        m = _bench_model("f16s8")
        m.engine.max_workspace_bytes = gib << 30
        train_step_mse(m, projection_spec(poses, W, W, 13.0 * W, S, 1400.0, 1600.0), tgt)
        flat[gib] = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
        assert float((flat[gib] - g32).norm() / g32.norm()) < TOL["f16s8"]["grad"], gib
        del m
        torch.cuda.empty_cache()
    assert float((flat[24] - flat[128]).norm() / flat[128].norm()) < 1e-5


# ------------------------------------------------------------------------------------------------ split-phase fused train step
def _ref_iteration_problem(n_rays, seed=11):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([[0.0, 0.0, 1500.0]]).repeat(n_rays, 1) + torch.randn(n_rays, 3, generator=g)
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g) * 0.03 + torch.tensor([0, 0, -1.0]), dim=-1)
    return o, d, torch.rand(n_rays, generator=g)


@pytest.mark.parametrize("layers,width,n_samples,n_rays", [(4, 128, 300, 5625), (8, 256, 192, 3000), (2, 64, 70, 33), (3, 128, 144, 1),
                                                              (4, 256, 300, 257)])
def test_split_train_step_vs_fp32_kernels_and_two_launch_path(layers, width, n_samples, n_rays):
    """afx_train_step_mse for rays that straddle the 256-sample workgroup tiles (the reference's own 300 samples per ray, nerf/run_nerf_acc.py:129;
    192 = the hierarchical 128 + 64; 70 -> 96 and 144 -> 160 padded): forward half, per-ray reduction, backward half - against the
    exact-fp32 kernels (render + autograd) and against the path it replaces (forward launch + backward kernel that recomputes the
    forward, here at the 16-bit-stash f16 precision, which has no split)."""
    from nerf_for_angiography_amd.render import render_rays, train_step_mse
    from nerf_for_angiography_amd.engine import RenderSpec
    torch.manual_seed(layers * 100 + width + n_samples)
    m = make_model(layers, width, precision="f32")
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-5.0)
    o, d, tgt = _ref_iteration_problem(n_rays)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)
    spec = RenderSpec(n_rays=n_rays, n_samples=n_samples, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)
    out = render_rays(m, o, d, n_samples, 1400.0, 1600.0, mode="acc")
    torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
    pix32 = out.rgb_map.detach().clone()
    g32 = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    res = {}
    for prec in ("f16s8", "f16"):
        m.zero_grad(set_to_none=True)
        m.precision = prec
        assert m.engine.fused_step_available(n_samples, prec) == (prec == "f16s8")
        loss, pix = train_step_mse(m, spec, tgt)
        res[prec] = (pix, torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double(), float(loss))
    assert torch.equal(res["f16s8"][0], res["f16"][0])                      # same forward arithmetic: identical pixels
    # (optical depths of 3 - 6 at these weights: a pixel's RELATIVE error is its optical depth's ABSOLUTE error, 1e-4 x od; the
    # 1e-4 pixel bar at benchmark-like transmittances is asserted by the full-size tests)
    few = n_rays * n_samples < 100000        # (few samples: the bf8 stash rounding has nothing to average over, see test_odd_shapes_vs_oracle)
    assert rel_l2(res["f16s8"][0].cpu().numpy(), pix32.cpu().numpy()) < (2e-2 if few else 2e-3)
    e8 = float((res["f16s8"][1] - g32).norm() / g32.norm())
    e16 = float((res["f16"][1] - g32).norm() / g32.norm())
    assert e16 < (4 if few else 1) * TOL["f16"]["grad"], e16
    assert e8 < (0.25 if few else TOL["f16s8"]["grad"]), e8
    assert abs(res["f16s8"][2] - float(torch.nn.functional.mse_loss(pix32, tgt))) < 1e-3 * res["f16s8"][2] + 1e-7


def test_split_train_step_chunking_and_per_ray_depths():
    """The split step over several ray chunks (chunks hold whole rays: multiples of lcm(s_pad, 256) / 256 tiles) and with the hierarchical
    pass's per-ray depths (dense convention, 128 + 64 = 192 depths per ray): same gradients as ONE chunk up to the order of the fp32
    partial sums, bit-identical pixels, and close to the exact-fp32 kernels."""
    from nerf_for_angiography_amd.render import render_rays, train_step_mse
    from nerf_for_angiography_amd.engine import RenderSpec
    n_rays, S = 20000, 192
    o, d, tgt = _ref_iteration_problem(n_rays, seed=5)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)
    z = torch.sort(1400.0 + 200.0 * torch.rand(n_rays, S, generator=torch.Generator().manual_seed(2)), dim=-1).values.to(DEV)
    out = {}
    for ws_mib in (24 << 10, 1500):
        m = _bench_model("f16s8", seed=3)
        with torch.no_grad():
            m.output_linear[0].bias.fill_(-26.0)              # dense convention: the 1e10 tail
        m.engine.max_workspace_bytes = ws_mib << 20
        spec = RenderSpec(n_rays=n_rays, n_samples=S, origins=o, dirs=d, mode="dense", z=z)
        loss, pix = train_step_mse(m, spec, tgt)
        out[ws_mib] = (pix, torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double())
        assert m.engine._ws.numel() <= (ws_mib << 20)
    assert torch.equal(out[24 << 10][0], out[1500][0])
    assert float((out[24 << 10][1] - out[1500][1]).norm() / out[1500][1].norm()) < 1e-5
    m.zero_grad(set_to_none=True)
    m.precision = "f32"
    r = render_rays(m, o, d, mode="dense", z=z)
    torch.nn.functional.mse_loss(r.rgb_map, tgt).backward()
    g32 = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    assert float(g32.norm()) > 0
    assert float((out[1500][1] - g32).norm() / g32.norm()) < TOL["f16s8"]["grad"]


def test_split_train_step_safe_waits_build_is_bit_identical():
    """The race detector (libafx_safe.so: every counted wait a full one) on the two half-kernels: 5 625 rays x 300 samples, 8x256."""
    from nerf_for_angiography_amd import build as afx_build
    from nerf_for_angiography_amd.engine import Engine, RenderSpec
    from nerf_for_angiography_amd.render import train_step_mse
    afx_build.build(variant="safe")
    o, d, tgt = _ref_iteration_problem(5625, seed=8)
    spec = RenderSpec(n_rays=5625, n_samples=300, origins=o.to(DEV), dirs=d.to(DEV), mode="acc", t_near=1400.0, t_far=1600.0)
    res = []
    for variant in ("", "safe"):
        m = _bench_model("f16s8", seed=1)
        m._engine = Engine(256, 8, "none", 0, variant=variant)
        _, pix = train_step_mse(m, spec, tgt.to(DEV))
        torch.cuda.synchronize()
        res.append((pix, torch.cat([p.grad.reshape(-1) for p in m._hip_params()])))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_force_split_matches_fused_kernel_at_full_size(monkeypatch):
    """AFX_FORCE_SPLIT=1 runs the two half-kernels also where the fused kernel applies (512^2 x 128): same pixels bit for bit, gradients
    equal up to rounding of the factored output-layer sums (sum g' H x dod instead of sum (dod g') H)."""
    from nerf_for_angiography_amd.engine import Engine
    from nerf_for_angiography_amd.render import train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import get_ray_values
    W, S = 512, 128
    _, _, m44, _, _ = get_ray_values(24.0, 8.0, 0.0, np.array([0, 0, 1500.0]), W, W, 13.0 * W, "cpu")
    spec = projection_spec(torch.from_numpy(m44[None]).to(DEV), W, W, 13.0 * W, S, 1400.0, 1600.0)
    tgt = torch.rand(W * W, generator=torch.Generator().manual_seed(9)).to(DEV)
    res = []
    for force in ("0", "1"):
        monkeypatch.setenv("AFX_FORCE_SPLIT", force)
        m = _bench_model("f16s8")
        m._engine = Engine(256, 8, "none", 0)          # the knob is read when the context is created
        _, pix = train_step_mse(m, spec, tgt)
        res.append((pix, torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()))
        del m
    assert torch.equal(res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).norm() / res[0][1].norm()) < 2e-3


def test_hierarchical_train_step_fused_pieces_vs_oracle_and_autograd_path():
    """render.hierarchical_train_step_mse - coarse forward leaving tau[R,S], afx_fine_depths_from_tau (weights formed per ray in the kernel),
    split-phase fused fine step - against (i) the oracle's fine depths from the oracle's coarse weights and (ii) the operator-by-operator path
    it replaces (render with want_aux -> fine_sampling -> mse_loss -> backward), at 8x256, 128 + 64 samples."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays, hierarchical_train_step_mse
    from nerf_for_angiography_amd.nerf.nerf_helpers import fine_sampling
    from nerf_for_angiography_amd.engine import fine_depths_from_tau, RenderSpec
    o, d, z, u, tgt, SC, NF = _c3_setup(4096)
    od, dd, zd, ud, td = o.to(DEV), d.to(DEV), z.to(DEV), u.to(DEV), tgt.to(DEV)
    m = _bench_model("f16s8", seed=5)
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-26.0)
    # (i) depths: tau from the exact-fp32 kernels -> in-kernel weights -> sample_pdf / merge, vs the oracle on 256 of the rays
    m.precision = "f32"
    with torch.no_grad():
        _, _, tau = m.engine.render_forward(m._prepared(), RenderSpec(
            n_rays=4096, n_samples=SC, origins=od, dirs=dd, mode="dense", z=zd), "f32", want_tau=True)
        zf = fine_depths_from_tau(zd, tau, ud)
    cfg = dict(num_early_layers=8, num_filters=256)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        raw_c = orc.cppn_forward(orc.points_dense(o[:256], d[:256], z).reshape(-1, 3), cfg, params).reshape(256, SC, 1)
        _, _, w_c, _, _ = orc.render_volume_density(raw_c, d[:256], z)
        zf_c = orc.fine_depths(z, w_c, u[:256], 256)
    same = (zf[:256].cpu() - zf_c).abs().max(-1).values < 1e-2
    assert float(same.float().mean()) > 0.97
    assert rel_l2(zf[:256].cpu()[same].numpy(), zf_c[same].numpy()) < 1e-6
    # (ii) the fused step vs the operator path, both at f16s8 / f16 arithmetic
    m.precision = "f16s8"
    loss, pix, z_all = hierarchical_train_step_mse(m, od, dd, zd, NF, td, u=ud)
    g_fused = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    m.zero_grad(set_to_none=True)
    with torch.no_grad():
        coarse = render_rays(m, od, dd, mode="dense", z=zd, want_aux=True)
    rgb, _, _ = fine_sampling(zd, coarse.weights, od, dd, m, None, NF, 131072, u=ud)
    torch.nn.functional.mse_loss(rgb, td).backward()
    g_ops = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    assert float(g_ops.norm()) > 0
    agree = (z_all - torch.sort(z_all, -1).values).abs().max() == 0
    assert bool(agree)
    assert rel_l2(pix.cpu().numpy(), rgb.detach().cpu().numpy()) < 2e-3
    assert float((g_fused - g_ops).norm() / g_ops.norm()) < 2 * TOL["f16s8"]["grad"]


# ------------------------------------------------------------------------------------------------ R7: tanh / sine in the forward kernels
def _act_model(g, act, layers=4, width=64, precision="f32", **extra):
    from nerf_for_angiography_amd.model.CPPN import CPPN
    md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func=act, fourier_sigma=5, num_img=1,
              device=torch.device(DEV), precision=precision, **extra)
    m = CPPN(md).to(DEV)
    return load_sd(m, g) if g is not None else m


@pytest.mark.parametrize("act,extra", [("tanh", {}), ("sine", {"sine_weights": 15})])
def test_tanh_sine_forward_kernels_vs_reference_fixture(golden, act, extra):
    """CPPN act_func 'tanh' / 'sine' (model/CPPN.py:53-60,278-300; G4 captured from the reference's CPPN with sine w0 = 15): the activation is an
    epilogue of the forward chain kernels.  Under torch.no_grad() the module's forward IS the kernel (exact-fp32 and split-bf16 at the
    1e-5 / 5e-5 bars of the ReLU fixtures; f16 bounded); with gradients recorded it is the exact-fp32 kernel pair at precision "f32" and the
    module's PyTorch operators at the 16-bit precisions, and all agree."""
    g = golden(f"g4_cppn_none_{act}_4x64")
    x = T(g["x"])
    m = _act_model(g, act, **extra)
    assert m.fused_forward and m.fused
    with torch.no_grad():
        y32 = m(x)
        assert rel_l2(y32.cpu().numpy(), g["y"]) < (1e-5 if act == "tanh" else 3e-5)      # sin(15 z): one ulp of z is 15 ulps of the argument
        m.precision = "bf16x3"
        # (split bf16 carries ~2^-17 per product: 4e-4 absolute on a first-layer pre-activation of raw coordinates +-100, which sin(15 z)
        # turns into 6e-3 rad: a sine model's strict precision is f32 - render.grid_precision picks it)
        assert rel_l2(m(x).cpu().numpy(), g["y"]) < (5e-5 if act == "tanh" else 3e-3)
        m.precision = "f16"
        assert rel_l2(m(x).cpu().numpy(), g["y"]) < (3e-3 if act == "tanh" else 3e-2)      # sin(15 z): f16 operands in front of a steep argument
        m.precision = "f32"
    y_k = m(x)                                    # gradients recorded, f32: the kernels (forward bit-identical to the no_grad launch)
    assert y_k.requires_grad and torch.equal(y_k.detach(), y32)
    y_k.sum().backward()
    gk = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    m.zero_grad(set_to_none=True)
    m.precision = "f16"
    assert not m.fused
    y_ops = m(x)                                  # gradients recorded, 16-bit precision: operator route
    assert y_ops.requires_grad and rel_l2(y_ops.detach().cpu().numpy(), g["y"]) < 3e-5
    y_ops.sum().backward()
    go = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    assert float((gk - go).norm() / go.norm()) < (1e-5 if act == "tanh" else 2e-4)


@pytest.mark.parametrize("act,extra", [("tanh", {}), ("sine", {"sine_weights": 2.0})])
def test_tanh_sine_render_and_density_grid_vs_oracle(act, extra):
    """Evaluation renders (in-kernel ray generation, acc convention) and the density grid of tanh / sine models under torch.no_grad(), 8 hidden
    layers of width 128, against the CPU oracle; asking a 16-bit precision for gradients through the fused renderer is refused."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays, density_grid
    torch.manual_seed(31)
    m = _act_model(None, act, layers=8, width=128, **extra)
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-4.0)
    o, d, _ = _ref_iteration_problem(300, seed=4)
    cfg = dict(num_early_layers=8, num_filters=128, act_func=act, **extra)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = orc.render_rays(o, d, cfg, params, near=1400.0, far=1600.0, n_samples=96, convention="acc")
        got = render_rays(m, o.to(DEV), d.to(DEV), 96, 1400.0, 1600.0, mode="acc").rgb_map
        assert rel_l2(got.cpu().numpy(), want.numpy()) < 1e-4
        grid = density_grid(m, 100.0, 12)
        pts = orc.density_grid_points(100.0, 12)
        sig = torch.sigmoid(orc.cppn_forward(pts, cfg, params)).reshape(13, 13, 13)
        assert rel_l2(grid.cpu().numpy(), sig.numpy()) < 1e-4
    m16 = _act_model(None, act, layers=8, width=128, precision="f16", **extra)      # 16-bit kernels: forward-only for tanh / sine
    with pytest.raises(NotImplementedError):
        render_rays(m16, o.to(DEV), d.to(DEV), 96, 1400.0, 1600.0, mode="acc")


@pytest.mark.parametrize("act,extra,layers,width", [("tanh", {}, 4, 64), ("sine", {"sine_weights": 2.0}, 4, 64), ("tanh", {}, 8, 256),
                                                    ("sine", {"sine_weights": 15}, 3, 128)])
def test_tanh_sine_training_in_the_fp32_kernels_vs_oracle_autograd(act, extra, layers, width):
    """tanh / sine models WITH gradients (model/CPPN.py:53-60,278-300): the exact-fp32 chain kernel keeps d act / dz per element in the slot
    its dZ_l stash later overwrites (ReLU: one mask bit in LDS).  Fused render + MSE backward and the points-mode module forward/backward
    (`get_predictions` route) against the CPU oracle's autograd."""
    from oracle import angio_oracle as orc
    from nerf_for_angiography_amd.render import render_rays
    torch.manual_seed(17)
    m = _act_model(None, act, layers=layers, width=width, **extra)
    assert m.fused and m.precision == "f32"
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-4.0)
    r, s = 193, 70
    o, d, tgt = _ref_iteration_problem(r, seed=13)
    if act == "sine" and extra["sine_weights"] > 10:      # sin(15 W x): unit-scale inputs as SIREN wants them (the reference feeds its scene coordinates as they are)
        o, near, far = o * 1e-3, 1.4, 1.6
    else:
        near, far = 1400.0, 1600.0
    cfg = dict(num_early_layers=layers, num_filters=width, act_func=act, **extra)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pix_c, loss_c, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=near, far=far, n_samples=s, convention="acc")
    out = render_rays(m, o.to(DEV), d.to(DEV), s, near, far, mode="acc")
    loss = torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV))
    loss.backward()
    assert rel_l2(out.rgb_map.detach().cpu().numpy(), pix_c.numpy()) < 1e-5
    got = _grads_by_name(m)
    assert set(grads_c) <= set(got)
    for k, v in grads_c.items():
        assert rel_l2(got[k], v.numpy()) < 2e-4, k
    # points mode: module forward / backward on explicit query points
    m.zero_grad(set_to_none=True)
    pts = (torch.rand(777, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1) * (0.1 if near < 10 else 100.0)
    w = torch.randn(777, generator=torch.Generator().manual_seed(3))
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    (orc.cppn_forward(pts, cfg, leaves).reshape(-1) * w).sum().backward()
    (m(pts.to(DEV)).reshape(-1) * w.to(DEV)).sum().backward()
    for k, v in leaves.items():
        if v.grad is not None:
            assert rel_l2(_grads_by_name(m)[k], v.grad.numpy()) < 2e-4, k


# ------------------------------------------------------------------------------------------------ fused packed step (grid-march iteration)
def _packed_problem(n_rays, seed, empty_every=0):
    """Rays, a ball-shaped occupancy grid, the march's packed samples (no visibility pruning so that the list is a pure function of the grid)."""
    from nerf_for_angiography_amd.nerf.occupancy import OccupancyGrid, ray_marching
    o, d, tgt = _ref_iteration_problem(n_rays, seed=seed)
    aabb = torch.tensor([-100.0, -100, -100, 100, 100, 100])
    res = 32
    c = (torch.stack(torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij"), -1).float() + 0.5) / res * 200 - 100
    mask = c.norm(dim=-1) < 60
    grid = OccupancyGrid(roi_aabb=aabb, resolution=res).to(DEV)
    grid._binary = mask.to(DEV)
    if empty_every:      # some rays that miss everything: no samples, pixel = 1
        d = d.clone()
        d[::empty_every] = torch.tensor([0.9, 0.3, -0.3])
    ri, ts, te, packed = ray_marching(o.to(DEV), d.to(DEV), scene_aabb=aabb, grid=grid, near_plane=1400.0, far_plane=1600.0,
                                      render_step_size=200.0 / 300, return_packed=True)
    return o.to(DEV), d.to(DEV), tgt.to(DEV), ri, ts, te, packed


@pytest.mark.parametrize("layers,width,n_rays,enc", [(4, 128, 5625, "none"), (8, 256, 2000, "none"), (2, 64, 3, "none"), (4, 128, 5625, "barf")])
def test_fused_packed_step_vs_operator_sequence_and_fp32(layers, width, n_rays, enc):
    """render.train_step_packed_mse - the reference's positions -> get_predictions -> acc_render_volume_density -> mse_loss -> backward
    (nerf/run_nerf_acc.py:289-306) as one fused pass over the grid march's packed samples - against that operator sequence through the mirrored
    functions at the exact-fp32 kernels (gradients) and at f16 (pixels bit-for-bit: same forward arithmetic); rays without samples render 1."""
    from nerf_for_angiography_amd.render import train_step_packed_mse
    from nerf_for_angiography_amd.nerf.nerf_helpers import get_predictions
    from nerf_for_angiography_amd.nerf.nerf_helpers_acc import acc_render_volume_density
    o, d, tgt, ri, ts, te, packed = _packed_problem(n_rays, seed=layers + width, empty_every=7)
    assert packed.n_groups * 32 >= ri.numel() > 0 and int(packed.group_offsets[-1]) == packed.n_groups
    cnt = torch.bincount(ri.long(), minlength=n_rays)
    assert int((cnt == 0).sum()) > 0 or n_rays < 7
    torch.manual_seed(3)
    m = make_model(layers, width, enc, precision="f32")
    if enc == "barf":
        m.update_barf_alpha(2.5, "pts")
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-5.0)

    def ops_step(prec):
        m.precision = prec
        m.zero_grad(set_to_none=True)
        pos = o[ri.long()] + d[ri.long()] * (ts + te) / 2.0
        pred, _ = acc_render_volume_density(get_predictions(m, pos, 131072), ri, ts, te, n_rays, 300)
        loss = torch.nn.functional.mse_loss(pred, tgt)
        loss.backward()
        return pred.detach(), torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double(), float(loss)

    pix32, g32, loss32 = ops_step("f32")
    pix16, _, _ = ops_step("f16")
    m.precision = "f16s8"
    m.zero_grad(set_to_none=True)
    loss, pix = train_step_packed_mse(m, o, d, packed, tgt)
    g8 = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    assert bool((pix[cnt == 0] == 1.0).all())
    few = ri.numel() < 100000
    assert rel_l2(pix.cpu().numpy(), pix16.cpu().numpy()) < 1e-6          # same f16 forward; the per-ray product is summed in a different order
    assert rel_l2(pix.cpu().numpy(), pix32.cpu().numpy()) < (2e-2 if few else 2e-3)
    assert abs(float(loss) - loss32) < 2e-3 * loss32 + 1e-7
    e = float((g8 - g32).norm() / g32.norm())
    assert e < (0.25 if few else TOL["f16s8"]["grad"]), e


def test_training_driver_grid_march_fused_packed_step(tmp_path):
    """--march grid: the reference's loop with the body behind the march as one fused packed step; --march grid_ops: call for call.  Both train
    (same seeds, same device sampler) and land on similar losses."""
    from nerf_for_angiography_amd.nerf.run_nerf_acc import main
    res = {}
    for march in ("grid", "grid_ops"):
        out = main(["--synthetic", "--img_size", "24", "--number_angles", "1", "--limited_size", "90", "--n_iters", "96", "--display_every", "48",
                    "--sample_size", "20", "--depth_samples", "96", "--num_layers", "4", "--num_hidden_units", "64", "--march", march,
                    "--log_dir", str(tmp_path / march)])
        h = out["history"]
        assert h[-1]["train_loss"] < h[0]["train_loss"] and all(np.isfinite(r["test_psnr"]) for r in h)
        assert h[-1]["marched_samples_per_iter"] > 0
        res[march] = h[-1]["train_loss"]
        assert os.path.exists(str(tmp_path / march / "vessel_acc_grid_binary.npy"))
    assert abs(res["grid"] - res["grid_ops"]) < 0.3 * res["grid_ops"]


def test_hip_graph_capture_of_the_split_train_step():
    """The split-phase step (forward half, per-ray reduction, backward half, weight gradients: ~12 launches and one memset) allocates nothing and
    never synchronises either: captured into a HIP graph at the reference's 5 625 x 300 and replayed, it reproduces the eager step bit for bit."""
    from nerf_for_angiography_amd.engine import RenderSpec
    o, d, tgt = _ref_iteration_problem(5625, seed=12)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)
    torch.manual_seed(2)
    m = make_model(4, 128, precision="f16s8")
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-5.0)
    spec = RenderSpec(n_rays=5625, n_samples=300, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)
    eng, prepared = m.engine, m._prepared()
    grad_eager = torch.zeros(eng.param_count, device=DEV)
    pix_eager = eng.train_step_mse(prepared, spec, tgt, 1.0 / 5625, grad_eager, "f16s8")      # also sizes the workspace
    torch.cuda.synchronize()
    grad_g = torch.zeros(eng.param_count, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            grad_g.zero_()
            pix_g = eng.train_step_mse(prepared, spec, tgt, 1.0 / 5625, grad_g, "f16s8")
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(pix_g, pix_eager) and torch.equal(grad_g, grad_eager)


# ------------------------------------------------------------------------------------------------ hierarchical step with coarse re-use
@pytest.mark.parametrize("n_rays,sc,nf,layers,width,shared_z", [(4096, 128, 64, 8, 256, True), (999, 96, 40, 4, 128, True), (300, 64, 33, 2, 64, False), (9, 35, 7, 3, 128, True)])
def test_hierarchical_step_with_coarse_reuse(n_rays, sc, nf, layers, width, shared_z):
    """afx_hier_train_step_mse (the coarse depths evaluated ONCE: forward half over the coarse set, sample_pdf, forward half over the new depths only,
    per-ray composite of the merged list, backward halves) against the step without re-use (coarse forward + split-phase fine step over all S + N_f
    depths): same merged depths, pixels equal up to the order of the optical-depth sum, gradients equal up to the stochastic rounding of
    the stash (the two paths visit the samples in different orders) - and both against the exact-fp32 kernels on the same merged depths.
    Ragged counts (S = 35 -> 64 padded, N_f = 7 / 33 / 40), per-ray coarse depths, several ray chunks."""
    from nerf_for_angiography_amd.render import render_rays, hierarchical_train_step_mse
    g = torch.Generator().manual_seed(n_rays + sc)
    o, d, tgt = _ref_iteration_problem(n_rays, seed=sc)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)
    z = torch.linspace(1400.0, 1600.0, sc)
    if not shared_z:
        z = torch.sort(1400.0 + 200.0 * torch.rand(n_rays, sc, generator=g), dim=-1).values
    z, u = z.to(DEV), torch.rand(n_rays, nf, generator=g).to(DEV)
    torch.manual_seed(layers + width)
    m = make_model(layers, width, precision="f16s8")
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0)
        m.output_linear[0].bias.fill_(-26.0)
    res = {}
    for reuse in (True, False):
        m.zero_grad(set_to_none=True)
        if reuse and n_rays == 4096:
            m.engine.max_workspace_bytes = 3 << 30          # several ray chunks
        loss, pix, z_all = hierarchical_train_step_mse(m, o, d, z, nf, tgt, u=u, reuse_coarse=reuse)
        m.engine.max_workspace_bytes = 24 << 30
        res[reuse] = (pix, z_all, torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double(), float(loss))
    assert torch.equal(res[True][1], res[False][1])                                   # the same merged depths, bit for bit
    assert torch.all(res[True][1][:, 1:] >= res[True][1][:, :-1])
    assert rel_l2(res[True][0].cpu().numpy(), res[False][0].cpu().numpy()) < 1e-5
    few = n_rays * (sc + nf) < 100000
    dg = float((res[True][2] - res[False][2]).norm() / res[False][2].norm())
    assert dg < (0.3 if few else 2 * TOL["f16s8"]["grad"]), dg
    m.zero_grad(set_to_none=True)
    m.precision = "f32"
    out = render_rays(m, o, d, mode="dense", z=res[True][1])
    torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
    g32 = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    assert float(g32.norm()) > 0
    # (the 1e10 tail amplifies the f16 error of the last sample's raw value, as in the C3 tests; a 2x64 model has few terms to average it over)
    assert rel_l2(res[True][0].cpu().numpy(), out.rgb_map.detach().cpu().numpy()) < (5e-2 if few else 2e-3)
    e = float((res[True][2] - g32).norm() / g32.norm())
    assert e < (0.25 if few else TOL["f16s8"]["grad"]), e


@pytest.mark.parametrize("enc", ["barf", "fourier"])
def test_split_train_step_with_input_encoding(enc, monkeypatch):
    """The split-phase step for ENCODED models (BARF / fourier, 33 encoded inputs; bf8 input stash, first layer and - fourier - the coefficient
    contraction as rows of k_wgrad_s8) at the reference's 5 625 x 300: against the exact-fp32 kernels (Linear gradients) and, for the trainable
    fourier coefficients, against the forward-launch + recomputing-backward path it replaces (AFX_NO_SPLIT=1)."""
    from nerf_for_angiography_amd.render import render_rays, train_step_mse
    from nerf_for_angiography_amd.engine import RenderSpec
    o, d, tgt = _ref_iteration_problem(5625, seed=21)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)
    spec = RenderSpec(n_rays=5625, n_samples=300, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)

    def model(prec):
        torch.manual_seed(4)
        m = make_model(4, 128, enc, precision=prec)
        if enc == "barf":
            m.update_barf_alpha(2.5, "pts")
        else:
            with torch.no_grad():
                m.fourier_coefficients.mul_(0.002)
            m.fourier_coefficients.requires_grad_(prec != "f32")
        with torch.no_grad():
            m.output_linear[0].weight.mul_(4.0)
            m.output_linear[0].bias.fill_(-5.0)
        return m

    m32 = model("f32")
    out = render_rays(m32, o, d, 300, 1400.0, 1600.0, mode="acc")
    torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
    g32 = torch.cat([p.grad.reshape(-1) for p in m32._hip_params()]).double()
    m = model("f16s8")
    assert m.engine.fused_step_available(300, "f16s8")
    _, pix = train_step_mse(m, spec, tgt)
    g8 = torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()
    assert rel_l2(pix.cpu().numpy(), out.rgb_map.detach().cpu().numpy()) < 2e-3
    assert float((g8 - g32).norm() / g32.norm()) < TOL["f16s8"]["grad"]
    if enc == "fourier":
        gc = m.fourier_coefficients.grad.double().clone()
        assert bool(torch.isfinite(gc).all()) and float(gc.abs().max()) > 0
        monkeypatch.setenv("AFX_NO_SPLIT", "1")
        m2 = model("f16s8")
        assert not m2.engine.fused_step_available(300, "f16s8")
        _, pix2 = train_step_mse(m2, spec, tgt)
        assert torch.equal(pix2, pix)
        gc2 = m2.fourier_coefficients.grad.double()
        assert float((gc - gc2).norm() / gc2.norm()) < 2 * TOL["f16s8"]["grad"]


@pytest.mark.parametrize("case", ["fused_8x256", "split_300", "barf_4x128", "hier_128_64"])
def test_six_bit_h_stash_variant_matches_the_default_library(case):
    """libafx_h6.so (build.py --variant=h6): the hidden activations stashed as bf6 (e3m2) with one E8M0 block scale per 32-sample group and
    64-feature tile pair, 12-byte LDS-DMA pieces and ds_read_b96_tr_b6 in k_wgrad_s8, A = bf8 x B = bf6 on the MX instruction.  Opt-in (it
    moves 12.5 % fewer stash bytes but is no faster), so it is pinned here: forward bit-identical to the default library, weight
    gradients within the default's own distance from the exact-fp32 kernels."""
    from nerf_for_angiography_amd import build as afx_build
    from nerf_for_angiography_amd.engine import Engine, RenderSpec
    from nerf_for_angiography_amd.render import render_rays, train_step_mse, hierarchical_train_step_mse
    afx_build.build(variant="h6")
    layers, width, enc = (8, 256, "none") if case in ("fused_8x256", "hier_128_64") else ((4, 128, "barf") if case == "barf_4x128" else (4, 128, "none"))
    n_rays, n_samples = {"fused_8x256": (2048, 128), "split_300": (1031, 300), "barf_4x128": (777, 64), "hier_128_64": (512, 128)}[case]
    o, d, tgt = _ref_iteration_problem(n_rays, seed=21)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)

    def model(prec, variant=None):
        torch.manual_seed(4)
        m = make_model(layers, width, pos_enc=enc, precision=prec)
        if enc == "barf":
            m.update_barf_alpha(2.5, "pts")
        with torch.no_grad():
            m.output_linear[0].weight.mul_(4.0)
            m.output_linear[0].bias.fill_(-26.0 if case == "hier_128_64" else -5.0)      # (dense convention: the last interval is 1e10 long)
        if variant is not None:
            m._engine = Engine(width, layers, enc, 5 if enc != "none" else 0, variant=variant)
        return m

    def step(m):
        if case == "hier_128_64":
            z = torch.linspace(1400.0, 1600.0, n_samples, device=DEV)
            u = torch.rand(n_rays, 64, generator=torch.Generator().manual_seed(3)).to(DEV)
            _, pix, _ = hierarchical_train_step_mse(m, o, d, z, 64, tgt, u=u)
        else:
            spec = RenderSpec(n_rays=n_rays, n_samples=n_samples, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)
            _, pix = train_step_mse(m, spec, tgt)
        torch.cuda.synchronize()
        return pix, torch.cat([p.grad.reshape(-1) for p in m._hip_params()]).double()

    pix8, g8 = step(model("f16s8", ""))
    pix6, g6 = step(model("f16s8", "h6"))
    assert torch.equal(pix8, pix6) and float(g8.norm()) > 0
    assert float((g6 - g8).norm() / g8.norm()) < 5e-3
    if case != "hier_128_64":
        m32 = model("f32")
        out = render_rays(m32, o, d, n_samples, 1400.0, 1600.0, mode="acc")
        torch.nn.functional.mse_loss(out.rgb_map, tgt).backward()
        g32 = torch.cat([p.grad.reshape(-1) for p in m32._hip_params()]).double()
        e8, e6 = float((g8 - g32).norm() / g32.norm()), float((g6 - g32).norm() / g32.norm())
        assert e6 < TOL["f16s8"]["grad"] and e6 < 1.5 * e8 + 1e-4, (e8, e6)


@pytest.mark.parametrize("n,k,prefetch", [(900000, 5625, 16), (20000, 512, 5), (1025, 1025, 3), (7, 3, 1)])
def test_batched_ray_sampler_equals_the_per_iteration_draws(n, k, prefetch):
    """afx_sample_batches / engine.RayBatchSampler: the batches of `prefetch` consecutive iterations from ONE launch sequence (blockIdx.y = the
    iteration) are, index for index, what sample_rays(seed, stream_id) draws per iteration (sample_pixel_rays, nerf/nerf_helpers.py:137-150);
    stream ids outside the prefetched block start a new block; ragged sizes, k = n, zero weights."""
    from nerf_for_angiography_amd import engine as eng
    g = torch.Generator().manual_seed(n)
    o, d, pix = torch.randn(n, 3, generator=g).to(DEV), torch.randn(n, 3, generator=g).to(DEV), torch.rand(n, generator=g).to(DEV)
    w = torch.rand(n, generator=g) + 0.01
    if n > 100 and k < n:
        w[::13] = 0.0
    w = w.to(DEV)
    sampler = eng.RayBatchSampler(o, d, pix, w, k, seed=11, prefetch=prefetch)
    for sid in list(range(4, 4 + 2 * prefetch + 1)) + [2, 1000, 1001]:
        bo, bd, bp, bidx = sampler.draw(sid)
        so, sd, sp, sidx = eng.sample_rays(o, d, pix, w, k, seed=11, stream_id=sid)
        assert torch.equal(bidx, sidx), sid
        assert torch.equal(bo, so) and torch.equal(bd, sd) and torch.equal(bp, sp)
    with pytest.raises(ValueError):
        eng.RayBatchSampler(o, d, pix, w, n + 1)


@pytest.mark.parametrize("fused", [False, True])
def test_prepared_weight_cache_follows_optimizer_steps(fused):
    """The re-tiled weights are cached between calls; torch.optim's fused implementations update the parameters without bumping any
    version counter (found when the driver moved to Adam(fused=True): ten steps rendered with the initial weights).  Every optimizer step of
    the process is counted into the cache key: training steps and the no-grad renders between them see the current weights."""
    from nerf_for_angiography_amd.render import render_rays
    torch.manual_seed(3)
    m = make_model(4, 64, precision="f16s8")
    with torch.no_grad():
        m.output_linear[0].bias.fill_(-4.0)
    o, d, tgt = _ref_iteration_problem(200, seed=5)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)
    opt = torch.optim.Adam(list(m.parameters()), lr=1e-2, fused=fused)
    with torch.no_grad():
        before = render_rays(m, o, d, 64, 1400.0, 1600.0, mode="acc").rgb_map.clone()
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(render_rays(m, o, d, 64, 1400.0, 1600.0, mode="acc").rgb_map, tgt).backward()
        opt.step()
    with torch.no_grad():
        after = render_rays(m, o, d, 64, 1400.0, 1600.0, mode="acc").rgb_map
        fresh = make_model(4, 64, precision="f16s8")
        fresh.load_state_dict(m.state_dict())
        want = render_rays(fresh, o, d, 64, 1400.0, 1600.0, mode="acc").rgb_map
    assert not torch.equal(before, after) and torch.equal(after, want)


def test_scene_box_read_back_is_cached_per_tensor_and_version():
    """ray_marching takes the scene box as a DEVICE tensor (run_nerf_acc.py:196,288); the read-back is remembered for that tensor object and
    version (one host synchronisation less per iteration), so an in-place change and a different tensor must both be seen."""
    from nerf_for_angiography_amd.nerf.occupancy import ray_marching
    o, d, _ = _ref_iteration_problem(64, seed=3)
    o, d = o.to(DEV), d.to(DEV)
    box = torch.tensor([-100.0, -100, -100, 100, 100, 100], device=DEV)
    n = lambda b: ray_marching(o, d, scene_aabb=b, near_plane=1400.0, far_plane=1600.0, render_step_size=1.0)[0].numel()
    n_full = n(box)
    assert n(box) == n_full and 64 * 150 < n_full <= 64 * 201
    box.mul_(0.5)                                     # same tensor, new version
    n_half = n(box)
    assert 0 < n_half < 0.6 * n_full
    del box
    other = torch.tensor([-25.0, -25, -25, 25, 25, 25], device=DEV)      # (may reuse the freed tensor's address)
    assert 0 < n(other) < 0.6 * n_half
    assert n([-100.0, -100, -100, 100, 100, 100]) == n_full


@pytest.mark.parametrize("layers,width,enc", [(4, 128, "none"), (8, 256, "none"), (4, 128, "barf")])
def test_one_call_grid_iteration_equals_the_call_by_call_sequence(layers, width, enc):
    """afx_march_train_step_mse / render.march_train_step_mse: the reference's grid iteration (run_nerf_acc.py:284-306) as one library call -
    the entry points of occupancy.ray_marching(return_packed=True) + render.train_step_packed_mse in the same order - gives the same
    pixels, gradients and kept count bit for bit; an empty grid skips the step; a too-small workspace grows and the call is repeated."""
    from nerf_for_angiography_amd.nerf.nerf_helpers_acc import acc_ray_marching
    from nerf_for_angiography_amd.nerf.occupancy import OccupancyGrid
    from nerf_for_angiography_amd.render import train_step_packed_mse, march_train_step_mse
    o, d, tgt = _ref_iteration_problem(1500, seed=17)
    o, d, tgt = o.to(DEV), d.to(DEV), tgt.to(DEV)
    aabb = torch.tensor([-100.0, -100, -100, 100, 100, 100], device=DEV)
    res = 64
    c = (torch.stack(torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij"), -1).float() + 0.5) / res * 200 - 100
    grid = OccupancyGrid(roi_aabb=aabb, resolution=res).to(DEV)
    grid._binary = (c.norm(dim=-1) < 55).to(DEV)

    def model():
        torch.manual_seed(8)
        m = make_model(layers, width, pos_enc=enc, precision="f16s8")
        if enc == "barf":
            m.update_barf_alpha(2.5, "pts")
        with torch.no_grad():
            m.output_linear[0].bias.fill_(-3.0)
        return m

    m1 = model()
    with torch.no_grad():
        ri, ts, te, packed = acc_ray_marching(m1, grid, aabb, o, d, 300, 1400.0, 1600.0, 1e-2, 1e-4, return_packed=True)
    loss1, pix1 = train_step_packed_mse(m1, o, d, packed, tgt)
    g1 = torch.cat([p.grad.reshape(-1) for p in m1._hip_params()])
    m2 = model()
    m2.engine._ws = None                       # start from the 64 MiB default: the 8x256 case has to grow it and run again
    loss2, pix2, kept = march_train_step_mse(m2, grid, aabb, o, d, 300, 1400.0, 1600.0, 1e-2, 1e-4, tgt)
    g2 = torch.cat([p.grad.reshape(-1) for p in m2._hip_params()])
    assert kept == ri.numel() > 1000
    assert torch.equal(pix1, pix2) and torch.equal(g1, g2) and float(loss1) == float(loss2)
    grid._binary = torch.zeros(res, res, res, dtype=torch.bool, device=DEV)
    m3 = model()
    assert march_train_step_mse(m3, grid, aabb, o, d, 300, 1400.0, 1600.0, 1e-2, 1e-4, tgt) == (None, None, 0)
    assert all(p.grad is None for p in m3._hip_params())
