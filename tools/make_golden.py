#!/usr/bin/env python3
"""Capture golden input/output vectors from the upstream reference.

Run ONCE in the build container (the only place /root/reference exists):

    cd /tmp && python /root/repo/tools/make_golden.py

It imports the reference's importable modules (model.CPPN, nerf.nerf_helpers,
phantomdata.proj_helpers), feeds them seeded inputs and writes small .npz
fixtures to tests/golden/.  Fixtures are data only (inputs + expected
outputs + the weights used); no reference source travels with the repo.
Nothing under tests/, bench.py or the package imports this script.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("AFX_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch

from model.CPPN import CPPN  # noqa: E402
from nerf import nerf_helpers as nh  # noqa: E402
from phantomdata import proj_helpers as ph  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
CPU = torch.device("cpu")


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **conv)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def model_def(layers, width, pos_enc="none", act="relu", late=0, basis=5):
    d = dict(num_early_layers=layers, num_late_layers=late, num_filters=width,
             num_input_channels=3, num_output_channels=1, num_input_channels_views=0,
             use_bias=True, pos_enc=pos_enc, pos_enc_basis=basis, act_func=act,
             fourier_sigma=5, num_img=1, device=CPU)
    if act == "sine":
        d["sine_weights"] = 15
    return d


def sd_arrays(model):
    return {"sd__" + k: v for k, v in model.state_dict().items()}


# ---- G1 pose ---------------------------------------------------------------
args, mats = [], []
for th in (0, 30, 90, 135, -45):
    for phi in (0, 30, 90, 135, -45):
        for larm in (0, 10):
            for tr in ((0, 0, 0), (-30, 10, -30)):
                args.append([th, phi, larm, *tr])
                mats.append(ph.source_matrix(np.array([0, 0, 1500.0]), th, phi, larm, list(tr)))
save("g1_pose", args=np.array(args, dtype=np.float64), src_pt=np.array([0, 0, 1500.0]),
     mats=np.array(mats))

# ---- G2 rays ---------------------------------------------------------------
g2 = {}
for tag, (w, h, f) in {"a": (64, 64, 1300.0), "b": (64, 48, 1300.0), "c": (30, 34, 400.0)}.items():
    pose = ph.source_matrix(np.array([0, 0, f + 200.0]), 35.0, -20.0, 5.0, [3.0, -2.0, 1.0])
    ii, jj = torch.meshgrid(torch.arange(0, w, dtype=torch.float64),
                            torch.arange(0, h, dtype=torch.float64), indexing="xy")
    _, o, d, z = ph.get_query_points(ii, jj, w, h, f, torch.from_numpy(pose), 8, f + 100, f + 300, CPU)
    g2.update({f"{tag}_whf": np.array([w, h, f]), f"{tag}_pose": pose, f"{tag}_o": o, f"{tag}_d": d,
               f"{tag}_z": z})
save("g2_rays", **g2)

# ---- G3 stratified depths ---------------------------------------------------
g3 = {}
for s in (32, 128):
    z = torch.linspace(0, 1, s) * 200 + 1400
    torch.manual_seed(100 + s)
    u = torch.rand(z.shape)
    torch.manual_seed(100 + s)
    out = nh.randomize_depth(z, CPU)
    g3.update({f"z{s}": z, f"u{s}": u, f"out{s}": out})
save("g3_stratify", **g3)

# ---- G4 CPPN forward --------------------------------------------------------
torch.manual_seed(7)
x_big = (torch.rand(384, 3) * 2 - 1) * 100
x_small = torch.rand(128, 3) * 2 - 1
x_all = torch.cat([x_big, x_small], 0)
cases = {
    "none_relu_4x64": (model_def(4, 64), None),
    "none_tanh_4x64": (model_def(4, 64, act="tanh"), None),
    "none_sine_4x64": (model_def(4, 64, act="sine"), None),
    "none_relu_4x64_late4": (model_def(4, 64, late=4), None),
    "fourier_relu_4x64": (model_def(4, 64, pos_enc="fourier"), None),
    "barf_relu_4x64": (model_def(4, 64, pos_enc="barf"), (0.0, 1.5, 2.5, 5.0)),
    "none_relu_4x128": (model_def(4, 128), None),
    "none_relu_8x256": (model_def(8, 256), None),
    "barf_relu_2x256": (model_def(2, 256, pos_enc="barf"), (2.5,)),
}
for i, (name, (mdef, alphas)) in enumerate(cases.items()):
    torch.manual_seed(1000 + i)
    m = CPPN(mdef)
    arrs = {"x": x_all}
    with torch.no_grad():
        if alphas is None:
            arrs["y"] = m(x_all)
        else:
            for a in alphas:
                m.update_barf_alpha(a, "pts")
                arrs[f"y_alpha{a}"] = m(x_all)
                arrs[f"w_alpha{a}"] = m.barf_weights
    arrs.update(sd_arrays(m))
    save("g4_cppn_" + name, **arrs)

# ---- G5 dense compositing ---------------------------------------------------
torch.manual_seed(11)
R, S = 48, 32
pose = ph.source_matrix(np.array([0, 0, 1500.0]), 40.0, 10.0)
ii, jj = torch.meshgrid(torch.arange(0, 8, dtype=torch.float64), torch.arange(0, 6, dtype=torch.float64),
                        indexing="xy")
_, o, d, _ = ph.get_query_points(ii * 8, jj * 8, 64, 48, 1300.0, torch.from_numpy(pose), 4, 1400, 1600, CPU)
d32 = d.reshape(-1, 3).float()
z1 = torch.linspace(0, 1, S) * 200 + 1400
z2 = torch.sort(z1[None, :] + torch.rand(R, S) * 5, -1)[0]
g5 = {"d": d32, "z1": z1, "z2": z2}
raws = {"n": torch.randn(R, S, 1), "m40": torch.full((R, S, 1), -40.0), "m3": torch.full((R, S, 1), -3.0),
        "tail": torch.cat([torch.randn(R, S - 1, 1) - 3, torch.full((R, 1, 1), -26.0)], 1),
        "c2": torch.randn(R, S, 2), "c3": torch.randn(R, S, 3)}
for rk, raw in raws.items():
    g5["raw_" + rk] = raw
    for zk, z in (("z1", z1), ("z2", z2)):
        rgb, dep, w, ent, (sig, col) = nh.render_volume_density(raw, d32, z)
        g5.update({f"{rk}_{zk}_rgb": rgb, f"{rk}_{zk}_depth": dep, f"{rk}_{zk}_weights": w,
                   f"{rk}_{zk}_entropy": ent, f"{rk}_{zk}_sigma": sig})
g5["cumprod_in"] = torch.rand(5, 9)
g5["cumprod_out"] = nh.cumprod_exclusive(g5["cumprod_in"].clone())
save("g5_render", **g5)

# ---- G7 sample_pdf ----------------------------------------------------------
g7 = {}
for tag, (r, s, nf) in {"a": (64, 32, 16), "b": (32, 128, 64)}.items():
    torch.manual_seed(70 + s)
    z = torch.linspace(0, 1, s) * 200 + 1400
    bins = (0.5 * (z[1:] + z[:-1])).repeat(r, 1)
    w = torch.rand(r, s - 2) ** 4
    w[0] = 0.25            # degenerate: all-equal weights
    w[1] = 0.0
    w[1, s // 3] = 1.0      # one-hot
    w[2] = 0.0              # all zero (only the +1e-5 floor)
    torch.manual_seed(700 + s)
    u = torch.rand(r, nf)
    torch.manual_seed(700 + s)
    out = nh.sample_pdf(bins, w, nf, CPU)
    g7.update({f"{tag}_bins": bins, f"{tag}_w": w, f"{tag}_u": u, f"{tag}_out": out})
save("g7_sample_pdf", **g7)

# ---- G8 end-to-end C1 (4 projections 64x64, 32 samples/ray, 4x64 MLP) -------
W = H = 64
F = 13.0 * W
NEAR, FAR, S = 1400.0, 1600.0, 32
torch.manual_seed(21)
os_, ds_ = [], []
for th, phi in ((0, 0), (0, 90), (90, 0), (90, 90)):
    pose = ph.source_matrix(np.array([0, 0, 1500.0]), th, phi)
    ii, jj = torch.meshgrid(torch.arange(0, W, dtype=torch.float64), torch.arange(0, H, dtype=torch.float64),
                            indexing="xy")
    _, o, d, _ = ph.get_query_points(ii, jj, W, H, F, torch.from_numpy(pose), S, NEAR, FAR, CPU)
    pick = torch.randperm(W * H)[:256]
    os_.append(o.reshape(-1, 3)[pick].float())
    ds_.append(d.reshape(-1, 3)[pick].float())
o = torch.cat(os_)
d = torch.cat(ds_)
target = torch.rand(o.shape[0])
z = torch.linspace(0., 1., S)
z = NEAR * (1. - z) + FAR * z
torch.manual_seed(22)
m = CPPN(model_def(4, 64))
with torch.no_grad():   # make densities non-trivial at raw world coordinates
    m.output_linear[0].weight.mul_(8.0)
    m.output_linear[0].bias.fill_(-6.0)
g8 = {"o": o, "d": d, "target": target, "z": z, "near_far_s": np.array([NEAR, FAR, S])}
g8.update({"init__" + k: v.clone() for k, v in m.state_dict().items()})

# (a) dense convention, literally the reference: points -> get_predictions -> render_volume_density
pts = (o[:, None, :] + d[:, None, :] * z[None, :, None]).reshape(-1, 3).float()
raw = nh.get_predictions(m, pts, 8192).reshape(o.shape[0], S, 1)
rgb, dep, wts, ent, _ = nh.render_volume_density(raw, d, z)
loss = torch.nn.functional.mse_loss(rgb, target)
m.zero_grad()
loss.backward()
g8.update({"dense_rgb": rgb, "dense_weights": wts, "dense_depth": dep, "dense_entropy": ent,
           "dense_loss": loss, "dense_raw": raw})
g8.update({"dense_grad__" + k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})

# (a') same, with the output bias at -26 so that the 1e10 tail term neither
# underflows to 0 nor saturates at 1: pixels and gradients are non-trivial.
with torch.no_grad():
    m.output_linear[0].bias.fill_(-26.0)
raw = nh.get_predictions(m, pts, 8192).reshape(o.shape[0], S, 1)
rgb, dep, wts, ent, _ = nh.render_volume_density(raw, d, z)
loss = torch.nn.functional.mse_loss(rgb, target)
m.zero_grad()
loss.backward()
g8.update({"dense26_rgb": rgb, "dense26_weights": wts, "dense26_loss": loss})
g8.update({"dense26_grad__" + k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
with torch.no_grad():
    m.output_linear[0].bias.fill_(-6.0)

# (b) acc convention: reference CPPN + get_predictions; compositing is the 4-line
# restatement of nerf_helpers_acc.py:46-58 for a dense packed batch (scatter_mul over a
# fixed S per ray == prod over the sample axis).  torch_scatter/nerfacc are absent here.
step = (FAR - NEAR) / S
ts = NEAR + torch.arange(S, dtype=torch.float32) * np.float32(step)
te = ts + np.float32(step)


def acc_pixels(model):
    ri = torch.arange(o.shape[0]).repeat_interleave(S)
    tsf, tef = ts.repeat(o.shape[0])[:, None], te.repeat(o.shape[0])[:, None]
    pos = o[ri] + d[ri] * (tsf + tef) / 2.0
    pred = nh.get_predictions(model, pos, 8192)
    alphas = torch.exp(-torch.sigmoid(pred) * (tef - tsf))
    return alphas.view(o.shape[0], S).prod(-1)


m.zero_grad()
pix = acc_pixels(m)
loss = torch.nn.functional.mse_loss(pix, target)
loss.backward()
g8.update({"acc_rgb": pix, "acc_loss": loss})
g8.update({"acc_grad__" + k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
opt = torch.optim.Adam(list(m.parameters()), lr=1e-4)
for it in range(10):
    opt.zero_grad()
    loss = torch.nn.functional.mse_loss(acc_pixels(m), target)
    loss.backward()
    opt.step()
    for gparam in opt.param_groups:
        gparam["lr"] = 1e-4 * (0.1 ** (it / 500000))
    if it in (0, 9):
        g8.update({f"acc_step{it + 1}__" + k: v.clone() for k, v in m.state_dict().items()})
        g8[f"acc_step{it + 1}_loss"] = loss.detach()
save("g8_e2e_c1", **g8)

# ---- G9 density grid --------------------------------------------------------
torch.manual_seed(31)
m = CPPN(model_def(4, 64))
t = np.linspace(-100, 100, 17)
qp = np.stack(np.meshgrid(t, t, t), -1).astype(np.float32)
flat = torch.from_numpy(qp.reshape(-1, 3))
with torch.no_grad():
    sig = torch.sigmoid(nh.get_predictions(m, flat, 1024)).reshape(17, 17, 17)
arrs = {"t": t, "points": flat, "sigma": sig}
arrs.update(sd_arrays(m))
save("g9_density_grid", **arrs)
print("done")
