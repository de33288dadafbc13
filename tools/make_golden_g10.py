#!/usr/bin/env python3
"""Capture fixtures G10 (ray_tracing on a voxelised analytic phantom) and G11 (transfer_func_ct, rev_sigmoid, line)
from the upstream reference's phantomdata/helpers.py.

CAPTURED WITH A FRANGI STAND-IN: that module's only missing import in this container is `skimage.filters.frangi`,
which is used by ONE function outside the hot path (`get_weighted_img`); an in-process stub module that raises when
called lets the rest of the file import unchanged.  None of the functions captured here touches it.

Run ONCE in the build container (the only place /root/reference exists):

    cd /tmp && python /root/repo/tools/make_golden_g10.py
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REF = os.environ.get("AFX_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch
from scipy.interpolate import RegularGridInterpolator


def _no_frangi(*a, **k):
    raise NotImplementedError("skimage is not installed here; get_weighted_img is out of scope")


for name in ("skimage", "skimage.filters"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["skimage.filters"].frangi = _no_frangi
sys.modules["skimage"].filters = sys.modules["skimage.filters"]

from phantomdata import helpers as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# ---- G11 transfer function / reverse sigmoid ------------------------------------------------------------------
rng = np.random.default_rng(11)
knots = np.array([-50.0, 0.0, 1.0, 752.99, 753.0, 1000.0, 1585.85, 2000.0, 2332.9, 3000.0, 3306.18, 3999.9, 4000.0, 5000.0])
vals = np.concatenate([knots, rng.uniform(-200, 4500, 200)])
g11 = dict(vals=vals, tf=rh.transfer_func_ct(vals), tf_binary=rh.transfer_func_ct(vals, binary=True),
           x=np.linspace(-6, 6, 49), )
g11["rev_sigmoid_c1_2"] = rh.rev_sigmoid(g11["x"], c1=2)
g11["rev_sigmoid_default"] = rh.rev_sigmoid(g11["x"])
np.savez(os.path.join(OUT, "g11_transfer.npz"), **g11)
print("g11_transfer:", {k: v.shape for k, v in g11.items()})

# ---- G10 ray_tracing on a 41^3 voxelised analytic phantom (sphere + capsule), 64 x 48 detector --------------------
n = 41
ax = np.linspace(-60.0, 60.0, n)
X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
sphere = np.sqrt((X - 10) ** 2 + (Y + 5) ** 2 + Z ** 2) - 25.0
a, b = np.array([-30.0, -20.0, -10.0]), np.array([25.0, 30.0, 20.0])
P = np.stack([X, Y, Z], -1)
t = np.clip(((P - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
capsule = np.linalg.norm(P - (a + t[..., None] * (b - a)), axis=-1) - 6.0
sdf = np.minimum(sphere, capsule)
mu = (rh.rev_sigmoid(sdf, c1=2) * 0.05).astype(np.float64)            # helpers.py:17-18,93: transfer of an SDF volume
interp = RegularGridInterpolator((ax, ax, ax), mu, method="linear", bounds_error=False, fill_value=float(mu.min()))   # helpers.py:98,152
W, H, f = 64, 48, 13.0 * 64
src = np.array([0.0, 0.0, 1500.0])
res = {}
dev = torch.device("cpu")
for tag, (theta, phi, larm) in {"a": (30.0, 10.0, 0.0), "b": (100.0, -35.0, 5.0)}.items():
    o, d, M, ii, jj = rh.get_ray_values(theta, phi, larm, src, W, H, f, dev)
    torch.manual_seed(0)
    z = rh.get_depth_values(1400.0, 1600.0, 96, dev, stratified=False)
    with tempfile.TemporaryDirectory() as tmp:
        img_ct = rh.ray_tracing(interp, (theta, phi, larm), o, d, z, W, H, ii, jj, 32, dev, tmp + "/", type="ct")
        img_sdf = rh.ray_tracing(interp, (theta, phi, larm), o, d, z, W, H, ii, jj, 32, dev, tmp + "/", type="sdf")
    res.update({f"{tag}_pose": M, f"{tag}_angles": np.array([theta, phi, larm]), f"{tag}_z": z.numpy(),
                f"{tag}_img_ct": img_ct.numpy(), f"{tag}_img_sdf": img_sdf.numpy(),
                f"{tag}_o": o.numpy().astype(np.float32), f"{tag}_d": d.numpy().astype(np.float32)})
res.update(axis=ax, mu=mu.astype(np.float32), fill=np.array(float(mu.min())), whf=np.array([W, H, f]))
np.savez_compressed(os.path.join(OUT, "g10_ray_tracing.npz"), **res)
print("g10_ray_tracing:", {k: np.asarray(v).shape for k, v in res.items()}, "img range", float(res["a_img_ct"].min()), float(res["a_img_ct"].max()))
