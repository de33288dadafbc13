"""CPU oracle for the NeRF-for-angiography hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  The product
(`nerf_for_angiography_amd`) never imports it and never falls back to it.

It is a plain PyTorch-CPU restatement (own code, own structure) of the
reference algorithm.  Every function cites the reference file:line (relative
to the upstream repository root) it follows.  Pinning status: every function
marked PINNED is checked against golden vectors captured from the reference
itself (``tools/make_golden.py`` -> ``tests/golden/*.npz``,
``tests/test_oracle_golden.py``).  Functions marked UNPINNED restate code
whose third-party dependencies (nerfacc 0.3.x, torch_scatter 2.x, neither
pinned nor vendored by the reference) are absent, so no reference output
could be captured for them: "parity unpinned" for exactly those.

All maths is float32 unless a dtype argument says otherwise (the reference
runs float32 on device, float64 only for pose / ray generation).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# R1  C-arm pose.  phantomdata/proj_helpers.py:34-77   (PINNED, G1)
# --------------------------------------------------------------------------

def _rot(axis: str, deg: float) -> np.ndarray:
    """4x4 homogeneous rotation about one axis (proj_helpers.py:34-56)."""
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    m = np.identity(4)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[axis]
    m[i, i] = c
    m[j, j] = c
    m[i, j] = -s
    m[j, i] = s
    return m


def _trans(v: Sequence[float]) -> np.ndarray:
    """4x4 translation (proj_helpers.py:58-61)."""
    m = np.identity(4)
    m[:3, 3] = np.asarray(v, dtype=np.float64)[:3]
    return m


def source_matrix(src_pt, theta, phi, larm=0.0, translation=(0.0, 0.0, 0.0)) -> np.ndarray:
    """M = T(translation) . inv(Rz(larm) Rx(theta) Ry(phi)) . T(src_pt), float64.

    proj_helpers.py:63-77.  Used by the reference as the camera->world matrix.
    """
    r = np.linalg.inv(_rot("z", larm).dot(_rot("x", theta).dot(_rot("y", phi))))
    return _trans(translation).dot(r.dot(_trans(src_pt)))


# --------------------------------------------------------------------------
# R2  ray generation ("get_rays").  phantomdata/helpers.py:156-175 ==
#     phantomdata/proj_helpers.py:9-16                       (PINNED, G2)
# --------------------------------------------------------------------------

def get_rays(pose: np.ndarray, width: int, height: int, focal: float,
             dtype=torch.float64) -> Tuple[Tensor, Tensor]:
    """Origins/directions for the full [H, W] pixel grid; d is NOT normalised.

    Pixel (row jj, col ii): dir_cam = ((ii - W/2)/f, -(jj - H/2)/f, -1),
    d = R dir_cam (written in the reference as a broadcast product summed over
    the last axis), o = pose[:3, 3].
    """
    m = torch.as_tensor(pose, dtype=dtype)
    ii = torch.arange(width, dtype=dtype)[None, :].expand(height, width)
    jj = torch.arange(height, dtype=dtype)[:, None].expand(height, width)
    dcam = torch.stack([(ii - width / 2) / focal, -(jj - height / 2) / focal,
                        -torch.ones_like(ii)], dim=-1)
    d = (dcam[..., None, :] * m[:3, :3]).sum(-1)
    o = m[:3, 3].expand(d.shape)
    return o, d


# --------------------------------------------------------------------------
# R3/R4  depths.  nerf/run_nerf_acc.py:131-139, nerf/nerf_helpers.py:13-22
#                                                             (PINNED, G3)
# --------------------------------------------------------------------------

def depth_values(near: float, far: float, n: int) -> Tensor:
    t = torch.linspace(0.0, 1.0, n)
    return near * (1.0 - t) + far * t


def stratify(z: Tensor, u: Tensor) -> Tensor:
    """randomize_depth with the uniform draw `u` supplied (same shape as z)."""
    mid = 0.5 * (z[..., 1:] + z[..., :-1])
    hi = torch.cat([mid, z[..., -1:]], -1)
    lo = torch.cat([z[..., :1], mid], -1)
    return lo + (hi - lo) * u


# --------------------------------------------------------------------------
# R5  points.  nerf_helpers.py:187 / proj_helpers.py:30 (dense),
#              run_nerf_acc.py:290-292 (acc mid-points)
# --------------------------------------------------------------------------

def points_dense(o: Tensor, d: Tensor, z: Tensor) -> Tensor:
    """[R,3],[R,3],[S]|[R,S] -> [R,S,3] ; p = o + d*z."""
    return o[..., None, :] + d[..., None, :] * z[..., :, None]


def march_uniform(near: float, far: float, n: int, n_rays: int):
    """Dense no-grid march of nerf_helpers_acc.py:27 (UNPINNED: nerfacc absent).

    step=(far-near)/n, t_s = near+i*step, t_e = t_s+step, no pruning.
    Returns ray_indices[n_rays*n], t_starts[n_rays*n,1], t_ends[n_rays*n,1].
    """
    step = np.float32((far - near) / n)
    i = torch.arange(n, dtype=torch.float32)
    ts = np.float32(near) + i * step
    te = ts + step
    ri = torch.arange(n_rays).repeat_interleave(n)
    return ri, ts.repeat(n_rays)[:, None], te.repeat(n_rays)[:, None]


def points_acc(o: Tensor, d: Tensor, ri: Tensor, ts: Tensor, te: Tensor) -> Tensor:
    return o[ri] + d[ri] * (ts + te) / 2.0


# --------------------------------------------------------------------------
# R7  CPPN.  model/CPPN.py:10-259                              (PINNED, G4)
# --------------------------------------------------------------------------

def barf_weights(alpha: float, n_freq: int, n_in: int = 3) -> Tensor:
    """CPPN.barf_coefficients (CPPN.py:244-259), literal incl. the 3.1415 and
    the (alpha - k + 1) argument."""
    ks = torch.repeat_interleave(torch.arange(0.0, n_freq), n_in)
    out = []
    for k in ks:
        g = alpha - (k + 1)
        if g < 0:
            out.append(0.0)
        elif g < 1:
            out.append(float((1 - torch.cos((alpha - k + 1) * 3.1415)) / 2))
        else:
            out.append(1.0)
    return torch.tensor(out, dtype=torch.float32)


def encode(x: Tensor, cfg: dict, params: Dict[str, Tensor]) -> Tensor:
    """CPPN.pos_enc (CPPN.py:207-234): [x, w*sin(v), w*cos(v)] or identity."""
    kind = cfg.get("pos_enc", "none")
    n_freq = cfg.get("pos_enc_basis", 0)
    if kind == "none" or n_freq <= 0:
        return x
    n_in = x.shape[-1]
    basis = torch.cat(n_freq * [x], dim=-1)
    if kind == "barf":
        ks = torch.repeat_interleave(torch.arange(0.0, n_freq), n_in)
        freq = torch.Tensor((2 ** ks * np.pi))  # float32, as CPPN.py:85
        v = freq * basis
        w = params["barf_weights"] if "barf_weights" in params else barf_weights(
            cfg.get("barf_alpha", 0.0), n_freq, n_in)
        return torch.cat([x, w * torch.sin(v), w * torch.cos(v)], dim=-1)
    if kind == "fourier":
        v = 2 * np.pi * basis * params["fourier_coefficients"]
        return torch.cat([x, torch.sin(v), torch.cos(v)], dim=-1)
    raise ValueError(kind)


def _act(name: str, first: bool, cfg: dict):
    if name == "relu":
        return torch.relu
    if name == "tanh":
        return torch.tanh
    if name == "sine":
        w0 = float(cfg["sine_weights"]) if first else 1.0
        return lambda t: torch.sin(w0 * t)
    raise ValueError(name)


def cppn_forward(x: Tensor, cfg: dict, params: Dict[str, Tensor]) -> Tensor:
    """CPPN.forward for the no-view-direction configuration (CPPN.py:166-205).

    `params` uses the reference state-dict key names:
    early_pts_layers.{0,2,..}.{weight,bias}, skip_connection.0.*,
    late_pts_layers.{0,2,..}.*, output_linear.0.*.
    """
    e = encode(x, cfg, params)
    h = e
    n_early = cfg["num_early_layers"]
    for i in range(n_early + 1):
        w = params[f"early_pts_layers.{2 * i}.weight"]
        b = params.get(f"early_pts_layers.{2 * i}.bias")
        h = torch.nn.functional.linear(h, w, b)
        h = _act(cfg.get("act_func", "relu"), i == 0, cfg)(h)
    n_late = cfg.get("num_late_layers", 0)
    if n_late > 0:
        act = _act(cfg.get("act_func", "relu"), False, cfg)
        h = act(torch.nn.functional.linear(
            torch.cat([e, h], -1), params["skip_connection.0.weight"],
            params.get("skip_connection.0.bias")))
        for i in range(n_late - 1):
            h = act(torch.nn.functional.linear(
                h, params[f"late_pts_layers.{2 * i}.weight"],
                params.get(f"late_pts_layers.{2 * i}.bias")))
    return torch.nn.functional.linear(h, params["output_linear.0.weight"],
                                      params.get("output_linear.0.bias"))


def init_params(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    """nn.Linear-default initialised parameters with the reference layout.
    (Own recipe, used for synthetic runs; golden tests use captured weights.)"""
    g = torch.Generator().manual_seed(seed)
    f = cfg["num_filters"]
    n_in = cfg.get("num_input_channels", 3)
    if cfg.get("pos_enc", "none") != "none":
        n_in = n_in + 2 * n_in * cfg["pos_enc_basis"]
    p: Dict[str, Tensor] = {}

    def lin(name, fan_out, fan_in):
        bound = 1.0 / math.sqrt(fan_in)
        p[name + ".weight"] = (torch.rand(fan_out, fan_in, generator=g) * 2 - 1) * bound
        p[name + ".bias"] = (torch.rand(fan_out, generator=g) * 2 - 1) * bound

    lin("early_pts_layers.0", f, n_in)
    for i in range(cfg["num_early_layers"]):
        lin(f"early_pts_layers.{2 * (i + 1)}", f, f)
    lin("output_linear.0", cfg.get("num_output_channels", 1), f)
    if cfg.get("pos_enc") == "fourier":
        p["fourier_coefficients"] = torch.randn(3 * cfg["pos_enc_basis"], generator=g) * cfg["fourier_sigma"]
    return p


# --------------------------------------------------------------------------
# R6  chunked evaluation.  nerf/nerf_helpers.py:24-45
# --------------------------------------------------------------------------

def get_predictions(fn, pts: Tensor, chunk: int) -> Tensor:
    return torch.cat([fn(pts[i:i + chunk]) for i in range(0, pts.shape[0], chunk)], 0)


# --------------------------------------------------------------------------
# R9  dense compositing ("raw2outputs").  nerf/nerf_helpers.py:47-135
#                                                              (PINNED, G5)
# --------------------------------------------------------------------------

def cumprod_exclusive(t: Tensor) -> Tensor:
    c = torch.cumprod(t, -1)
    return torch.cat([torch.ones_like(c[..., :1]), c[..., :-1]], -1)


def ray_entropy(sigma: Tensor, rgb_map: Tensor, threshold: float = 0.4) -> Tensor:
    """nerf_helpers.py:125-135."""
    dens = sigma / (sigma.sum(-1).unsqueeze(1) + 1e-10)
    ent = -(dens * torch.log(dens + 1e-10)).sum(-1)
    return ent * ((1 - rgb_map) > threshold).detach()


def render_volume_density(raw: Tensor, d: Tensor, z: Tensor):
    """raw[R,S,C], d[R,3], z[S]|[R,S] -> (rgb_map, depth_map, weights, entropy,
    [sigma, rgb]).  Includes the 1e10 last distance, the ||d||
    scaling (D4) and depth_map = sum(alpha*z) (D5) exactly as the reference."""
    big = torch.tensor([1e10], dtype=d.dtype).expand(z[..., :1].shape)
    dists = torch.cat((z[..., 1:] - z[..., :-1], big), -1)
    nd = dists * torch.norm(d[..., None, :], dim=-1)
    c = raw.shape[-1]
    if c == 2:
        sigma = torch.relu(raw[..., -1])
        rgb = torch.sigmoid(raw[..., :-1])
        alpha = 1.0 - torch.exp(-sigma * dists)
        weights = alpha * cumprod_exclusive(1.0 - alpha + 1e-10)
        rgb_map = torch.squeeze((weights[..., None] * rgb).sum(-2))
        depth_map = (weights * z).sum(-1)
        asum = alpha.sum(-1)
        dens = alpha / (asum.unsqueeze(-1) + 1e-10)
        ent = -(dens * torch.log(dens + 1e-10)).sum(-1) * (asum > 0.7).detach()
        return rgb_map, depth_map, weights, ent.mean(), [sigma, rgb]
    sigma = torch.relu(raw.mean(-1)) if c > 1 else torch.sigmoid(raw[..., -1])
    rgb = torch.ones(sigma.shape[0], sigma.shape[1], 1)
    alpha = torch.exp(-sigma * nd)
    weights = (1 - alpha + 1e-10) * cumprod_exclusive(alpha)
    rgb_map = alpha.prod(-1)
    depth_map = (alpha * z).sum(-1)
    return rgb_map, depth_map, weights, ray_entropy(sigma, rgb_map), [sigma, rgb]


# --------------------------------------------------------------------------
# R8  the compositing that trains.  nerf/nerf_helpers_acc.py:45-63
#     (UNPINNED: torch_scatter absent; scatter_mul over ray index ==
#      index_reduce 'prod' into ones)
# --------------------------------------------------------------------------

def acc_render_volume_density(pred: Tensor, ri: Tensor, ts: Tensor, te: Tensor,
                              n_rays: int, zero_idx=()) -> Tensor:
    sig = torch.sigmoid(pred)
    if len(zero_idx) > 0:
        sig = sig.clone()
        sig[zero_idx] = 0
    alpha = torch.exp(-sig * (te - ts))
    out = torch.ones(n_rays, alpha.shape[-1], dtype=alpha.dtype)
    out = out.index_reduce(0, ri.long(), alpha, "prod", include_self=True)
    return out.squeeze(-1).float()


# --------------------------------------------------------------------------
# R10  hierarchical sampling.  nerf/nerf_helpers.py:178-222     (PINNED, G7)
# --------------------------------------------------------------------------

def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    """Inverse-CDF sampling with the uniform draw `u[R,N_f]` supplied."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    idx = torch.searchsorted(cdf, u.contiguous(), right=True)
    lo = (idx - 1).clamp(min=0)
    hi = idx.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def fine_depths(z: Tensor, weights_coarse: Tensor, u: Tensor, n_rays: int) -> Tensor:
    """Depth part of fine_sampling (nerf_helpers.py:179-186): bins = coarse
    mid-points, weights = coarse weights[...,1:-1], merged and sorted."""
    zz = z.repeat(n_rays, 1) if z.dim() == 1 else z
    mid = 0.5 * (zz[..., 1:] + zz[..., :-1])
    s = sample_pdf(mid, weights_coarse[..., 1:-1], u)
    return torch.sort(torch.cat([zz, s.detach()], -1), -1)[0]


# --------------------------------------------------------------------------
# R14  ground-truth projector.  phantomdata/helpers.py:192-224
#      (restated for an analytic or callable mu(p); formula == R9's rgb_map
#       with mu in place of sigmoid(raw))
# --------------------------------------------------------------------------

def project_mu(mu_fn, o: Tensor, d: Tensor, z: Tensor) -> Tensor:
    big = torch.tensor([1e10], dtype=z.dtype)
    dists = torch.cat((z[1:] - z[:-1], big), -1)
    p = points_dense(o, d, z)
    mu = mu_fn(p.reshape(-1, 3)).reshape(p.shape[:-1]).to(z.dtype)
    nd = dists * torch.norm(d[..., None, :], dim=-1)
    return torch.exp(-mu * nd).prod(-1)


# --------------------------------------------------------------------------
# R15  density grid.  visualization/visualization.py:100-102,209-229,
#      visualization/helpers.py:21-45                           (PINNED, G9)
# --------------------------------------------------------------------------

def density_grid_points(outside: float, n: int) -> Tensor:
    """meshgrid(t,t,t) with numpy's default 'xy' indexing:
    point[i,j,k] = (t[j], t[i], t[k]); returned flattened [(n+1)^3, 3]."""
    t = np.linspace(-outside, outside, n + 1)
    gx, gy, gz = np.meshgrid(t, t, t)
    return torch.from_numpy(np.stack([gx.flatten(), gy.flatten(), gz.flatten()], 1)).float()


def density_grid(fn, outside: float, n: int, chunk: int = 16384) -> Tensor:
    return torch.sigmoid(get_predictions(fn, density_grid_points(outside, n), chunk)).reshape(
        n + 1, n + 1, n + 1)


# --------------------------------------------------------------------------
# R11/R12  loss, PSNR, one training step.  nerf/run_nerf_acc.py:289-328
# --------------------------------------------------------------------------

def psnr(mse: Tensor) -> Tensor:
    return -10.0 * torch.log10(mse)


def render_rays(o: Tensor, d: Tensor, cfg: dict, params: Dict[str, Tensor], *,
                near: float, far: float, n_samples: int, convention: str,
                z: Optional[Tensor] = None, chunk: int = 131072) -> Tensor:
    """Pixel values for a ray batch in one of the two compositing conventions.

    convention='acc'   : uniform mid-point march + R8 (what run_nerf_acc.py runs)
    convention='dense' : z (given, or linspace) + R9 (1e10 tail, ||d|| scaling)
    """
    fn = lambda p: cppn_forward(p, cfg, params)
    r = o.shape[0]
    if convention == "acc":
        ri, ts, te = march_uniform(near, far, n_samples, r)
        pred = get_predictions(fn, points_acc(o, d, ri, ts, te), chunk)
        return acc_render_volume_density(pred, ri, ts, te, r)
    zz = depth_values(near, far, n_samples) if z is None else z
    pts = points_dense(o, d, zz).reshape(-1, 3).float()
    raw = get_predictions(fn, pts, chunk).reshape(r, -1, 1)
    return render_volume_density(raw, d, zz)[0]


def loss_and_grads(o, d, target, cfg, params, **kw):
    """MSE loss (mean over rays) and gradients w.r.t. every tensor in `params`
    that the forward touches; autograd on CPU."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()
              if v.dtype.is_floating_point and k != "barf_weights"}
    full = dict(params)
    full.update(leaves)
    pix = render_rays(o, d, cfg, full, **kw)
    loss = torch.nn.functional.mse_loss(pix, target)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    return pix.detach(), loss.detach(), {k: g for k, g in zip(leaves, grads) if g is not None}


def lr_at(step: int, lr0: float = 1e-4, decay_rate: float = 0.1, decay_steps: int = 500000) -> float:
    """run_nerf_acc.py:323-328."""
    return lr0 * decay_rate ** (step / decay_steps)


# --------------------------------------------------------------------------
# Synthetic analytic phantoms (own construction, not reference
# code).  mu(p) for a union of capsules, used by tests, smoke and bench.
# --------------------------------------------------------------------------

def capsule_tree(levels: int = 5, seed: int = 0, extent: float = 75.0,
                 r0: float = 3.0, r1: float = 0.75):
    """Binary tree of 2^levels-1 capsules inside +-extent. Returns [N,7]:
    (ax,ay,az,bx,by,bz,radius)."""
    rng = np.random.RandomState(seed)
    segs = []
    front = [(np.array([0.0, -0.8 * extent, 0.0]), np.array([0.0, 1.0, 0.0]), 0)]
    n_total = 2 ** levels - 1
    while front and len(segs) < n_total:
        a, dirv, lev = front.pop(0)
        length = 0.55 * extent * (0.72 ** lev)
        b = np.clip(a + dirv * length, -0.95 * extent, 0.95 * extent)
        rad = r0 + (r1 - r0) * lev / max(levels - 1, 1)
        segs.append(np.concatenate([a, b, [rad]]))
        for sgn in (-1.0, 1.0):
            pert = rng.normal(size=3) * 0.35
            nd = dirv + sgn * np.cross(dirv, [0.3, 0.2, 1.0]) * 0.8 + pert
            front.append((b, nd / np.linalg.norm(nd), lev + 1))
    return np.asarray(segs, dtype=np.float32)


def capsule_mu(p: Tensor, caps: np.ndarray, mu: float = 0.2) -> Tensor:
    """Binary attenuation: mu inside any capsule, 0 outside."""
    c = torch.as_tensor(caps, dtype=p.dtype)
    a, b, r = c[:, 0:3], c[:, 3:6], c[:, 6]
    ab = b - a
    inside = torch.zeros(p.shape[0], dtype=torch.bool)
    for i in range(c.shape[0]):
        t = ((p - a[i]) @ ab[i] / (ab[i] @ ab[i])).clamp(0, 1)
        dist = torch.norm(p - (a[i] + t[:, None] * ab[i]), dim=-1)
        inside |= dist <= r[i]
    return inside.to(p.dtype) * mu


# --------------------------------------------------------------------------
# R14 over a voxel volume.  phantomdata/helpers.py:72-154 (interpolator) + :192-224 (ray_tracing).
# Literal: scipy RegularGridInterpolator(method='linear', bounds_error=False, fill_value=min), float64 points,
# weights = exp(-mu * dists * ||d||) ('ct') or exp(-mu), product over samples.  (scipy is the reference's own
# dependency and is present, so this IS the reference algorithm, not a restatement of it.)
# --------------------------------------------------------------------------

def project_volume_scipy(axes, values, o: Tensor, d: Tensor, z: Tensor, type_ct: bool = True, fill_value=None):
    from scipy.interpolate import RegularGridInterpolator
    fv = float(np.min(values)) if fill_value is None else fill_value
    interp = RegularGridInterpolator(tuple(np.asarray(a, dtype=np.float64) for a in axes), np.asarray(values, dtype=np.float64),
                                     method="linear", bounds_error=False, fill_value=fv)
    o64, d64 = o.double(), d.double()
    pts = o64[..., None, :] + d64[..., None, :] * z[..., :, None]
    big = torch.tensor([1e10], dtype=z.dtype)
    dists = torch.cat((z[1:] - z[:-1], big), -1)
    mu = torch.from_numpy(interp(pts.numpy().reshape(-1, 3))).reshape(pts.shape[:-1])
    if type_ct:
        w = torch.exp(-mu * (dists * torch.norm(d64[..., None, :], dim=-1)))
    else:
        w = torch.exp(-mu)
    return w.prod(-1).float()


# --------------------------------------------------------------------------
# transfer functions of the phantom generators.  phantomdata/helpers.py:17-18 (rev_sigmoid), :20-31 (line),
# :33-70 (transfer_func_ct)                                           (PINNED, G11; captured with a frangi stand-in)
# --------------------------------------------------------------------------

def rev_sigmoid(x, c1: float = 1.0, c2: float = 0.0):
    return 1.0 / (1.0 + np.exp(c1 * (np.asarray(x, dtype=np.float64) - c2)))


def transfer_func_ct(vals, binary: bool = False):
    """Six knots, straight lines between them written as m*x + b (helpers.py:20-31), constants outside."""
    xs = [0.0, 753.0, 1585.85, 2332.9, 3306.18, 4000.0]
    ys = [0.0, 0.0, 0.0 if binary else 0.05, 0.0, 0.2, 0.4]
    v = np.asarray(vals, dtype=np.float64)
    out = np.where(v < xs[0], ys[0], np.where(v >= xs[-1], ys[-1], 0.0))
    for (x1, y1), (x2, y2) in zip(zip(xs[:-1], ys[:-1]), zip(xs[1:], ys[1:])):
        m, b0 = (y1 - y2) / (x1 - x2), (x1 * y2 - x2 * y1) / (x1 - x2)
        seg = (v >= x1) & (v < x2)
        out = np.where(seg, m * v + b0, out)
    return out


# --------------------------------------------------------------------------
# K1/K2  occupancy grid + grid-skipping march.  nerf/run_nerf_acc.py:196-198,284-287,
#        nerf/nerf_helpers_acc.py:10-31,65-78 -> nerfacc 0.3.x (OccupancyGrid, ray_marching,
#        render_visibility).  nerfacc is absent (not vendored, not pinned): this restates its
#        PUBLISHED algorithm.                                             (UNPINNED)
# --------------------------------------------------------------------------

def grid_cell_index(pts: Tensor, aabb: Tensor, res: Sequence[int]):
    """(flat cell index, inside) of world points for an AABB-contracted grid."""
    lo, hi = aabb[:3], aabb[3:]
    u = (pts - lo) / (hi - lo)
    inside = ((u >= 0) & (u < 1)).all(-1)
    r = torch.tensor(list(res), dtype=torch.float32)
    ijk = torch.minimum((u * r).floor().long().clamp(min=0), (r - 1).long())
    return (ijk[:, 0] * int(res[1]) + ijk[:, 1]) * int(res[2]) + ijk[:, 2], inside


def grid_jittered_points(cells: Tensor, jitter: Tensor, aabb: Tensor, res: Sequence[int]) -> Tensor:
    """OccupancyGrid._update: x = (coords + u) / resolution, un-contracted to world space."""
    r = [int(v) for v in res]
    k = cells % r[2]
    j = (cells // r[2]) % r[1]
    i = cells // (r[2] * r[1])
    coords = torch.stack([i, j, k], -1).float()
    x = (coords + jitter) / torch.tensor(r, dtype=torch.float32)
    return x * (aabb[3:] - aabb[:3]) + aabb[:3]


def grid_update(occs: Tensor, cells: Tensor, occ_new: Tensor, ema_decay: float, occ_thre: float):
    """occs[c] = max(occs[c] * decay, occ) (a cell drawn twice: max over its draws); binary = occs > min(mean, thre)."""
    out = occs.clone()
    dec = occs * ema_decay
    out[cells] = dec[cells]
    out = out.index_reduce(0, cells, occ_new.reshape(-1).clamp(min=0), "amax", include_self=True)
    thr = min(float(out.double().mean()), occ_thre)
    return out, out > thr


def ray_aabb(o: Tensor, d: Tensor, aabb: Tensor):
    inv = 1.0 / torch.where(d == 0, torch.full_like(d, 1e-12), d)
    t0, t1 = (aabb[:3] - o) * inv, (aabb[3:] - o) * inv
    tmin = torch.minimum(t0, t1).amax(-1)
    tmax = torch.maximum(t0, t1).amin(-1)
    miss = tmax < torch.clamp(tmin, min=0)
    return torch.where(miss, torch.full_like(tmin, 1e10), torch.clamp(tmin, min=0)), torch.where(miss, torch.full_like(tmax, 1e10), tmax)


def march_grid(o: Tensor, d: Tensor, scene_aabb: Optional[Tensor], near: Optional[float], far: Optional[float], dt: float,
               binary: Optional[Tensor] = None, grid_aabb: Optional[Tensor] = None):
    """Fixed-step lattice t_min + k dt, steps whose mid-point lies in [t_min, t_max); a step is kept when the cell of its mid-point is occupied.
    -> packed ray-sorted (ray_indices, t_starts, t_ends), plain Python loops (small cases only)."""
    n = o.shape[0]
    if scene_aabb is not None:
        tmin, tmax = ray_aabb(o, d, scene_aabb)
    else:
        tmin, tmax = torch.zeros(n), torch.full((n,), 1e10)
    if near is not None:
        tmin = torch.clamp(tmin, min=near)
    if far is not None:
        tmax = torch.clamp(tmax, max=far)
    dtf = torch.tensor(dt, dtype=torch.float32)
    ri, ts, te = [], [], []
    for r in range(n):
        if float(tmin[r]) >= 1e10:
            continue
        # nerfacc 0.3.x marches `while (t_mid < far)`: a step is the ray's when its mid-point lies before t_max
        ns = int(max(0.0, math.ceil(float((tmax[r] - tmin[r]) / dtf)))) + 2
        k = torch.arange(ns, dtype=torch.float32)
        s = tmin[r] + k * dtf
        e = s + dtf
        inside_range = ((s + e) * 0.5) < tmax[r]
        ns = int(inside_range.sum())                 # mid-points are monotone: a prefix
        s, e = s[:ns], e[:ns]
        keep = torch.ones(ns, dtype=torch.bool)
        if binary is not None and ns > 0:
            mid = o[r][None, :] + d[r][None, :] * ((s + e) * 0.5)[:, None]
            idx, inside = grid_cell_index(mid, grid_aabb, binary.shape)
            keep = inside & binary.flatten()[idx]
        ri.append(torch.full((int(keep.sum()),), r, dtype=torch.int64)); ts.append(s[keep]); te.append(e[keep])
    cat = lambda xs, dt_: torch.cat(xs) if xs else torch.zeros(0, dtype=dt_)
    return cat(ri, torch.int64), cat(ts, torch.float32), cat(te, torch.float32)


def render_visibility(alphas: Tensor, ri: Tensor, early_stop_eps: float, alpha_thre: float) -> Tensor:
    """nerfacc's render_visibility: per ray, in order: stop once T < eps; skip (without attenuating T) alpha < thre;
    else keep and T *= 1 - alpha."""
    keep = torch.zeros(alphas.shape[0], dtype=torch.bool)
    T, cur = 1.0, -1
    for i in range(alphas.shape[0]):
        if int(ri[i]) != cur:
            cur, T = int(ri[i]), torch.tensor(1.0)
        if T < early_stop_eps:
            continue
        a = alphas[i]
        if a < alpha_thre:
            continue
        keep[i] = True
        T = T * (1.0 - a)
    return keep
