"""Ray values / depth values / ground-truth projector — the hot-path part of the reference's
phantomdata/helpers.py (:156-224), plus analytic synthetic phantoms (the reference's CT/STL data is private)."""
import numpy as np
import torch

from .proj_helpers import source_matrix


def rev_sigmoid(x, c1=1, c2=0):
    """helpers.py:17-18."""
    return 1 / (1 + np.exp(c1 * (x - c2)))


# knots of the CT transfer function (helpers.py:33-70): HU-like value -> attenuation, piecewise linear, clamped at both ends
_TF_X = (0.0, 753.0, 1585.85, 2332.9, 3306.18, 4000.0)
_TF_Y = {False: (0.0, 0.0, 0.05, 0.0, 0.2, 0.4), True: (0.0, 0.0, 0.0, 0.0, 0.2, 0.4)}


def transfer_func_ct(vals, binary=False, cathlab=False):
    """helpers.py:33-70: the manual "x-ray" transfer function (binary: vessels only)."""
    return np.interp(np.asarray(vals, dtype=np.float64), _TF_X, _TF_Y[bool(binary)])


def get_ray_values(theta, phi, larm, src_pt, img_width, img_height, focal_length, device, translation=np.array([0, 0, 0])):
    """helpers.py:156-175 -> (ray_origins[H,W,3], ray_directions[H,W,3], src_matrix, ii, jj), float64."""
    from .._geometry import camera_rays
    src_matrix = source_matrix(src_pt, theta, phi, larm, translation)
    pose = torch.from_numpy(src_matrix).to(device)
    cols = torch.arange(img_width, dtype=pose.dtype, device=pose.device)
    rows = torch.arange(img_height, dtype=pose.dtype, device=pose.device)
    ii, jj = cols[None, :].expand(img_height, img_width), rows[:, None].expand(img_height, img_width)      # 'xy' meshgrid: [H, W]
    ray_origins, ray_directions = camera_rays(pose, ii, jj, img_width, img_height, focal_length)
    return ray_origins, ray_directions, src_matrix, ii, jj


def get_depth_values(near_thresh, far_thresh, depth_samples_per_ray, device, stratified=True):
    """helpers.py:177-190."""
    from .._geometry import uniform_depths, jitter_depths
    z_vals = uniform_depths(near_thresh, far_thresh, depth_samples_per_ray)
    if stratified:
        z_vals = jitter_depths(z_vals, torch.rand(z_vals.shape))
    return z_vals.to(device)


def capsule_tree(levels=5, seed=0, extent=75.0, r0=3.0, r1=0.75):
    """Synthetic vessel tree: binary tree of 2^levels - 1 capsules inside +-extent -> [N,7] (a, b, radius)."""
    rng = np.random.RandomState(seed)
    segs, front = [], [(np.array([0.0, -0.8 * extent, 0.0]), np.array([0.0, 1.0, 0.0]), 0)]
    n_total = 2 ** levels - 1
    while front and len(segs) < n_total:
        a, dirv, lev = front.pop(0)
        b = np.clip(a + dirv * 0.55 * extent * (0.72 ** lev), -0.95 * extent, 0.95 * extent)
        segs.append(np.concatenate([a, b, [r0 + (r1 - r0) * lev / max(levels - 1, 1)]]))
        for sgn in (-1.0, 1.0):
            nd = dirv + sgn * np.cross(dirv, [0.3, 0.2, 1.0]) * 0.8 + rng.normal(size=3) * 0.35
            front.append((b, nd / np.linalg.norm(nd), lev + 1))
    return np.asarray(segs, dtype=np.float32)


def capsule_mu(points, capsules, mu=0.2):
    """Binary attenuation field: mu inside any capsule, 0 outside. points [P,3] tensor (any device)."""
    c = torch.as_tensor(capsules, dtype=points.dtype, device=points.device)
    inside = torch.zeros(points.shape[0], dtype=torch.bool, device=points.device)
    for i in range(c.shape[0]):
        a, ab, r = c[i, 0:3], c[i, 3:6] - c[i, 0:3], c[i, 6]
        t = ((points - a) @ ab / (ab @ ab)).clamp(0, 1)
        inside |= torch.norm(points - (a + t[:, None] * ab), dim=-1) <= r
    return inside.to(points.dtype) * mu


class VoxelVolume:
    """A voxel phantom on a regular grid — what get_interpolator_from_vol_* / _from_grid (helpers.py:72-154) wrap in a
    scipy RegularGridInterpolator(method='linear', bounds_error=False, fill_value=min(scalars)).  `values` [nx,ny,nz]
    lives on the GPU; projection through it runs in the HIP kernel behind afx_project_volume."""

    def __init__(self, points_x, points_y, points_z, values, fill_value=None, device="cuda:0"):
        self.axes = [np.asarray(a, dtype=np.float64) for a in (points_x, points_y, points_z)]
        for a in self.axes:
            if len(a) < 2 or not np.allclose(np.diff(a), a[1] - a[0], rtol=1e-9, atol=1e-12):
                raise ValueError("VoxelVolume: the grid must be regular (uniform spacing) along every axis")
        v = np.asarray(values, dtype=np.float32)
        if v.shape != tuple(len(a) for a in self.axes):
            raise ValueError("VoxelVolume: values shape does not match the axes")
        self.fill_value = float(np.min(v)) if fill_value is None else float(fill_value)
        self.values = torch.from_numpy(np.ascontiguousarray(v)).to(device)
        self.origin = [float(a[0]) for a in self.axes]
        self.spacing = [float(a[1] - a[0]) for a in self.axes]


def ray_tracing(interpolator, angles, ray_origins, ray_directions, depth_values, img_width, img_height, ii, jj, batch_size,
                device, proj_folder_name=None, type='ct', invert=False):
    """Upstream signature (helpers.py:192).  `interpolator`: a VoxelVolume (fused HIP projector; the tiling by
    `batch_size` and the per-row PNG dumps of the reference are unnecessary) or a callable mu(points)."""
    if isinstance(interpolator, VoxelVolume):
        from ..engine import project_volume
        dev = interpolator.values.device
        o = ray_origins.reshape(-1, 3).to(dev, torch.float32).contiguous()
        d = ray_directions.reshape(-1, 3).to(dev, torch.float32).contiguous()
        img = project_volume(interpolator.values, interpolator.origin, interpolator.spacing, interpolator.fill_value,
                             depth_values.to(dev, torch.float32), origins=o, dirs=d, type_ct=(type == 'ct'))
        return img.reshape(int(np.ceil(img_height)), int(np.ceil(img_width)))
    return ray_tracing_fn(interpolator, ray_origins, ray_directions, depth_values).reshape(int(img_height), int(img_width))


def ray_tracing_fn(mu_fn, ray_origins, ray_directions, depth_values, batch_rays=8192):
    """Ground-truth X-ray projection ('ct' branch) for a callable attenuation field mu(points):
    img = prod_s exp(-mu(o + d z_s) * dz_s * ||d||), dz_last = 1e10 (harmless: mu(far plane) = 0)."""
    shape = ray_origins.shape[:-1]
    o, d = ray_origins.reshape(-1, 3), ray_directions.reshape(-1, 3)
    big = torch.tensor([1e10], dtype=depth_values.dtype, device=depth_values.device)
    dists = torch.cat((depth_values[1:] - depth_values[:-1], big), -1)
    out = []
    for i in range(0, o.shape[0], batch_rays):
        oo, dd = o[i:i + batch_rays], d[i:i + batch_rays]
        pts = oo[:, None, :] + dd[:, None, :] * depth_values[:, None]
        mu = mu_fn(pts.reshape(-1, 3)).reshape(pts.shape[:-1]).to(depth_values.dtype)
        out.append(torch.exp(-mu * dists * torch.norm(dd[:, None, :], dim=-1)).prod(-1))
    return torch.cat(out).reshape(shape)
