"""Evaluation sweep — the metric loop of the reference's visualization/visualization.py:188-191,277-454: render the model
from a grid of C-arm angles (th x ph, the "37 x 37" sweep at limited_size_vis = 180, angle_step_vis = 5), compare every
projection with the ground truth and tabulate per-view metrics.

What differs from upstream is where the work happens: ALL views of the sweep are ONE fused launch (rays of every pose are
generated in the kernel from the [n_views, 3, 4] pose table, `render_projection`), the ground truth of a voxel phantom is one
`afx_project_volume` launch over the same poses, and the metrics are reductions on the GPU.  Metrics kept: PSNR (:406-409,
data range 1), normalised DOT 2D (:440-450), DICE 2D on the binarised projections (:433-438; prediction binarised by zeroing
densities below `binary_thresh`, :172,349-352).  SSIM / LPIPS / DISTS upstream come from piq / torchmetrics networks (absent
here) and are out of scope."""
from __future__ import annotations

import itertools

import numpy as np
import pandas as pd
import torch

from ..phantomdata.proj_helpers import source_matrix
from ..render import render_projection


def sweep_angles(limited_size_vis: float = 180.0, angle_step_vis: float = 5.0):
    """visualization.py:188-191: th, ph in arange(-L//2, L//2 + 1, step), all pairs."""
    a = np.arange(-limited_size_vis // 2, limited_size_vis // 2 + 1, angle_step_vis).astype("float64")
    return np.array([np.array(v) for v in itertools.product(a, a)])


def _poses(angles, src_pt, translation, device):
    mats = []
    for theta, phi in angles:
        th = theta if theta >= 0 else 360 + theta            # :280-281
        ph = phi if phi >= 0 else 360 + phi
        mats.append(source_matrix(np.asarray(src_pt, dtype=np.float64), th, ph, 0.0, np.asarray(translation, dtype=np.float64)))
    return torch.from_numpy(np.stack(mats)).to(device)


@torch.no_grad()
def evaluation_sweep(model, targets, angles, img_width, img_height, focal_length, src_pt, near_thresh, far_thresh,
                     depth_samples_per_ray, translation=(0.0, 0.0, 0.0), binary_thresh=0.05, binary_targets=None,
                     views_per_launch=512):
    """Per-view metrics of `model` over `angles` [n,2] (theta, phi in degrees).

    targets: [n, H, W] ground-truth projections on the model's device (e.g. `ground_truth_sweep`), binary_targets likewise
    (optional; DICE 2D needs them).  Returns a DataFrame with the reference's columns (image_id, theta, phi, larm,
    theta_360, phi_360, cam_pose_x/y/z, PSNR, DOT 2D[, DICE 2D]) and the predicted images [n, H, W]."""
    dev = model.flat_params.device
    n = len(angles)
    poses = _poses(angles, src_pt, translation, dev)
    hw = int(img_width) * int(img_height)
    preds = torch.empty(n, hw, device=dev)
    bin_preds = torch.empty(n, hw, device=dev) if binary_targets is not None else None
    for v0 in range(0, n, views_per_launch):
        v1 = min(n, v0 + views_per_launch)
        out = render_projection(model, poses[v0:v1], img_width, img_height, focal_length, depth_samples_per_ray, near_thresh,
                                far_thresh, want_aux=bin_preds is not None)
        preds[v0:v1] = out.rgb_map.view(v1 - v0, hw)
        if bin_preds is not None:
            # densities below the threshold are zeroed before compositing (zero_idx of acc_render_volume_density)
            step = (far_thresh - near_thresh) / depth_samples_per_ray
            tau = out.sigma * (out.sigma >= binary_thresh) * step
            bin_preds[v0:v1] = torch.exp(-tau.sum(-1)).view(v1 - v0, hw)
    tgt = targets.reshape(n, hw).to(dev, torch.float32)
    mse = ((preds - tgt) ** 2).mean(-1)
    psnr = -10.0 * torch.log10(mse)
    def norm01(x):
        x = x - x.min(-1, keepdim=True).values
        return x / x.max(-1, keepdim=True).values
    dot2d = (norm01(preds) * norm01(tgt)).mean(-1)
    cols = {"image_id": [f"{t}-{p}".replace(".", ",") for t, p in angles], "theta": [float(a[0]) for a in angles],
            "phi": [float(a[1]) for a in angles], "larm": [0] * n,
            "theta_360": [float(a[0] if a[0] >= 0 else 360 + a[0]) for a in angles],
            "phi_360": [float(a[1] if a[1] >= 0 else 360 + a[1]) for a in angles],
            "cam_pose_x": poses[:, 0, 3].cpu().tolist(), "cam_pose_y": poses[:, 1, 3].cpu().tolist(),
            "cam_pose_z": poses[:, 2, 3].cpu().tolist(), "PSNR": psnr.cpu().tolist(), "DOT 2D": dot2d.cpu().tolist()}
    if bin_preds is not None:
        bp = (bin_preds >= 1).to(torch.int64)               # :434-435: everything below 1 is vessel -> 0
        bt = (binary_targets.reshape(n, hw).to(dev) >= 1).to(torch.int64)
        # Dice(average='micro') over the two classes of a binary image = pixel accuracy
        cols["DICE 2D"] = (bp == bt).float().mean(-1).cpu().tolist()
    return pd.DataFrame(cols), preds.view(n, int(img_height), int(img_width))


@torch.no_grad()
def ground_truth_sweep(volume, angles, img_width, img_height, focal_length, src_pt, depth_values, translation=(0.0, 0.0, 0.0),
                       type_ct=True):
    """Ground-truth projections of a `VoxelVolume` for every view of the sweep: one afx_project_volume launch, rays generated
    in the kernel from the poses (helpers.py:192-224 per view upstream)."""
    from ..engine import project_volume
    dev = volume.values.device
    poses = _poses(angles, src_pt, translation, dev)
    img = project_volume(volume.values, volume.origin, volume.spacing, volume.fill_value, depth_values.to(dev, torch.float32),
                         poses=poses, width=int(img_width), height=int(img_height), focal=float(focal_length), type_ct=type_ct)
    return img.view(len(angles), int(img_height), int(img_width))
