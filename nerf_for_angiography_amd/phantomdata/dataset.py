"""Projection / ray data frames in the reference's wire format, a synthetic generator for them, and `load_data`.

Schema, written by the reference's generators (phantomdata/cttoray.py:255-308, sdftoray.py) and
read by its trainer (nerf/run_nerf_acc.py:82-124) as ';'-separated CSV:
  df-<file_name>-<binary_str>-cttoproj.csv : one row per projection —
      image_id, theta, phi, larm, theta_shift, phi_shift, larm_shift, translation_{x,y,z}, tform_cam2world,
      unshifted_tform_cam2world, image_data, image_distance_data, org_img_width, org_img_height, focal_length,
      near_thresh, far_thresh, depth_sample, grid_scaling_factor, depth_values, src_pt_z
  df-rays-<file_name>-<binary_str>-<img_height>.csv : one row per pixel —
      image_id, pixel_value, distance_pixel_value, x_position, y_position, ray_origins_{x,y,z}, ray_directions_{x,y,z}
The trainer calls `load_data(data_name, file_name, unseen, binary, data_size, step_size)` but the reference never
defines it; it is supplied here.  The reference's volumes are private, so `make_synthetic_dataset`
produces the same frames from an analytic capsule-tree phantom."""
from __future__ import annotations

import os
from ast import literal_eval

import numpy as np
import pandas as pd
import torch
from scipy.ndimage import distance_transform_edt

from .helpers import capsule_mu, capsule_tree, get_depth_values, get_ray_values, ray_tracing_fn as ray_tracing

PROJ_COLUMNS = ["image_id", "theta", "phi", "larm", "theta_shift", "phi_shift", "larm_shift", "translation_x",
                "translation_y", "translation_z", "tform_cam2world", "unshifted_tform_cam2world", "image_data",
                "image_distance_data", "org_img_width", "org_img_height", "focal_length", "near_thresh", "far_thresh",
                "depth_sample", "grid_scaling_factor", "depth_values", "src_pt_z"]
RAY_COLUMNS = ["image_id", "pixel_value", "distance_pixel_value", "x_position", "y_position", "ray_origins_x",
               "ray_origins_y", "ray_origins_z", "ray_directions_x", "ray_directions_y", "ray_directions_z"]
_LIST_COLUMNS = ["tform_cam2world", "unshifted_tform_cam2world", "image_data", "image_distance_data", "depth_values"]


def sampling_weights(img: np.ndarray, sampling_strategy: str = "segmentation") -> np.ndarray:
    """Per-pixel ray-sampling weights (get_weighted_img, phantomdata/helpers.py:226-247): normalised Euclidean
    distance transform of the vessel mask, +1e-10.  'random' gives uniform weights (cttoray.py:221-222)."""
    if sampling_strategy == "random":
        return np.ones(img.shape)
    if sampling_strategy == "frangi":
        raise NotImplementedError("the Frangi vesselness filter (scikit-image) is not part of this build; "
                                  "use sampling_strategy='segmentation' or 'random'")
    mask = np.zeros(img.shape)
    mask[img < 1] = 1
    mask -= mask.min()
    if mask.max() > 0:
        mask /= mask.max()
    edt = distance_transform_edt(mask)
    edt -= edt.min()
    if edt.max() > 0:
        edt /= edt.max()
    return edt + 1e-10


def angle_grid(limited_size: float, number_angles: int, center_point=(90, 0)):
    """(theta, phi) grid of cttoray.py:88-105: number_angles+1 samples per axis over `limited_size` degrees around
    `center_point`, plus the centre itself as the last (held-out) projection."""
    if number_angles > 0:
        step = limited_size / number_angles
        th = center_point[0] - limited_size / 2 + step * np.arange(number_angles + 1)
        ph = center_point[1] - limited_size / 2 + step * np.arange(number_angles + 1)
    else:
        th, ph = np.array([center_point[0]]), np.array([center_point[1]])
    grid = [(float(t), float(p)) for t in th for p in ph]
    grid.append((float(center_point[0]) + 7.0, float(center_point[1]) + 5.0))     # held-out test view
    return grid


def make_synthetic_dataset(angles, img_size: int = 64, depth_samples_per_ray: int = 160, outside: float = 100.0,
                           src_z: float = 1500.0, sampling_strategy: str = "segmentation", device="cpu", seed: int = 0,
                           binary: bool = True):
    """(proj_df, ray_df) for a capsule-tree phantom seen from `angles` = [(theta, phi), ...]; larm = 0, no shifts."""
    w = h = int(img_size)
    focal = 13.0 * w                       # same field of view as the reference's f=1300 @ 100 px
    src_pt = np.array([0.0, 0.0, src_z])
    near, far = src_z - outside, src_z + outside
    caps = capsule_tree(levels=5, seed=seed)
    proj_rows, ray_frames = [], []
    for image_id, (theta, phi) in enumerate(angles):
        o, d, mat, ii, jj = get_ray_values(theta, phi, 0.0, src_pt, w, h, focal, device)
        z = get_depth_values(near, far, depth_samples_per_ray, device, stratified=False)
        with torch.no_grad():
            img = ray_tracing(lambda p: capsule_mu(p, caps), o.reshape(-1, 3).float(), d.reshape(-1, 3).float(),
                              z.float(), batch_rays=8192).reshape(h, w).cpu()
        img_np = img.numpy().astype(np.float64)
        wts = sampling_weights(img_np, sampling_strategy)
        proj_rows.append(dict(image_id=image_id, theta=theta, phi=phi, larm=0.0, theta_shift=0.0, phi_shift=0.0,
                              larm_shift=0.0, translation_x=0.0, translation_y=0.0, translation_z=0.0,
                              tform_cam2world=mat.tolist(), unshifted_tform_cam2world=mat.tolist(),
                              image_data=img_np.tolist(), image_distance_data=wts.tolist(), org_img_width=w,
                              org_img_height=h, focal_length=focal, near_thresh=near, far_thresh=far,
                              depth_sample=depth_samples_per_ray, grid_scaling_factor=1,
                              depth_values=z.cpu().numpy().tolist(), src_pt_z=src_z))
        oo, dd = o.reshape(-1, 3).cpu().numpy(), d.reshape(-1, 3).cpu().numpy()
        ray_frames.append(pd.DataFrame({
            "image_id": np.repeat(image_id, w * h), "pixel_value": img_np.flatten(), "distance_pixel_value": wts.flatten(),
            "x_position": ii.flatten().cpu().numpy(), "y_position": jj.flatten().cpu().numpy(),
            "ray_origins_x": oo[:, 0], "ray_origins_y": oo[:, 1], "ray_origins_z": oo[:, 2],
            "ray_directions_x": dd[:, 0], "ray_directions_y": dd[:, 1], "ray_directions_z": dd[:, 2]}))
    proj_df = pd.DataFrame(proj_rows, columns=PROJ_COLUMNS)
    # the reference normalises the stacked images of the projection frame (cttoray.py:266-268)
    imgs = np.array(proj_df["image_data"].tolist())
    imgs = imgs - imgs.min()
    if imgs.max() > 0:
        imgs = imgs / imgs.max()
    proj_df["image_data"] = imgs.tolist()
    return proj_df, pd.concat(ray_frames, ignore_index=True)[RAY_COLUMNS]


def dataset_paths(data_folder: str, file_name: str, binary: bool, img_height: int):
    b = "binary" if binary else "non-binary"
    return (os.path.join(data_folder, f"df-{file_name}-{b}-cttoproj.csv"),
            os.path.join(data_folder, f"df-rays-{file_name}-{b}-{img_height}.csv"))


def save_dataset(proj_df, ray_df, data_folder: str, file_name: str, binary: bool = True):
    """Write both frames exactly as cttoray.py:286,308 does (';' separator, index column included)."""
    os.makedirs(data_folder, exist_ok=True)
    p, r = dataset_paths(data_folder, file_name, binary, int(proj_df["org_img_height"].iloc[0]))
    proj_df.to_csv(p, sep=";")
    ray_df.to_csv(r, sep=";")
    return p, r


def load_data(data_name, file_name, unseen=False, binary=False, data_size=100, step_size=None, data_root="data"):
    """The loader nerf/run_nerf_acc.py:82 calls -> (proj_df, ray_df, store_folder_name, unseen_ray_df).

    Reads <data_root>/<data_name>/df-<file_name>-<binary|non-binary>-cttoproj.csv and the matching df-rays-*-<size>.csv;
    list-valued cells come back as Python lists (the reference parses them with literal_eval, nerf_helpers.py:8-11)."""
    folder = os.path.join(data_root, str(data_name))
    p, r = dataset_paths(folder, file_name, binary, int(data_size))
    if not (os.path.exists(p) and os.path.exists(r)):
        raise FileNotFoundError(f"dataset not found: {p} / {r}")
    proj_df = pd.read_csv(p, sep=";", index_col=0)
    for col in _LIST_COLUMNS:
        proj_df[col] = proj_df[col].apply(literal_eval)
    ray_df = pd.read_csv(r, sep=";", index_col=0)
    unseen_ray_df = ray_df.iloc[0:0].copy() if unseen else None
    return proj_df, ray_df, folder, unseen_ray_df
