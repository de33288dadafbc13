"""C-arm pose and ray geometry — mirror of the reference's phantomdata/proj_helpers.py (host side, float64)."""
import numpy as np
import torch


def _rotation(axis, angle):
    c, s = np.cos(angle), np.sin(angle)
    m = np.identity(4)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[axis]
    m[i, i] = m[j, j] = c
    m[i, j], m[j, i] = -s, s
    return m


def x_rotation_matrix(angle):
    return _rotation("x", angle)


def y_rotation_matrix(angle):
    return _rotation("y", angle)


def z_rotation_matrix(angle):
    return _rotation("z", angle)


def translation_matrix(vec):
    m = np.identity(4)
    m[:3, 3] = np.asarray(vec, dtype=np.float64)[:3]
    return m


def get_rotation(theta, phi, larm, type='rotation'):
    """inv(Rz(larm) Rx(theta) Ry(phi)), degrees (proj_helpers.py:63-66)."""
    return np.linalg.inv(z_rotation_matrix(np.deg2rad(larm)).dot(
        x_rotation_matrix(np.deg2rad(theta)).dot(y_rotation_matrix(np.deg2rad(phi)))))


def source_matrix(source_pt, theta, phi, larm=0, translation=[0, 0, 0], type='rotation'):
    """T(translation) . get_rotation . T(source_pt): used as camera->world (proj_helpers.py:68-77)."""
    return translation_matrix(translation).dot(get_rotation(theta, phi, larm).dot(translation_matrix(source_pt)))


def get_query_points(x, y, img_width, img_height, focal_length, tform_cam2world, depth_samples_per_ray,
                     near_thresh, far_thresh, device, randomize=False):
    """proj_helpers.py:9-32 -> (query_points, ray_origins, ray_directions, depth_values); d is not normalised."""
    from .._geometry import camera_rays, uniform_depths, jitter_depths
    ray_origins, ray_directions = camera_rays(tform_cam2world, x.to(tform_cam2world), y.to(tform_cam2world), img_width,
                                              img_height, focal_length)
    ray_origins, ray_directions = ray_origins.to(device), ray_directions.to(device)
    depth_values = uniform_depths(near_thresh, far_thresh, depth_samples_per_ray)
    if randomize:
        depth_values = jitter_depths(depth_values, torch.rand(depth_values.shape))
    depth_values = depth_values.to(ray_origins.device)
    return ray_origins.unsqueeze(-2) + ray_directions.unsqueeze(-2) * depth_values.unsqueeze(-1), ray_origins, ray_directions, depth_values
