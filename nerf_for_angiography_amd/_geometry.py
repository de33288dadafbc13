"""Host-side geometry shared by the mirrors of the reference's helper modules: pinhole rays of a C-arm pose and depth
schedules along a ray.  float64 for the pose arithmetic (the reference generates rays in float64 and casts when it
batches them); one implementation, used by phantomdata.helpers, phantomdata.proj_helpers and nerf.nerf_helpers."""
import torch


def camera_rays(cam2world: torch.Tensor, px: torch.Tensor, py: torch.Tensor, width: float, height: float, focal: float):
    """Rays through detector pixels (px, py): camera looks down -z, y up; direction = R . ((px - W/2)/f, -(py - H/2)/f, -1)
    (NOT normalised), origin = the pose's translation column (phantomdata/helpers.py:156-175, proj_helpers.py:9-16)."""
    rot, centre = cam2world[:3, :3], cam2world[:3, 3]
    in_camera = torch.stack(((px - width / 2) / focal, (height / 2 - py) / focal, -torch.ones_like(px)), dim=-1)
    # one multiply-and-sum per output component: the same accumulation order as the reference's broadcast product
    directions = (in_camera.unsqueeze(-2) * rot).sum(dim=-1)
    return centre.expand(directions.shape), directions


def uniform_depths(near: float, far: float, count: int) -> torch.Tensor:
    """`count` depths from near to far inclusive, written as the convex combination the reference uses."""
    s = torch.linspace(0.0, 1.0, count)
    return near * (1.0 - s) + far * s


def jitter_depths(z: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """Stratified jitter (nerf/nerf_helpers.py:13-22): sample i moves uniformly inside the interval bounded by the mid-points
    to its neighbours (the end samples keep their outer bound); u in [0,1) has z's shape."""
    half = 0.5 * (z[..., 1:] + z[..., :-1])
    left = torch.cat((z[..., :1], half), dim=-1)
    right = torch.cat((half, z[..., -1:]), dim=-1)
    return left + (right - left) * u
