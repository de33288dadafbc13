"""Host-side driver of libafx.so: owns the torch buffers (prepared weights, workspace) the C-ABI
works on and exposes the fused kernels to the Python mirror of the reference interface.
PyTorch is used for device memory and streams only."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import AfxError, ModelDesc, RenderArgs


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a tensor")
    if t.device != device:
        raise ValueError(f"{name}: on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        raise ValueError(f"{name}: dtype {t.dtype}, expected float32")
    return t if t.is_contiguous() else t.contiguous()


@dataclass
class RenderSpec:
    """One ray batch for the fused renderer.

    Rays: either origins/dirs [R,3] fp32 (what sample_pixel_rays returns), or poses [n_proj,3,4] float64 +
    ray_ids (int32 index into [n_proj,H,W]; None => rays ray_id0 .. ray_id0+R-1) + width/height/focal
    (in-kernel get_ray_values).  Depths: mode 'acc' (uniform mid-point march t_near..t_far, the convention of
    nerf_helpers_acc.py), 'dense' with z [S] or [R,S] (render_volume_density convention), 'stratified' (dense convention
    with randomize_depth(linspace(t_near, t_far, S)) drawn in the kernel from Philox (jitter_seed, jitter_stream))."""
    n_rays: int
    n_samples: int
    origins: Optional[torch.Tensor] = None
    dirs: Optional[torch.Tensor] = None
    poses: Optional[torch.Tensor] = None
    ray_ids: Optional[torch.Tensor] = None
    ray_id0: int = 0
    width: int = 0
    height: int = 0
    focal: float = 0.0
    mode: str = "acc"
    t_near: float = 0.0
    t_far: float = 0.0
    z: Optional[torch.Tensor] = None
    jitter_seed: int = 0          # mode 'stratified': Philox stream of the in-kernel randomize_depth draw
    jitter_stream: int = 0


class Engine:
    """One CPPN geometry on one GPU."""

    def __init__(self, width: int, n_hidden: int, enc: str = "none", n_freq: int = 0,
                 max_workspace_bytes: int = 24 << 30, variant: str = "", act: str = "relu", act_w0: float = 1.0):
        self.lib = _lib.load(variant)
        self.act = act
        self.desc = ModelDesc(3, _lib.ENC[enc], int(n_freq), int(width), int(n_hidden), _lib.ACT[act], float(act_w0))
        h = C.c_void_p()
        self._check(self.lib.afx_create(C.byref(self.desc), C.byref(h)), "afx_create")
        self.h = h
        self.width, self.n_hidden, self.enc = width, n_hidden, enc
        self.param_count = int(self.lib.afx_query(h, _lib.Q_PARAM_COUNT, 0, 0, 0))
        self.k0 = int(self.lib.afx_query(h, _lib.Q_K0, 0, 0, 0))
        self.max_workspace_bytes = int(max_workspace_bytes)
        self._prepared = {}      # prec -> (buffer, version key)
        self._ws = None
        self._ws_captured = False
        self._retired = []       # workspaces a captured graph still points at

    def _check(self, rc, what):
        _lib.check(rc, what, self.lib)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.afx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- measurement ---------------------------------------------------------------------------
    KERNELS = {"chain_fwd": 0, "chain_bwd": 1, "wgrad": 2}

    def profile(self, on: bool):
        self._check(self.lib.afx_profile_enable(self.h, int(on)), "afx_profile_enable")

    def profile_read(self, kernel: str):
        """(total device ms, launches) of one kernel kind since the last read (HIP events on the launch stream)."""
        ms, n = C.c_double(), C.c_int64()
        self._check(self.lib.afx_profile_read(self.h, self.KERNELS[kernel], C.byref(ms), C.byref(n)), "afx_profile_read")
        return ms.value, n.value

    # ---- parameter layout ------------------------------------------------------------------
    def layout(self, layer: int):
        wo, bo, r, c = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        self._check(self.lib.afx_param_layout(self.h, layer, C.byref(wo), C.byref(bo), C.byref(r), C.byref(c)),
                   "afx_param_layout")
        return wo.value, bo.value, r.value, c.value

    @staticmethod
    def _stream(device):
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        """The (one) workspace buffer, grown on demand.  A HIP graph captured over a call keeps the buffer's address:
        a buffer that a capture has seen is never freed when a later, larger request replaces it, and growing it
        DURING a capture is refused (run the step once eagerly first - that sizes it)."""
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            if capturing:
                raise AfxError("the workspace must be sized before graph capture: run the same call once eagerly first")
            if self._ws is not None and self._ws_captured:
                self._retired.append(self._ws)
            # a request that outgrows the buffer gets 25 % head-room (within the cap): the packed / points paths ask for a slightly different
            # size every iteration, and a hipMalloc per new maximum (several ms at GB sizes) would dominate them
            grow = self._ws is not None and self._ws.device == device
            self._ws = None
            size = int(nbytes)
            if grow:
                size = max(size, min(int(size * 1.25), max(self.max_workspace_bytes, size)))
            self._ws = torch.empty(size, dtype=torch.uint8, device=device)
            self._ws_captured = False
        if capturing:
            self._ws_captured = True
        return self._ws

    # ---- weights -----------------------------------------------------------------------------
    def prepare(self, flat: torch.Tensor, enc_aux: Optional[torch.Tensor], prec: str, key=None):
        """Re-tile the flat parameters for `prec` unless `key` says the cached copy is current."""
        if not flat.is_cuda:
            raise AfxError("the fused MI355X path needs parameters on a GPU; there is no CPU fallback")
        flat = _f32(flat, "flat parameters", flat.device)
        if flat.numel() != self.param_count:
            raise ValueError(f"flat parameters: {flat.numel()} floats, expected {self.param_count}")
        p = _lib.PREC[prec]
        cached = self._prepared.get(prec)
        if cached is not None and key is not None and cached[1] == key and cached[0].device == flat.device:
            return cached[0]
        nbytes = int(self.lib.afx_query(self.h, _lib.Q_PREPARED_BYTES, p, 0, 0))
        buf = cached[0] if cached is not None and cached[0].device == flat.device else torch.empty(
            nbytes, dtype=torch.uint8, device=flat.device)
        if enc_aux is not None:
            enc_aux = _f32(enc_aux, "enc_aux", flat.device)
        self._check(self.lib.afx_prepare_weights(self.h, p, _ptr(flat), _ptr(enc_aux), _ptr(buf), nbytes,
                                                self._stream(flat.device)), "afx_prepare_weights")
        self._prepared[prec] = (buf, key)
        return buf

    # ---- MLP on explicit points --------------------------------------------------------------
    def infer(self, prepared: torch.Tensor, pts: torch.Tensor, prec: str, apply_sigmoid: bool = False):
        dev = prepared.device
        pts = _f32(pts, "points", dev)
        if pts.dim() != 2 or pts.shape[1] != 3:
            raise ValueError(f"points: shape {tuple(pts.shape)}, expected [P,3]")
        out = torch.empty(pts.shape[0], dtype=torch.float32, device=dev)
        self._check(self.lib.afx_mlp_infer(self.h, _lib.PREC[prec], _ptr(prepared), _ptr(pts), pts.shape[0], _ptr(out),
                                          int(apply_sigmoid), self._stream(dev)), "afx_mlp_infer")
        return out

    @contextlib.contextmanager
    def encoding_grad(self, flat: torch.Tensor, d_aux: Optional[torch.Tensor]):
        """Backward calls inside this scope also accumulate d loss / d fourier coefficients into `d_aux`
        (afx_set_encoding_grad); None: plain scope."""
        if d_aux is None:
            yield
            return
        self._check(self.lib.afx_set_encoding_grad(self.h, _ptr(flat), _ptr(d_aux)), "afx_set_encoding_grad")
        try:
            yield
        finally:
            self.lib.afx_set_encoding_grad(self.h, None, None)

    def mlp_backward(self, prepared, pts, d_out, grad_flat, prec: str):
        dev = prepared.device
        pts = _f32(pts, "points", dev)
        d_out = _f32(d_out, "d_out", dev)
        n = pts.shape[0]
        if d_out.numel() != n:
            raise ValueError("d_out: one value per point expected")
        full = int(self.lib.afx_query(self.h, _lib.Q_BWD_WORKSPACE_FULL, 0, n, _lib.PREC[prec]))
        ws = self._workspace(min(full, self.max_workspace_bytes), dev)
        self._check(self.lib.afx_mlp_backward(self.h, _lib.PREC[prec], _ptr(prepared), _ptr(pts), n, _ptr(d_out),
                                             _ptr(grad_flat), _ptr(ws), ws.numel(), self._stream(dev)),
                   "afx_mlp_backward")

    # ---- fused renderer ----------------------------------------------------------------------
    def _render_args(self, spec: RenderSpec, dev, pixel, sigma=None, tau=None):
        a = RenderArgs()
        keep = []
        a.n_rays, a.n_samples = int(spec.n_rays), int(spec.n_samples)
        if spec.poses is not None:
            poses = spec.poses
            if poses.device != dev or poses.dtype != torch.float64:
                raise ValueError("poses: expected a float64 tensor on the model's device")
            poses = poses.contiguous().reshape(-1, 12)
            keep.append(poses)
            a.ray_mode, a.poses = _lib.RAYS_POSE, poses.data_ptr()
            if spec.ray_ids is not None:
                ids = spec.ray_ids
                if ids.device != dev or ids.dtype != torch.int32 or ids.numel() != spec.n_rays:
                    raise ValueError("ray_ids: expected int32 [n_rays] on the model's device")
                ids = ids.contiguous()
                keep.append(ids)
                a.ray_ids = ids.data_ptr()
            a.ray_id0, a.width, a.height, a.focal = int(spec.ray_id0), int(spec.width), int(spec.height), float(spec.focal)
        else:
            o, d = _f32(spec.origins, "origins", dev), _f32(spec.dirs, "dirs", dev)
            if tuple(o.shape) != (spec.n_rays, 3) or tuple(d.shape) != (spec.n_rays, 3):
                raise ValueError("origins/dirs: expected [n_rays,3]")
            keep += [o, d]
            a.ray_mode, a.origins, a.dirs = _lib.RAYS_ARRAYS, o.data_ptr(), d.data_ptr()
        if spec.mode == "acc":
            a.depth_mode, a.t_near, a.t_far = _lib.DEPTH_UNIFORM_MID, float(spec.t_near), float(spec.t_far)
        elif spec.mode == "dense":
            z = _f32(spec.z, "z", dev)
            if z.dim() == 1 and z.shape[0] == spec.n_samples:
                a.depth_mode = _lib.DEPTH_SHARED_Z
            elif tuple(z.shape) == (spec.n_rays, spec.n_samples):
                a.depth_mode = _lib.DEPTH_PER_RAY_Z
            else:
                raise ValueError(f"z: shape {tuple(z.shape)}, expected [S] or [R,S]")
            keep.append(z)
            a.z = z.data_ptr()
        elif spec.mode == "stratified":
            a.depth_mode, a.t_near, a.t_far = _lib.DEPTH_STRATIFIED, float(spec.t_near), float(spec.t_far)
            a.jitter_seed, a.jitter_stream = int(spec.jitter_seed), int(spec.jitter_stream)
        else:
            raise ValueError(f"mode {spec.mode!r}: expected 'acc', 'dense' or 'stratified'")
        a.pixel = pixel.data_ptr()
        if sigma is not None:
            a.sigma = sigma.data_ptr()
        if tau is not None:
            a.tau = tau.data_ptr()
        return a, keep

    def render_forward(self, prepared, spec: RenderSpec, prec: str, want_sigma=False, want_tau=False):
        dev = prepared.device
        pixel = torch.empty(spec.n_rays, dtype=torch.float32, device=dev)
        sigma = torch.empty(spec.n_rays, spec.n_samples, dtype=torch.float32, device=dev) if want_sigma else None
        tau = torch.empty(spec.n_rays, spec.n_samples, dtype=torch.float32, device=dev) if want_tau else None
        a, keep = self._render_args(spec, dev, pixel, sigma, tau)
        ws = self._workspace(int(self.lib.afx_query(self.h, _lib.Q_FWD_WORKSPACE, spec.n_rays, spec.n_samples, 0)), dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        self._check(self.lib.afx_render_forward(self.h, _lib.PREC[prec], _ptr(prepared), C.byref(a), self._stream(dev)),
                   "afx_render_forward")
        del keep
        return pixel, sigma, tau

    def render_backward(self, prepared, spec: RenderSpec, pixel, d_pixel, grad_flat, prec: str):
        dev = prepared.device
        pixel = _f32(pixel, "pixel", dev)
        d_pixel = _f32(d_pixel, "d_pixel", dev)
        a, keep = self._render_args(spec, dev, pixel)
        full = int(self.lib.afx_query(self.h, _lib.Q_BWD_WORKSPACE_FULL, spec.n_rays, spec.n_samples, _lib.PREC[prec]))
        ws = self._workspace(min(full, self.max_workspace_bytes), dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        self._check(self.lib.afx_render_backward(self.h, _lib.PREC[prec], _ptr(prepared), C.byref(a), _ptr(d_pixel),
                                                _ptr(grad_flat), self._stream(dev)), "afx_render_backward")
        del keep


    def train_step_packed_mse(self, prepared, origins, dirs, packed: "PackedGroups", target, inv_n: float, grad_flat, prec: str):
        """afx_train_step_packed_mse: the reference's iteration body on the march's packed samples; returns the pixels [n_rays]."""
        dev = prepared.device
        o, d, target = _f32(origins, "origins", dev), _f32(dirs, "dirs", dev), _f32(target, "target", dev)
        n_rays = packed.n_rays
        if tuple(o.shape) != (n_rays, 3) or tuple(d.shape) != (n_rays, 3) or target.numel() != n_rays:
            raise ValueError("train_step_packed_mse: origins/dirs [n_rays,3] and target [n_rays] expected")
        pixel = torch.empty(n_rays, dtype=torch.float32, device=dev)
        full = int(self.lib.afx_query(self.h, _lib.Q_BWD_WORKSPACE_FULL, 0, max(packed.n_groups, 1) * 32, _lib.PREC[prec]))
        full += 4 * (n_rays + packed.n_groups) + 1024
        if full > self.max_workspace_bytes:
            raise AfxError(f"train_step_packed_mse: {packed.n_groups * 32} packed samples need a {full >> 20} MiB workspace in one piece "
                           f"(max_workspace_bytes = {self.max_workspace_bytes >> 20} MiB)")
        ws = self._workspace(full, dev)
        self._check(self.lib.afx_train_step_packed_mse(self.h, _lib.PREC[prec], _ptr(prepared), _ptr(o), _ptr(d), n_rays,
                                                      _ptr(packed.group_offsets), _ptr(packed.group_ray), packed.n_groups,
                                                      _ptr(packed.ts_pad), _ptr(packed.te_pad), _ptr(target), float(inv_n), _ptr(pixel),
                                                      _ptr(grad_flat), _ptr(ws), ws.numel(), self._stream(dev)), "afx_train_step_packed_mse")
        return pixel

    def march_train_step_mse(self, prepared, origins, dirs, target, inv_n: float, grad_flat, prec: str, scene_aabb, near_plane, far_plane,
                             step: float, early_stop_eps: float, alpha_thre: float, grid_bits=None, grid_aabb=None, grid_res=None):
        """afx_march_train_step_mse: the reference's grid iteration (march, alpha pass, visibility, packed training step) in one call.
        Returns (pixels [n_rays], n_candidates, n_kept); n_kept == 0: nothing survived, pixels / grad_flat untouched."""
        dev = prepared.device
        o, d, target = _f32(origins, "origins", dev), _f32(dirs, "dirs", dev), _f32(target, "target", dev)
        n_rays = o.shape[0]
        if tuple(o.shape) != (n_rays, 3) or tuple(d.shape) != (n_rays, 3) or target.numel() != n_rays:
            raise ValueError("march_train_step_mse: origins/dirs [n_rays,3] and target [n_rays] expected")
        pixel = torch.empty(n_rays, dtype=torch.float32, device=dev)
        a = _lib.MarchTrainArgs()
        _fill_march_args(a.march, o, d, scene_aabb, near_plane, far_plane, step, grid_bits, grid_aabb, grid_res)
        a.early_stop_eps, a.alpha_thre, a.inv_n = float(early_stop_eps), float(alpha_thre), float(inv_n)
        a.target, a.pixel, a.grad_flat = target.data_ptr(), pixel.data_ptr(), grad_flat.data_ptr()
        need = max(64 << 20, self._ws.numel() if self._ws is not None and self._ws.device == dev else 0)
        for _ in range(4):
            if need > self.max_workspace_bytes:
                raise AfxError(f"march_train_step_mse: the iteration needs a {need >> 20} MiB workspace (max_workspace_bytes = "
                               f"{self.max_workspace_bytes >> 20} MiB)")
            ws = self._workspace(need, dev)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            rc = self.lib.afx_march_train_step_mse(self.h, _lib.PREC[prec], _ptr(prepared), C.byref(a), self._stream(dev))
            if rc == -2 and a.workspace_needed > ws.numel():      # AFX_E_WORKSPACE: sizes are data - grow (25 % head-room) and run the iteration again
                need = int(a.workspace_needed)
                continue
            self._check(rc, "afx_march_train_step_mse")
            return pixel, int(a.n_candidates), int(a.n_kept)
        raise AfxError("march_train_step_mse: the workspace did not converge")

    def hier_train_step_mse(self, prepared, spec: RenderSpec, n_fine: int, u, target, inv_n: float, grad_flat, prec: str, want_z_all=True):
        """afx_hier_train_step_mse: hierarchical step with coarse re-use; `spec` carries the rays and the COARSE depths (mode 'dense').
        Returns (pixels [R], merged depths [R, S + n_fine] | None)."""
        dev = prepared.device
        target, u = _f32(target, "target", dev), _f32(u, "u", dev)
        if target.numel() != spec.n_rays or tuple(u.shape) != (spec.n_rays, int(n_fine)):
            raise ValueError("hier_train_step_mse: target [n_rays] and u [n_rays, n_fine] expected")
        pixel = torch.empty(spec.n_rays, dtype=torch.float32, device=dev)
        z_all = torch.empty(spec.n_rays, spec.n_samples + int(n_fine), dtype=torch.float32, device=dev) if want_z_all else None
        a, keep = self._render_args(spec, dev, pixel)
        full = int(self.lib.afx_hier_workspace_bytes(self.h, spec.n_rays, spec.n_samples, int(n_fine)))
        ws = self._workspace(min(full, self.max_workspace_bytes), dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        self._check(self.lib.afx_hier_train_step_mse(self.h, _lib.PREC[prec], _ptr(prepared), C.byref(a), int(n_fine), _ptr(u), _ptr(target),
                                                    float(inv_n), _ptr(z_all), _ptr(grad_flat), self._stream(dev)), "afx_hier_train_step_mse")
        del keep
        return pixel, z_all

    def fused_step_available(self, n_samples: int, prec: str) -> bool:
        """afx_train_step_mse takes this ray length at this precision: rays that fit a 256-sample workgroup tile always (one
        kernel per chunk); any other length (300, 128 + 64, ...) as two half-kernels per chunk with the 8-bit-stash kernel
        (include/afx.h)."""
        s_pad = (int(n_samples) + 31) // 32 * 32
        return 256 % s_pad == 0 or (prec == "f16s8" and os.environ.get("AFX_SMALL_IN_KERNEL", "1") != "0"
                                    and os.environ.get("AFX_NO_SPLIT", "0") == "0")      # (AFX_NO_SPLIT=1: A/B against the two-launch path)

    def train_step_mse(self, prepared, spec: RenderSpec, target, inv_n: float, grad_flat, prec: str):
        """Fused forward + MSE + backward (afx_train_step_mse); returns the rendered pixels."""
        dev = prepared.device
        target = _f32(target, "target", dev)
        if target.numel() != spec.n_rays:
            raise ValueError("target: one value per ray expected")
        pixel = torch.empty(spec.n_rays, dtype=torch.float32, device=dev)
        a, keep = self._render_args(spec, dev, pixel)
        full = int(self.lib.afx_query(self.h, _lib.Q_BWD_WORKSPACE_FULL, spec.n_rays, spec.n_samples, _lib.PREC[prec]))
        ws = self._workspace(min(full, self.max_workspace_bytes), dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        self._check(self.lib.afx_train_step_mse(self.h, _lib.PREC[prec], _ptr(prepared), C.byref(a), _ptr(target),
                                               float(inv_n), _ptr(grad_flat), self._stream(dev)), "afx_train_step_mse")
        del keep
        return pixel


# ---- packed samples of the occupancy-grid march, group-aligned for the fused packed training step
class PackedGroups:
    """A packed, ray-sorted sample list re-laid out for the fused packed training step (afx_pack_groups): every ray starts on a
    32-sample group (`group_offsets` int64 [R+1] in groups), dead padding behind a ray's last sample, `group_ray` int32 [n_groups]."""

    def __init__(self, n_rays, n_groups, group_offsets, group_ray, ts_pad, te_pad):
        self.n_rays, self.n_groups = int(n_rays), int(n_groups)
        self.group_offsets, self.group_ray, self.ts_pad, self.te_pad = group_offsets, group_ray, ts_pad, te_pad


def pack_groups(offsets, t_starts, t_ends, n_groups=None, group_offsets=None) -> PackedGroups:
    """offsets int64 [R+1] (exclusive scan of the per-ray sample counts), t_starts / t_ends [n] -> PackedGroups."""
    lib = _lib.load()
    dev = offsets.device
    if dev.type != "cuda":
        raise AfxError("pack_groups: the sample list must live on a GPU; there is no CPU fallback")
    n_rays = offsets.numel() - 1
    if group_offsets is None:
        cnt = offsets[1:] - offsets[:-1]
        group_offsets = torch.zeros(n_rays + 1, dtype=torch.int64, device=dev)
        torch.cumsum((cnt + 31) // 32, 0, out=group_offsets[1:])
    if n_groups is None:
        n_groups = int(group_offsets[-1])
    ts_pad = torch.empty(n_groups * 32, device=dev)
    te_pad = torch.empty(n_groups * 32, device=dev)
    group_ray = torch.empty(n_groups, dtype=torch.int32, device=dev)
    if n_groups:
        ts, te = _f32(t_starts.reshape(-1), "t_starts", dev), _f32(t_ends.reshape(-1), "t_ends", dev)
        _lib.check(lib.afx_pack_groups(_ptr(offsets), _ptr(group_offsets), n_rays, _ptr(ts), _ptr(te), _ptr(ts_pad), _ptr(te_pad),
                                       _ptr(group_ray), Engine._stream(dev)), "afx_pack_groups")
    return PackedGroups(n_rays, n_groups, group_offsets, group_ray, ts_pad, te_pad)


# ---- stand-alone compositing / sampling kernels (no model) -----------------------------------
def composite_dense(raw, dirs, z, want_aux=True):
    lib = _lib.load()
    dev = raw.device
    raw, dirs, z = _f32(raw, "raw", dev), _f32(dirs, "dirs", dev), _f32(z, "z", dev)
    r, s = raw.shape
    rgb = torch.empty(r, device=dev)
    depth = torch.empty(r, device=dev) if want_aux else None
    w = torch.empty(r, s, device=dev) if want_aux else None
    ent = torch.empty(r, device=dev) if want_aux else None
    sig = torch.empty(r, s, device=dev) if want_aux else None
    _lib.check(lib.afx_composite_dense(_ptr(raw), _ptr(dirs), _ptr(z), int(z.dim() == 2), r, s, _ptr(rgb), _ptr(depth),
                                       _ptr(w), _ptr(ent), _ptr(sig), Engine._stream(dev)), "afx_composite_dense")
    return rgb, depth, w, ent, sig


def composite_dense_backward(raw, dirs, z, rgb, d_rgb):
    lib = _lib.load()
    dev = raw.device
    d_raw = torch.empty_like(raw)
    r, s = raw.shape
    _lib.check(lib.afx_composite_dense_backward(_ptr(raw), _ptr(dirs), _ptr(z), int(z.dim() == 2), r, s, _ptr(rgb),
                                                _ptr(_f32(d_rgb, "d_rgb", dev)), _ptr(d_raw), Engine._stream(dev)),
               "afx_composite_dense_backward")
    return d_raw


def composite_packed(pred, ray_indices, t_starts, t_ends, n_rays):
    lib = _lib.load()
    dev = pred.device
    rgb = torch.empty(n_rays, device=dev)
    _lib.check(lib.afx_composite_packed(_ptr(pred), _ptr(ray_indices), _ptr(t_starts), _ptr(t_ends), pred.numel(), n_rays,
                                        _ptr(rgb), Engine._stream(dev)), "afx_composite_packed")
    return rgb


def composite_packed_backward(pred, ray_indices, t_starts, t_ends, n_rays, rgb, d_rgb):
    lib = _lib.load()
    dev = pred.device
    d_pred = torch.empty_like(pred)
    _lib.check(lib.afx_composite_packed_backward(_ptr(pred), _ptr(ray_indices), _ptr(t_starts), _ptr(t_ends), pred.numel(),
                                                 n_rays, _ptr(rgb), _ptr(_f32(d_rgb, "d_rgb", dev)), _ptr(d_pred),
                                                 Engine._stream(dev)), "afx_composite_packed_backward")
    return d_pred


def project_volume(vol, origin, spacing, fill_value, depth_values, origins=None, dirs=None, poses=None, width=0, height=0,
                   focal=0.0, ray_ids=None, ray_id0=0, n_rays=None, type_ct=True):
    """afx_project_volume: X-ray projection of a voxel volume (trilinear lookup) along rays -> pixel[R]."""
    lib = _lib.load()
    dev = vol.device
    if not vol.is_cuda:
        raise AfxError("project_volume: the volume must live on a GPU; there is no CPU fallback")
    vol = _f32(vol, "volume", dev)
    if vol.dim() != 3:
        raise ValueError("volume: expected [nx,ny,nz]")
    z = _f32(depth_values, "depth_values", dev)
    a = RenderArgs()
    keep = [vol, z]
    if poses is not None:
        poses = poses[:, :3, :].to(dev, torch.float64).contiguous().reshape(-1, 12)
        keep.append(poses)
        n = n_rays if n_rays is not None else (ray_ids.numel() if ray_ids is not None else poses.shape[0] * width * height - ray_id0)
        a.ray_mode, a.poses = _lib.RAYS_POSE, poses.data_ptr()
        if ray_ids is not None:
            ray_ids = ray_ids.to(dev, torch.int32).contiguous()
            keep.append(ray_ids)
            a.ray_ids = ray_ids.data_ptr()
        a.ray_id0, a.width, a.height, a.focal = int(ray_id0), int(width), int(height), float(focal)
    else:
        o, d = _f32(origins, "origins", dev), _f32(dirs, "dirs", dev)
        keep += [o, d]
        n = o.shape[0]
        a.ray_mode, a.origins, a.dirs = _lib.RAYS_ARRAYS, o.data_ptr(), d.data_ptr()
    pixel = torch.empty(int(n), dtype=torch.float32, device=dev)
    a.n_rays, a.n_samples, a.depth_mode, a.z, a.pixel = int(n), int(z.numel()), _lib.DEPTH_SHARED_Z, z.data_ptr(), pixel.data_ptr()
    org = (C.c_double * 3)(*[float(x) for x in origin])
    spc = (C.c_double * 3)(*[float(x) for x in spacing])
    _lib.check(lib.afx_project_volume(_ptr(vol), vol.shape[0], vol.shape[1], vol.shape[2], org, spc, float(fill_value),
                                      C.byref(a), int(bool(type_ct)), Engine._stream(dev)), "afx_project_volume")
    del keep
    return pixel


def fine_depths(z_coarse, w_coarse, u):
    lib = _lib.load()
    dev = w_coarse.device
    z_coarse, w_coarse, u = _f32(z_coarse, "z_coarse", dev), _f32(w_coarse, "w_coarse", dev), _f32(u, "u", dev)
    r, s = w_coarse.shape
    nf = u.shape[1]
    out = torch.empty(r, s + nf, device=dev)
    _lib.check(lib.afx_fine_depths(_ptr(z_coarse), int(z_coarse.dim() == 2), _ptr(w_coarse), _ptr(u), r, s, nf, _ptr(out),
                                   Engine._stream(dev)), "afx_fine_depths")
    return out


def fine_depths_from_tau(z_coarse, tau_coarse, u):
    """afx_fine_depths_from_tau: merged coarse + fine depths from the coarse pass's per-sample optical depths [R,S]."""
    lib = _lib.load()
    dev = tau_coarse.device
    z_coarse, tau_coarse, u = _f32(z_coarse, "z_coarse", dev), _f32(tau_coarse, "tau_coarse", dev), _f32(u, "u", dev)
    r, s = tau_coarse.shape
    nf = u.shape[1]
    out = torch.empty(r, s + nf, device=dev)
    _lib.check(lib.afx_fine_depths_from_tau(_ptr(z_coarse), int(z_coarse.dim() == 2), _ptr(tau_coarse), _ptr(u), r, s, nf, _ptr(out),
                                            Engine._stream(dev)), "afx_fine_depths_from_tau")
    return out


# ---- occupancy grid / ray marching / device ray sampler (no model) ---------------------------------------------
def _grid_desc(roi_aabb, resolution) -> "_lib.GridDesc":
    g = _lib.GridDesc()
    for i, v in enumerate([float(x) for x in roi_aabb]):
        g.roi_aabb[i] = v
    for i, v in enumerate([int(x) for x in resolution]):
        g.resolution[i] = v
    return g


def philox_uniform(seed: int, stream_id: int, n: int, device) -> torch.Tensor:
    """n uniform [0,1) numbers of the counter-based stream (seed, stream_id) - the generator of every perf-mode draw."""
    lib = _lib.load()
    out = torch.empty(int(n), dtype=torch.float32, device=device)
    _lib.check(lib.afx_philox_uniform(int(seed), int(stream_id), int(n), _ptr(out), Engine._stream(out.device)), "afx_philox_uniform")
    return out


def grid_points(roi_aabb, resolution, cell_idx, n, jitter=None, seed=0, stream_id=0, device=None):
    lib = _lib.load()
    dev = device if device is not None else (cell_idx.device if cell_idx is not None else jitter.device)
    if torch.device(dev).type != "cuda":
        raise AfxError("grid_points: the occupancy grid lives on a GPU; there is no CPU fallback")
    g = _grid_desc(roi_aabb, resolution)
    pts = torch.empty(int(n), 3, dtype=torch.float32, device=dev)
    if jitter is not None:
        jitter = _f32(jitter, "jitter", pts.device)
    _lib.check(lib.afx_grid_points(C.byref(g), _ptr(cell_idx), int(n), _ptr(jitter), int(seed), int(stream_id), _ptr(pts),
                                   Engine._stream(pts.device)), "afx_grid_points")
    return pts


def grid_update(roi_aabb, resolution, occs, cell_idx, occ_new, ema_decay, scratch):
    lib = _lib.load()
    g = _grid_desc(roi_aabb, resolution)
    occ_new = _f32(occ_new.reshape(-1), "occ_new", occs.device)
    _lib.check(lib.afx_grid_update(C.byref(g), _ptr(occs), _ptr(scratch), _ptr(cell_idx), occ_new.numel(), _ptr(occ_new),
                                   float(ema_decay), Engine._stream(occs.device)), "afx_grid_update")


def grid_binarize(roi_aabb, resolution, occs, occ_thre, binary_u8, bits, partial_ws):
    lib = _lib.load()
    g = _grid_desc(roi_aabb, resolution)
    _lib.check(lib.afx_grid_binarize(C.byref(g), _ptr(occs), float(occ_thre), _ptr(binary_u8), _ptr(bits), _ptr(partial_ws),
                                     Engine._stream(occs.device)), "afx_grid_binarize")


def grid_pack(roi_aabb, resolution, binary_u8, bits):
    """afx_grid_pack: the march's bitfield from a byte mask (a restored grid)."""
    lib = _lib.load()
    if binary_u8.device.type != "cuda":
        raise AfxError("grid_pack: the occupancy grid lives on a GPU; there is no CPU fallback")
    g = _grid_desc(roi_aabb, resolution)
    _lib.check(lib.afx_grid_pack(C.byref(g), _ptr(binary_u8), _ptr(bits), Engine._stream(binary_u8.device)), "afx_grid_pack")


def _fill_march_args(m, o, d, scene_aabb, near_plane, far_plane, step, grid_bits, grid_aabb, grid_res):
    m.origins, m.dirs, m.n_rays = o.data_ptr(), d.data_ptr(), o.shape[0]
    if scene_aabb is not None:
        m.has_aabb = 1
        for i, v in enumerate([float(x) for x in scene_aabb]):
            m.scene_aabb[i] = v
    if near_plane is not None:
        m.has_near, m.near_plane = 1, float(near_plane)
    if far_plane is not None:
        m.has_far, m.far_plane = 1, float(far_plane)
    m.step = float(step)
    if grid_bits is not None:
        m.grid_bits = grid_bits.data_ptr()
        m.grid = _grid_desc(grid_aabb, grid_res)


def march(origins, dirs, scene_aabb, near_plane, far_plane, step, grid_bits=None, grid_aabb=None, grid_res=None, want_points=True):
    """Grid-skipping fixed-step march -> packed (ray_indices int32 [n], t_starts [n], t_ends [n], mid-points [n,3] | None,
    offsets int64 [R+1])."""
    lib = _lib.load()
    dev = origins.device
    if dev.type != "cuda":
        raise AfxError("march: rays must live on a GPU; there is no CPU fallback")
    o, d = _f32(origins, "origins", dev), _f32(dirs, "dirs", dev)
    m = _lib.MarchArgs()
    _fill_march_args(m, o, d, scene_aabb, near_plane, far_plane, step, grid_bits, grid_aabb, grid_res)
    st = Engine._stream(dev)
    counts = torch.empty(o.shape[0], dtype=torch.int32, device=dev)
    _lib.check(lib.afx_march_count(C.byref(m), _ptr(counts), st), "afx_march_count")
    offsets = torch.empty(o.shape[0] + 1, dtype=torch.int64, device=dev)
    totals = torch.empty(2, dtype=torch.int64, device=dev)
    _lib.check(lib.afx_ray_offsets(_ptr(counts), o.shape[0], _ptr(offsets), None, _ptr(totals), st), "afx_ray_offsets")
    n = int(totals[0])      # the march's one host read
    ri = torch.empty(n, dtype=torch.int32, device=dev)
    ts, te = torch.empty(n, device=dev), torch.empty(n, device=dev)
    pts = torch.empty(n, 3, device=dev) if want_points else None
    if n > 0:
        _lib.check(lib.afx_march_write(C.byref(m), _ptr(offsets), _ptr(ri), _ptr(ts), _ptr(te), _ptr(pts), st), "afx_march_write")
    return ri, ts, te, pts, offsets


def march_visibility(raw, ts, te, offsets, early_stop_eps, alpha_thre, is_alpha=False, return_offsets=False):
    """nerfacc render_visibility on the candidates' raw MLP outputs -> compacted (ray_indices, t_starts, t_ends)."""
    lib = _lib.load()
    dev = ts.device
    n_rays = offsets.numel() - 1
    raw = _f32(raw.reshape(-1), "raw", dev)
    keep = torch.empty(ts.numel(), dtype=torch.uint8, device=dev)
    counts = torch.empty(n_rays, dtype=torch.int32, device=dev)
    st = Engine._stream(dev)
    _lib.check(lib.afx_march_visibility(_ptr(raw), int(bool(is_alpha)), _ptr(ts), _ptr(te), _ptr(offsets), n_rays, float(early_stop_eps),
                                        float(alpha_thre), _ptr(keep), _ptr(counts), st), "afx_march_visibility")
    off2 = torch.empty(n_rays + 1, dtype=torch.int64, device=dev)
    totals = torch.empty(2, dtype=torch.int64, device=dev)
    goff = torch.empty(n_rays + 1, dtype=torch.int64, device=dev) if return_offsets == "groups" else None      # (the group-aligned copy's offsets)
    _lib.check(lib.afx_ray_offsets(_ptr(counts), n_rays, _ptr(off2), _ptr(goff), _ptr(totals), st), "afx_ray_offsets")
    n2, n_groups = totals.tolist()      # one host read for both
    ri2 = torch.empty(n2, dtype=torch.int32, device=dev)
    ts2, te2 = torch.empty(n2, device=dev), torch.empty(n2, device=dev)
    if n2 > 0:
        _lib.check(lib.afx_march_compact(_ptr(keep), _ptr(offsets), _ptr(off2), n_rays, _ptr(ts), _ptr(te), _ptr(ri2), _ptr(ts2),
                                         _ptr(te2), st), "afx_march_compact")
    if return_offsets == "groups":
        return ri2, ts2, te2, off2, goff, int(n_groups)
    if return_offsets:
        return ri2, ts2, te2, off2
    return ri2, ts2, te2


def topk_indices(keys: torch.Tensor, k: int) -> torch.Tensor:
    """Indices (int64, ascending) of the k largest keys - afx_topk_indices: radix select, deterministic, no full sort."""
    lib = _lib.load()
    if keys.device.type != "cuda":
        raise AfxError("topk_indices: keys must live on a GPU; there is no CPU fallback")
    keys = keys.contiguous()
    n = keys.numel()
    if not 0 <= k <= n:
        raise ValueError(f"topk_indices: k = {k} outside [0, {n}]")
    out = torch.empty(int(k), dtype=torch.int64, device=keys.device)
    if k == 0:
        return out
    nbytes = int(lib.afx_topk_workspace_bytes(n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=keys.device)
    _lib.check(lib.afx_topk_indices(_ptr(keys), n, int(k), _ptr(out), _ptr(ws), nbytes, Engine._stream(keys.device)), "afx_topk_indices")
    return out


class RayBatchSampler:
    """sample_rays for a training loop: the batches of `prefetch` consecutive iterations are drawn by ONE launch sequence
    (afx_sample_batches; a single draw is launch latency, ~70 us of the reference's 1.3 ms iteration) and handed out one per call.
    draw(stream_id) returns exactly what sample_rays(origins, dirs, pixels, weights, k, seed=seed, stream_id=stream_id) returns; stream ids
    are expected to advance by one per iteration (any other id starts a new block of `prefetch` draws at that id)."""

    def __init__(self, origins, dirs, pixels, weights, k, seed=0, prefetch=16):
        dev = origins.device
        if dev.type != "cuda":
            raise AfxError("RayBatchSampler: the ray table must live on a GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.origins, self.dirs, self.weights = _f32(origins, "origins", dev), _f32(dirs, "dirs", dev), _f32(weights, "weights", dev)
        self.pixels = _f32(pixels, "pixels", dev) if pixels is not None else None
        n = self.origins.shape[0]
        if tuple(self.origins.shape) != (n, 3) or tuple(self.dirs.shape) != (n, 3) or self.weights.numel() != n \
                or (self.pixels is not None and self.pixels.numel() != n):
            raise ValueError("RayBatchSampler: expected origins/dirs [n,3], pixels/weights [n]")
        if not 0 < int(k) <= n or not 1 <= int(prefetch) <= 65535:
            raise ValueError("RayBatchSampler: need 0 < k <= n and 1 <= prefetch <= 65535")
        self.n, self.k, self.seed, self.prefetch, self.dev = n, int(k), int(seed), int(prefetch), dev
        self._ws = torch.empty(int(self.lib.afx_sample_batches_workspace_bytes(n, self.prefetch)), dtype=torch.uint8, device=dev)
        self._idx = torch.empty(self.prefetch, self.k, dtype=torch.int64, device=dev)
        self._first = None      # stream id of row 0 of _idx

    def draw(self, stream_id):
        stream_id = int(stream_id)
        st = Engine._stream(self.dev)
        if self._first is None or not self._first <= stream_id < self._first + self.prefetch:
            _lib.check(self.lib.afx_sample_batches(_ptr(self.weights), self.n, self.seed, stream_id, self.prefetch, self.k, _ptr(self._idx),
                                                   _ptr(self._ws), self._ws.numel(), st), "afx_sample_batches")
            self._first = stream_id
        idx = self._idx[stream_id - self._first]
        o, d = torch.empty(self.k, 3, device=self.dev), torch.empty(self.k, 3, device=self.dev)
        p = torch.empty(self.k, device=self.dev) if self.pixels is not None else None
        _lib.check(self.lib.afx_gather_rays(_ptr(self.origins), _ptr(self.dirs), _ptr(self.pixels), _ptr(idx), self.k, _ptr(o), _ptr(d), _ptr(p), st),
                   "afx_gather_rays")
        return o, d, p, idx


def sample_rays(origins, dirs, pixels, weights, k, u=None, seed=0, stream_id=0):
    """Weighted sample WITHOUT replacement of k rows of a device-resident ray table (sample_pixel_rays, nerf_helpers.py:137-150):
    Efraimidis-Spirakis keys log(u)/w, radix-select top-k on the device (afx_topk_indices), gather.  Returns (origins[k,3], dirs[k,3], pixels[k] | None, idx)."""
    lib = _lib.load()
    dev = origins.device
    if dev.type != "cuda":
        raise AfxError("sample_rays: the ray table must live on a GPU; there is no CPU fallback")
    # (row-major [n,3] tables: a frame's `df[[x, y, z]].to_numpy()` is usually column-major, and .float().to(device) keeps those strides)
    origins, dirs, weights = _f32(origins, "origins", dev), _f32(dirs, "dirs", dev), _f32(weights, "weights", dev)
    if pixels is not None:
        pixels = _f32(pixels, "pixels", dev)
    if u is not None:
        u = _f32(u, "u", dev)
    n = origins.shape[0]
    if tuple(origins.shape) != (n, 3) or tuple(dirs.shape) != (n, 3) or weights.numel() != n or (pixels is not None and pixels.numel() != n) \
            or (u is not None and u.numel() != n):
        raise ValueError("sample_rays: expected origins/dirs [n,3], pixels/weights/u [n]")
    keys = torch.empty(n, device=dev)
    st = Engine._stream(dev)
    _lib.check(lib.afx_sample_keys(_ptr(weights), n, _ptr(u), int(seed), int(stream_id), _ptr(keys), st), "afx_sample_keys")
    idx = topk_indices(keys, int(k))      # ascending table order: the batch is a set
    o, d = torch.empty(int(k), 3, device=dev), torch.empty(int(k), 3, device=dev)
    p = torch.empty(int(k), device=dev) if pixels is not None else None
    _lib.check(lib.afx_gather_rays(_ptr(origins), _ptr(dirs), _ptr(pixels), _ptr(idx), int(k), _ptr(o), _ptr(d), _ptr(p), st),
               "afx_gather_rays")
    return o, d, p, idx
