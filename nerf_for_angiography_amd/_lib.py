"""ctypes binding of libafx.so (include/afx.h).  There is NO fallback: if the HIP library is
missing or a call fails, an exception is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libafx.so")

ACT = {"relu": 0, "tanh": 1, "sine": 2}
ENC = {"none": 0, "barf": 1, "fourier": 2}
PREC = {"f32": 0, "bf16x3": 1, "bf16": 2, "f16": 3, "f16s8": 4}
RAYS_ARRAYS, RAYS_POSE = 0, 1
DEPTH_UNIFORM_MID, DEPTH_SHARED_Z, DEPTH_PER_RAY_Z, DEPTH_STRATIFIED = 0, 1, 2, 3
Q_PARAM_COUNT, Q_K0, Q_PREPARED_BYTES, Q_FWD_WORKSPACE, Q_BWD_WORKSPACE_MIN, Q_BWD_WORKSPACE_FULL = range(6)


class AfxError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [("n_in", C.c_int32), ("enc", C.c_int32), ("n_freq", C.c_int32), ("width", C.c_int32),
                ("n_hidden", C.c_int32), ("act", C.c_int32), ("act_w0", C.c_float)]


class RenderArgs(C.Structure):
    _fields_ = [("n_rays", C.c_int64), ("n_samples", C.c_int32), ("ray_mode", C.c_int32),
                ("origins", C.c_void_p), ("dirs", C.c_void_p), ("poses", C.c_void_p), ("ray_ids", C.c_void_p),
                ("ray_id0", C.c_int64), ("width", C.c_int32), ("height", C.c_int32), ("focal", C.c_double),
                ("depth_mode", C.c_int32), ("t_near", C.c_float), ("t_far", C.c_float), ("z", C.c_void_p),
                ("pixel", C.c_void_p), ("sigma", C.c_void_p), ("tau", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("jitter_seed", C.c_uint64), ("jitter_stream", C.c_uint64)]


class GridDesc(C.Structure):
    _fields_ = [("roi_aabb", C.c_float * 6), ("resolution", C.c_int32 * 3)]


class MarchArgs(C.Structure):
    _fields_ = [("origins", C.c_void_p), ("dirs", C.c_void_p), ("n_rays", C.c_int64), ("has_aabb", C.c_int32),
                ("scene_aabb", C.c_float * 6), ("has_near", C.c_int32), ("has_far", C.c_int32), ("near_plane", C.c_float),
                ("far_plane", C.c_float), ("step", C.c_float), ("grid_bits", C.c_void_p), ("grid", GridDesc)]


class MarchTrainArgs(C.Structure):
    _fields_ = [("march", MarchArgs), ("early_stop_eps", C.c_float), ("alpha_thre", C.c_float), ("target", C.c_void_p), ("inv_n", C.c_float),
                ("pixel", C.c_void_p), ("grad_flat", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("n_candidates", C.c_int64), ("n_kept", C.c_int64), ("n_groups", C.c_int64), ("workspace_needed", C.c_size_t)]


_SIGS = {
    "afx_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "afx_destroy": (None, [C.c_void_p]),
    "afx_last_error": (C.c_char_p, []),
    "afx_query": (C.c_int64, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64]),
    "afx_param_layout": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "afx_prepare_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "afx_mlp_infer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "afx_mlp_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "afx_render_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(RenderArgs), C.c_void_p]),
    "afx_render_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(RenderArgs), C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "afx_train_step_mse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(RenderArgs), C.c_void_p, C.c_float, C.c_void_p,
                                     C.c_void_p]),
    "afx_composite_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_composite_dense_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_composite_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                       C.c_void_p]),
    "afx_composite_packed_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_project_volume": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.c_float, C.POINTER(RenderArgs), C.c_int, C.c_void_p]),
    "afx_topk_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "afx_topk_indices": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "afx_sample_batches_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "afx_sample_batches": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "afx_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "afx_set_encoding_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "afx_fine_depths": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p]),
    "afx_fine_depths_from_tau": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p]),
    "afx_grid_points": (C.c_int, [C.POINTER(GridDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "afx_grid_update": (C.c_int, [C.POINTER(GridDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_void_p]),
    "afx_grid_binarize": (C.c_int, [C.POINTER(GridDesc), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_grid_pack": (C.c_int, [C.POINTER(GridDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_hier_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32]),
    "afx_hier_train_step_mse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(RenderArgs), C.c_int32, C.c_void_p, C.c_void_p, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_pack_groups": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_train_step_packed_mse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "afx_march_count": (C.c_int, [C.POINTER(MarchArgs), C.c_void_p, C.c_void_p]),
    "afx_march_write": (C.c_int, [C.POINTER(MarchArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_march_visibility": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_march_train_step_mse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(MarchTrainArgs), C.c_void_p]),
    "afx_ray_offsets": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_march_compact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "afx_sample_keys": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "afx_gather_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "afx_philox_uniform": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
}

_libs = {}


def exported_symbols():
    """Every entry point include/afx.h declares."""
    return sorted(_SIGS)


def load(variant: str = ""):
    """Load libafx.so (or a build variant such as the race-detector build "safe" = libafx_safe.so); raise loudly when
    it has not been built."""
    if variant in _libs:
        return _libs[variant]
    # torch first: it ships its own libamdhip64.so.7; libafx.so must bind to THAT runtime (the one torch's
    # allocator and streams live in), not to a second copy pulled in through its RUNPATH.
    import torch  # noqa: F401
    path = LIB_PATH if not variant else os.path.join(_HERE, "csrc", f"libafx_{variant}.so")
    if not os.path.exists(path):
        raise AfxError(f"HIP library not built: {path} is missing. Run `python -m nerf_for_angiography_amd.build` "
                       "(needs hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _libs[variant] = lib
    return lib


def check(rc, what="afx call", lib=None):
    if rc != 0:
        msg = (lib or load()).afx_last_error()
        raise AfxError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
