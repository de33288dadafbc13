"""Build the HIP shared library in-tree (gfx950 only).  `python -m nerf_for_angiography_amd.build`

The library is several translation units compiled in parallel (the fused chain kernels are large: one unit per layer
width and direction) and linked into csrc/libafx.so.  `variant` builds a second library from the same sources with extra
defines (e.g. the race-detector build "safe": every counted wait replaced by a full one)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
DEPS = ["afx_api.hip", "afx_kernels_f32.hip", "afx_kernels_bf16.hip", "afx_kernels_grid.hip", "afx_inst.h", "afx_inst_chain16.hip",
        "afx_internal.h", os.path.join("..", "..", "include", "afx.h")]
VARIANTS = {"": [], "safe": ["-DAFX_SAFE_WAITS"],
            "window": ["-DAFX_STASH_WINDOW"],     # measurement: stash stores into an L2-resident window (no HBM write stream; wrong results)
            "nogmax": ["-DAFX_NO_GMAX_ATOMIC"],      # measurement: the backward kernels without the max|g| atomics (wrong gradients; what do the remaining atomics cost?)
            "p2one": ["-DAFX_P2_OCC2=0"],      # A/B: the backward half of the split step with ONE workgroup per CU at widths <= 128
            "h6": ["-DAFX_H6=1"],      # the 6-bit (bf6 + block scales) H stash: 12.5 % fewer stash bytes, no faster; tests/ compare it with the default
            "h6c": ["-DAFX_H6=1", "-DAFX_H6_CONST"],  # measurement: ... without the scale computation (scale 1; gradients wrong when H leaves [1/16, 28])
            "gaps": ["-DAFX_GAPS=1"],      # A/B: the 8-bit-stash backward kernel with the MFMA-gap schedule (slower)
            "stamp": ["-DAFX_STAMP", "-DAFX_SINGLE_TU"],
            "stamp128": ["-DAFX_STAMP", "-DAFX_STAMP_F=128", "-DAFX_SINGLE_TU"]}      # ... of the width-128 kernels      # diagnostic: per-phase cycle stamps of the backward chain kernel (one translation unit)


def lib_path(variant: str = "") -> str:
    return os.path.join(CSRC, "libafx.so" if not variant else f"libafx_{variant}.so")


LIB = lib_path()


def _units():
    """(object name, source, extra defines)"""
    out = [("api", "afx_api.hip", [])]
    for f in (64, 128, 256):
        for bwd in (0, 1, 2):
            out.append((f"chain16_w{f}_{('fwd', 'bwd', 'phases')[bwd]}", "afx_inst_chain16.hip", [f"-DAFX_INST_F={f}", f"-DAFX_INST_BWD={bwd}"]))
    return out


def _newest_dep() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, d)) for d in DEPS if os.path.exists(os.path.join(CSRC, d)))


def _stale(variant: str = "") -> bool:
    lib = lib_path(variant)
    return not os.path.exists(lib) or os.path.getmtime(lib) < _newest_dep()


def build(force: bool = False, verbose: bool = False, variant: str = "", jobs: int = 0) -> str:
    lib = lib_path(variant)
    if not force and not _stale(variant):
        return lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"] + VARIANTS[variant]
    objdir = os.path.join(CSRC, "_build", variant or "default")
    os.makedirs(objdir, exist_ok=True)
    newest = _newest_dep()

    def compile_one(unit):
        name, src, defs = unit
        obj = os.path.join(objdir, name + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            return obj
        cmd = [hipcc] + flags + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    jobs = jobs or min(8, os.cpu_count() or 1)
    units = [u for u in _units() if u[0] == "api"] if "-DAFX_SINGLE_TU" in VARIANTS[variant] else _units()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, units))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    v = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")]
    print(build(force="--force" in sys.argv, verbose=True, variant=v[0] if v else ""))
