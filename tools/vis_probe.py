#!/usr/bin/env python3
"""What bounds k_march_visibility?  R rays x S candidates each, HIP-event timing of afx_march_visibility alone: S sweep, raw vs ready-made alpha
(no transcendental work), early stop on / off.  Run under `rocprofv3 --kernel-trace` for the kernel's own time.
Findings: 12.7 / 21 / 38.6 / 86 us for S = 32 / 64 / 128 / 300 with the original loop (two branches and a ds_bpermute per sample: latency-bound);
6.6 / 6.6 / 11 / 18 us with the branch-free in-order product on LDS-broadcast factors.  (Three rewrites first "measured" exactly the old time:
csrc/afx_kernels_grid.hip was missing from build.py's dependency list, so none of them had been compiled.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nerf_for_angiography_amd import engine as eng
dev = torch.device("cuda:0")
R = 5625
for S in (32, 64, 128, 300):
    for is_alpha, eps in ((False, 1e-2), (True, 1e-2), (False, 0.0)):
        n = R * S
        raw = (torch.randn(n, device=dev) - 5.0) if not is_alpha else torch.rand(n, device=dev) * 0.01
        ts = torch.arange(S, device=dev, dtype=torch.float32).repeat(R) * 0.66 + 1400
        te = ts + 0.66
        off = torch.arange(R + 1, device=dev, dtype=torch.int64) * S
        for _ in range(3): eng.march_visibility(raw, ts, te, off, eps, 1e-4, is_alpha=is_alpha)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # time the whole wrapper minus nothing: dominated by the visibility kernel + offsets + compact; so also time a no-op pair
        t0 = time.perf_counter(); a.record()
        for _ in range(20): out = eng.march_visibility(raw, ts, te, off, eps, 1e-4, is_alpha=is_alpha)
        b.record(); torch.cuda.synchronize()
        print(f"S {S:4d} is_alpha {int(is_alpha)} eps {eps:g}: {a.elapsed_time(b) / 20 * 1e3:7.1f} us per call (wrapper: visibility + offsets + compact), kept {out[0].numel()}", flush=True)
