#!/usr/bin/env python3
"""Compile the HIP library with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re, subprocess, sys
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-c", "-o", "/tmp/kres.o",
                      "nerf_for_angiography_amd/csrc/afx_api.hip", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(": ")[1]; rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, r in rows.items():
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dn: continue
    print(f"{dn[:70]:70s} V={r.get('VGPRs')} A={r.get('AGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} vspill={r.get('VGPRs Spill')} sspill={r.get('SGPRs Spill')} occ={r.get('Occupancy [waves/SIMD]')}")
