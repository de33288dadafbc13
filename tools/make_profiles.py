#!/usr/bin/env python3
"""profiles/<tag>_<prec>_kernel_stats.md, profiles/<tag>_<prec>_pmc.md, profiles/<tag>_pmc_traffic.json and
profiles/<tag>_bench_<prec>.json from the reduced rocprofv3 outputs in gpurun_out/ (written by tools/profile_round.sh
through tools/pmc_reduce.py).  usage: make_profiles.py [prec=f16s8] [tag=r03]"""
import json, os, subprocess, sys

P = sys.argv[1] if len(sys.argv) > 1 else "f16s8"
T = sys.argv[2] if len(sys.argv) > 2 else "r03"
G = "gpurun_out"
load = lambda n: json.load(open(f"{G}/{T}_{P}_{n}.json"))
st = load("stats")["kernel_stats"]
fe, wr = load("fetch")["counters"], load("write")["counters"]
bench = json.load(open(f"{G}/{T}_bench_{P}.json")) if os.path.exists(f"{G}/{T}_bench_{P}.json") else None
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
os.makedirs("profiles", exist_ok=True)
SAMPLES, F, N = 512 * 512 * 128, 256, 8
esz = {"f16s8": 1, "f16": 2, "bf16": 2}[P]
extra = {"f16s8": 4 / 32, "f16": 4, "bf16": 0}[P]          # group exponents / dL/draw per sample
WG = {"f16s8": "k_wgrad_s8", "f16": "k_wgrad_bf16", "bf16": "k_wgrad_bf16"}[P]


def find(d, key):
    return next((k for k in d if key in k), None)


hot = [r for r in st if r["Name"].startswith(("void k_", "k_"))]
kc = find({r["Name"]: 1 for r in st}, "k_chain_bf16")
calls = next(int(r["Calls"]) for r in st if r["Name"] == kc)
chunks = calls // 4            # the stats pass runs 1 warm-up + 3 timed steps
with open(f"profiles/{T}_{P}_kernel_stats.md", "w") as f:
    f.write(f"# Round {T[1:]} - rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu --precision {P}` (fused train step), commit {commit}\n\n")
    f.write(f"MI355X, 512x512 projection x 128 samples, 8x256 CPPN, 1 warm-up + 3 timed steps ({chunks} ray chunks per step, 128 GiB workspace).\n")
    f.write("Hot-path kernels only (the rest of the trace is the synthetic phantom's set-up and PyTorch's Adam / loss).\n\n| kernel | calls | total ms | avg ms |\n|---|---|---|---|\n")
    for r in hot[:12]:
        f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} |\n")
    if bench:
        k = bench["roofline"]
        f.write(f"\nbench.py (same command, un-profiled run on the same box) times the same kernels with HIP events on the launch stream: "
                f"k_chain<bwd> {k['avg_launch_ms']} ms per launch, kernel ms per step {k['kernel_ms_per_step']}; step {bench['ms_per_step']} ms.\n")

tr = {}
with open(f"profiles/{T}_{P}_pmc.md", "w") as f:
    f.write(f"# Round {T[1:]} - HBM traffic and SQ counters from rocprofv3 PMC passes ({P} fused train step), commit {commit}\n\n")
    f.write("Separate passes (`--kernel-trace --pmc FETCH_SIZE`, `--kernel-trace --pmc WRITE_SIZE`, two SQ passes) of\n"
            f"`python bench.py --no-cpu --steps 1 --warmup 1 --precision {P}` (tools/profile_round.sh).  Counter unit: KiB.  Per MI355X_MICROARCH.md: on gfx950\n")
    f.write("FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (doubled below); WRITE_SIZE is exact for 16-B/lane stores.\n\n")
    f.write("| kernel | launches | FETCH_SIZE/launch (GB, x2-corrected) | WRITE_SIZE/launch (GB) | algorithmic bytes/launch (GB) |\n|---|---|---|---|---|\n")
    nl = fe[find(fe, "k_chain_bf16")]["FETCH_SIZE"]["rows"]
    chunk = SAMPLES * 2 / nl            # 2 steps in the PMC runs
    alg = {"k_chain_bf16": chunk * (2 * N * F * esz + (3 * F + 8) * 4 / 32 + extra) / 1e9, WG: chunk * (2 * N * F * esz + extra) / 1e9,
           "k_small_from_groups": chunk * ((3 * F + 8) * 4 / 32) / 1e9}
    cw = wf = 0.0
    for k in fe:
        n = fe[k]["FETCH_SIZE"]["rows"]
        fg = fe[k]["FETCH_SIZE"]["sum"] * 1024 * 2 / n / 1e9
        wg = wr.get(k, {}).get("WRITE_SIZE", {"sum": 0})["sum"] * 1024 / n / 1e9
        a = next((v for kk, v in alg.items() if kk in k), None)
        f.write(f"| `{k[:70]}` | {n} | {fg:.3f} | {wg:.3f} | {'' if a is None else f'{a:.2f}'} |\n")
        if "k_chain_bf16" in k: tr["chain_bwd"] = (fg + wg) * 1e9; cw = wg
        if WG in k: tr["wgrad"] = (fg + wg) * 1e9; wf = fg
    f.write(f"\nk_chain<bwd> writes the stash exactly once: measured {cw:.1f} GB (WRITE_SIZE) vs {alg['k_chain_bf16']:.1f} GB algorithmic per "
            f"{chunk/1e6:.2f} M-sample chunk; {WG} reads it back once ({wf:.1f} GB after the x2 correction vs {alg[WG]:.1f} GB algorithmic). No re-reads.\n")
    for name in ("sq", "sq2"):
        if not os.path.exists(f"{G}/{T}_{P}_{name}.json"):
            continue
        sq = load(name)["counters"]
        f.write(f"\n## SQ counters per launch ({name}; sums over all CUs/SIMDs; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_* count quad-cycles, "
                "SQ_VALU_MFMA_BUSY_CYCLES cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs)\n\n")
        ctrs = sorted({c for k in sq for c in sq[k]})
        f.write("| kernel | " + " | ".join(ctrs) + " |\n|---|" + "---|" * len(ctrs) + "\n")
        for k in sq:
            if not any(t in k for t in ("k_chain_bf16", WG, "k_small_from_groups")):
                continue
            f.write(f"| `{k[:60]}` | " + " | ".join(f"{sq[k][c]['sum'] / sq[k][c]['rows']:.4g}" if c in sq[k] else "" for c in ctrs) + " |\n")
        for key, mfma_cyc, what in (("k_chain_bf16", chunk / 32 * 2 * 8 * 8 * 16 * 32, "samples/32 x 2 directions x 8 layers x 8 tiles x 16 MFMAs x 32 cycles"),
                                    (WG, chunk / 64 * 8 * 72 * (64 if P == "f16s8" else 128), "stages x 8 layers x 72 MFMAs per stage x cycles")):
            kk = find(sq, key)
            if kk and "SQ_VALU_MFMA_BUSY_CYCLES" in sq[kk]:
                per = lambda x: sq[kk][x]["sum"] / sq[kk][x]["rows"]
                f.write(f"\n{key}: SQ_VALU_MFMA_BUSY_CYCLES / launch = {per('SQ_VALU_MFMA_BUSY_CYCLES'):.4g}; algorithmic MFMA cycles ({what}) = {mfma_cyc:.4g}.\n")
            if kk and "GRBM_GUI_ACTIVE" in sq[kk]:
                ms = next((float(r["AverageNs"]) / 1e6 for r in hot if key in r["Name"]), None)
                if ms:
                    f.write(f"\n{key}: effective clock = GRBM_GUI_ACTIVE / 8 / launch time = {sq[kk]['GRBM_GUI_ACTIVE']['sum'] / sq[kk]['GRBM_GUI_ACTIVE']['rows'] / 8 / (ms * 1e-3) / 1e9:.2f} GHz "
                            f"(launch {ms:.2f} ms in the stats pass).\n")
json.dump({P: {"bytes_per_launch": tr["chain_bwd"], "kernel": "k_chain<bwd>", "commit": commit,
               "source": f"profiles/{T}_{P}_pmc.md (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, commit {commit})",
               "wgrad_bytes_per_launch": tr["wgrad"]}}, open(f"profiles/{T}_pmc_traffic.json", "w"), indent=1)
if bench:
    open(f"profiles/{T}_bench_{P}.json", "w").write(json.dumps(bench) + "\n")
print(open(f"profiles/{T}_{P}_pmc.md").read()[-3000:])
