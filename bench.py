#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

A "step" is one training iteration over one full synthetic 512x512 projection: 262 144 rays x 128
samples/ray through the 8x256 CPPN, fused forward (ray generation from the C-arm pose -> uniform
mid-point sampling -> MLP -> Beer-Lambert product), MSE loss, fused backward, gradient all-reduce
(N > 1) and the PyTorch Adam step — the body of nerf/run_nerf_acc.py:263-307 upstream.
Weak scaling: every rank renders its own projection each step.

    python bench.py --gpus N --steps K --warmup W [--precision f32|bf16x3|bf16]

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit, ms_per_step,
`roofline` for the dominant kernel (HIP-event timed inside the library on the launch stream) and
`cpu_baseline` (the CPU oracle timed on this host's cores on a bounded ray sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}     # MI355X_MICROARCH.md: dense MFMA peaks
DTYPE_NAME = {"f32": "f32", "bf16x3": "bf16x3 (split bf16, f32 accumulate)", "bf16": "bf16 (f32 accumulate)"}


def flops_per_sample(width, layers, k0=3):
    """Algorithmic FLOPs per ray-sample: MAC x 2, no padding, no recomputation."""
    fwd = 2 * (k0 * width + layers * width * width + width)
    dgrad = 2 * (layers * width * width + width)
    wgrad = fwd
    return fwd, dgrad, wgrad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default=os.environ.get("AFX_BENCH_PRECISION", "bf16"), choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--cpu-rays", type=int, default=2048)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workspace-gib", type=float, default=128.0,
                    help="backward stash workspace per GPU (288 GB HBM: 128 GiB holds the 512^2x128 projection in 3 ray chunks)")
    ap.add_argument("--unfused", action="store_true",
                    help="render -> mse_loss -> autograd backward (forward rendered separately) instead of the fused train step")
    args = ap.parse_args()

    from nerf_for_angiography_amd import dist as afx_dist
    from nerf_for_angiography_amd.model.CPPN import CPPN
    from nerf_for_angiography_amd.render import render_projection, render_rays, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.proj_helpers import source_matrix
    from nerf_for_angiography_amd.phantomdata.helpers import capsule_tree, capsule_mu, ray_tracing_fn as ray_tracing, get_ray_values

    rank, world, device = afx_dist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    W = H = args.res
    S = args.samples
    focal, near, far = 13.0 * W, 1400.0, 1600.0
    torch.manual_seed(0)
    np.random.seed(0)
    md = dict(num_early_layers=args.layers, num_late_layers=0, num_filters=args.width, num_input_channels=3,
              num_output_channels=1, num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5,
              act_func="relu", fourier_sigma=5, num_img=1, device=device, precision=args.precision)
    model = CPPN(md).to(device)
    with torch.no_grad():        # non-degenerate densities at raw world coordinates (random init otherwise saturates)
        model.output_linear[0].weight.mul_(4.0)
        model.output_linear[0].bias.fill_(-5.0)
    model.engine.max_workspace_bytes = int(args.workspace_gib * (1 << 30))
    afx_dist.broadcast_parameters(model)
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}     # parity is reported at these (seeded) weights
    afx_dist.GradSync().install()
    opt = torch.optim.Adam(list(model.parameters()), lr=1e-4)

    # synthetic P-ANGIO phantom: 31-capsule vessel tree, mu = 0.2, targets by the GT projector
    caps = capsule_tree(levels=5, seed=0)
    n_proj = args.steps + args.warmup
    poses, targets = [], []
    z_gt = torch.linspace(0., 1., 160, device=device) * (far - near) + near
    for i in range(n_proj):
        theta = 2.0 * (i * world + rank)
        o, d, m44, _, _ = get_ray_values(theta, 0.0, 0.0, np.array([0, 0, 1500.0]), W, H, focal, device)
        poses.append(torch.from_numpy(m44[None]).to(device))
        with torch.no_grad():
            targets.append(ray_tracing(lambda p: capsule_mu(p, caps), o.reshape(-1, 3).float(), d.reshape(-1, 3).float(),
                                       z_gt, batch_rays=16384).reshape(-1).contiguous())
    del o, d

    fused = not args.unfused and args.precision != "f32"

    def step(i):
        opt.zero_grad(set_to_none=True)
        if fused:       # forward + MSE + backward in one pass per ray chunk (afx_train_step_mse)
            loss, _ = train_step_mse(model, projection_spec(poses[i], W, H, focal, S, near, far), targets[i],
                                     n_global=W * H)
        else:
            out = render_projection(model, poses[i], W, H, focal, S, near, far)
            loss = torch.nn.functional.mse_loss(out.rgb_map, targets[i])
            loss.backward()
        opt.step()
        return loss

    # the backward workspace is allocated before any timing (also with --warmup 0): a 128 GiB hipMalloc is setup, not a step
    if fused:
        from nerf_for_angiography_amd import _lib as afx_lib
        full = int(model.engine.lib.afx_query(model.engine.h, afx_lib.Q_BWD_WORKSPACE_FULL, 0, W * H * ((S + 31) // 32 * 32),
                                              afx_lib.PREC[args.precision]))
        model.engine._workspace(min(full, model.engine.max_workspace_bytes), device)
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    model.engine.profile(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.engine.profile(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    samples_per_step = W * H * S
    value = world * samples_per_step * args.steps / elapsed
    fwd_f, dgrad_f, wgrad_f = flops_per_sample(args.width, args.layers)
    prof = {k: model.engine.profile_read(k) for k in ("chain_fwd", "chain_bwd", "wgrad")}
    peak = PEAK_TFLOPS[args.precision]
    dom_ms, dom_n = prof["chain_bwd"]
    per_launch_samples = samples_per_step * args.steps / max(dom_n, 1)
    avg_s = dom_ms / max(dom_n, 1) * 1e-3
    # k_chain<bwd> does the forward (fused: it IS the step's forward pass; unfused: a recompute that is not
    # counted) and the input-gradient chain.  Algorithmic FLOPs per sample of that launch:
    alg = (fwd_f + dgrad_f) if fused else dgrad_f
    achieved = alg * per_launch_samples / avg_s / 1e12 if dom_ms > 0 else 0.0
    esz = 4 if args.precision == "f32" else 2
    stash_bytes = 2 * (args.layers + 1) * args.width * esz          # H_l and dZ_l written per sample by that launch
    wgrad_bytes = 2 * args.layers * args.width * esz                # H_{l-1}, dZ_l of the hidden layers read back per sample
    if args.precision != "f32" and os.environ.get("AFX_SMALL_IN_KERNEL", "1") != "0":
        # bf16 rays mode: H_N and dZ_0 are not stashed; per 32-sample group 3 x width + 8 floats of partial sums instead
        stash_bytes = wgrad_bytes + (3 * args.width + 8) * 4 / 32
    roofline = {"bound": "mfma", "kernel": "k_chain<bwd>: forward + Beer-Lambert + input-gradient chain + stash of H_l, dZ_l (+ first/output-layer group sums)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": None, "launches": dom_n, "avg_launch_ms": round(avg_s * 1e3, 3),
                "algorithmic_flop_per_sample": alg,
                "hbm_write_GBps_algorithmic": round(stash_bytes * per_launch_samples / avg_s / 1e9, 1),
                "kernel_ms_per_step": {k: round(v[0] / args.steps, 2) for k, v in prof.items()},
                "wgrad_hbm_read_GBps_algorithmic": round(wgrad_bytes * samples_per_step / max(prof["wgrad"][0] / args.steps * 1e-3, 1e-9) / 1e9, 1),
                "step_tflops_algorithmic": round((fwd_f + dgrad_f + wgrad_f) * value / world / 1e12, 2),
                "step_frac_of_peak": round((fwd_f + dgrad_f + wgrad_f) * value / world / 1e12 / peak, 4)}
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r01_bf16_pmc.md);
    # only valid for the configuration those passes ran (the default one)
    traffic_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    default_cfg = (W, S, args.layers, args.width, fused, abs(args.workspace_gib - 128.0) < 1e-9) == (512, 128, 8, 256, True, True)
    if default_cfg and os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file)).get(args.precision)
            if t:
                roofline["traffic"] = float(t["bytes_per_launch"])
                roofline["traffic_source"] = t["source"]
                roofline["algorithmic_bytes_per_launch"] = float(stash_bytes * per_launch_samples)
        except Exception:
            pass

    result = {"metric": "ray-samples/sec (fwd+bwd)", "value": round(value, 1), "unit": "ray-samples/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
              "config": {"workload": f"{W}x{H} projection, {S} samples/ray, {args.layers}x{args.width} CPPN MLP, "
                                     "uniform mid-point march (acc convention), fwd+bwd+Adam, one projection per GPU per step"
                                     + (", fused train step" if fused else ", render + autograd"),
                         "rays_per_step_per_gpu": W * H, "parallelism": f"ray-batch dp{world}"},
              "final_loss": round(float(loss), 6), "roofline": roofline}

    if rank == 0 and world == 1 and not args.no_cpu:
        model.load_state_dict(init_state)       # a well-defined state: the trained weights depend on the chunking's summation order
        result.update(cpu_leg(model, args, poses[0], targets[0], W, H, focal, near, far, S, device, render_rays))
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_leg(model, args, pose, target, W, H, focal, near, far, S, device, render_rays):
    """CPU oracle (kind "port") timed on this host on a bounded ray sample + GPU-vs-oracle parity on those rays."""
    from oracle import angio_oracle as orc
    g = torch.Generator().manual_seed(1234)
    pick = torch.randperm(W * H, generator=g)[:args.cpu_rays]
    o_all, d_all = orc.get_rays(pose[0].cpu().numpy(), W, H, focal)
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    tgt = target.cpu()[pick]
    cfg = dict(num_early_layers=args.layers, num_filters=args.width)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    pix_by_prec = {}
    keep = model.precision
    with torch.no_grad():
        for prec in dict.fromkeys([keep, "bf16x3", "bf16"]):
            model.precision = prec
            pix_by_prec[prec] = render_rays(model, o.to(device), d.to(device), S, near, far, mode="acc").rgb_map.cpu()
    model.precision = keep
    pix_gpu = pix_by_prec[keep]
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = min(ncpu, int(os.environ.get("AFX_CPU_THREADS", "16")))   # a 1-GPU box owns a 16-core share of the host
    torch.set_num_threads(ncpu)
    leaves = {k: v.requires_grad_(True) for k, v in params.items() if k.startswith(("early", "output"))}
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-4)
    times, pix_cpu = [], None
    for it in range(3):
        t0 = time.perf_counter()
        opt.zero_grad()
        pix = orc.render_rays(o, d, cfg, params, near=near, far=far, n_samples=S, convention="acc")
        torch.nn.functional.mse_loss(pix, tgt).backward()
        if it == 0:
            pix_cpu = pix.detach().clone()
        opt.step()
        times.append(time.perf_counter() - t0)
    best = min(times[1:]) if len(times) > 1 else times[0]
    err = pix_gpu - pix_cpu
    rel = float(err.norm() / pix_cpu.norm())
    mse = float((err.double() ** 2).mean())
    return {"cpu_baseline": {"value": round(args.cpu_rays * S / best, 1), "unit": "ray-samples/s",
                             "cores": torch.get_num_threads(), "kind": "port",
                             "sample": f"{args.cpu_rays} rays x {S} samples of the same projection, {args.layers}x{args.width} MLP, "
                                       "fp32 PyTorch-CPU oracle, fwd+bwd+Adam, best of 2 timed steps after 1 warm-up"},
            "parity_vs_cpu_oracle": {"rays": args.cpu_rays, "weights": "the seeded initial weights (not the trained ones)",
                                     # rendered projections / density grids are produced at render_precision (the training
                                     # driver's --eval_precision default); the timed training step runs at train_precision
                                     "render_precision": "bf16x3" if keep != "f32" else "f32",
                                     "rel_l2": float((pix_by_prec["bf16x3" if keep != "f32" else "f32"] - pix_cpu).norm() / pix_cpu.norm()),
                                     "psnr_db": round(-10 * np.log10(max(float(((pix_by_prec["bf16x3" if keep != "f32" else "f32"] - pix_cpu).double() ** 2).mean()), 1e-30)), 2),
                                     "train_precision": keep, "train_precision_rel_l2": rel,
                                     "train_precision_psnr_db": round(-10 * np.log10(max(mse, 1e-30)), 2),
                                     "rel_l2_by_precision": {k: float((v - pix_cpu).norm() / pix_cpu.norm())
                                                             for k, v in pix_by_prec.items()}}}


if __name__ == "__main__":
    main()
