#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

A "step" is one training iteration over one full synthetic 512x512 projection: 262 144 rays x 128
samples/ray through the 8x256 CPPN, fused forward (ray generation from the C-arm pose -> uniform
mid-point sampling -> MLP -> Beer-Lambert product), MSE loss, fused backward, gradient all-reduce
(N > 1) and the PyTorch Adam step — the body of nerf/run_nerf_acc.py:263-307 upstream.
Weak scaling: every rank renders its own projection each step.

    python bench.py --gpus N --steps K --warmup W [--precision f16|f32|bf16x3|bf16]

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit, ms_per_step,
`roofline` for the dominant kernel (HIP-event timed inside the library on the launch stream) and
`cpu_baseline` (the CPU oracle timed on this host's cores on a bounded ray sample, rank 0, N=1 only).
The run FAILS (non-zero exit, no JSON) when the rendered pixels at the timed precision are outside the
1e-4 relative-L2 parity bar against the CPU oracle, at the seeded initial weights or at the weights the
timed steps produced.  N > 1 adds `dp_grad_parity` (all-reduced N-rank gradient vs rank 0's 1-rank
gradient of the same global batch), `rccl_ranks` and the all-reduce time.
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

PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "f16": 2500.0, "f16s8": 2500.0}     # MI355X_MICROARCH.md: dense MFMA peaks (f16 = bf16 rate)
DTYPE_NAME = {"f32": "f32", "bf16x3": "bf16x3 (split bf16, f32 accumulate)", "bf16": "bf16 (f32 accumulate)",
              "f16": "f16 (f32 accumulate; first layer split bf16)",
              "f16s8": "f16 (f32 accumulate; first layer split bf16; backward stash kept as bf8)"}
PARITY_BAR = 1e-4
# weight-gradient bar per precision (relative L2 of the whole gradient vs the exact-fp32 kernels; tests/test_gpu_parity.py TOL)
# (f16s8 measures 1.9e-3 on this benchmark's phantom targets with the stochastically rounded dZ' stash - 2.8e-2 with round-to-nearest -
# f16 4.7e-4, bf16 3.1e-3)
GRAD_BAR = {"f16s8": 1e-2, "f16": 1e-2, "bf16x3": 3e-2, "bf16": 6e-2}


def flops_per_sample(width, layers, k0=3):
    """Algorithmic FLOPs per ray-sample: MAC x 2, no padding, no recomputation."""
    fwd = 2 * (k0 * width + layers * width * width + width)
    dgrad = 2 * (layers * width * width + width)
    wgrad = fwd
    return fwd, dgrad, wgrad


def stash_bytes_per_sample(width, layers, precision, in_kernel_small=True):
    """(bytes written by k_chain<bwd>, bytes read back by k_wgrad) per ray-sample: DESIGN.md section 2."""
    esz = 4 if precision == "f32" else 2
    wgrad = 2 * layers * width * esz                     # H_{l-1}, dZ_l (J_l) of the hidden layers
    if precision == "f32" or not in_kernel_small:
        return 2 * (layers + 1) * width * esz, wgrad
    # 16-bit rays mode: H_N / dZ_0 are not stashed; per 32-sample group 3 x width + 8 floats of partial sums instead
    chain = wgrad + (3 * width + 8) * 4 / 32
    if precision == "f16s8":                              # one byte per stash element
        chain, wgrad = chain - wgrad + wgrad // 2, wgrad // 2
    if precision == "f16":
        chain += 4                                        # dL/draw per sample (the chain is normalised by it)
        wgrad += 4
    if precision == "f16s8":
        chain += 4 / 32                                   # one exponent per 32-sample group
        wgrad += 4 / 32
    return chain, wgrad


class EventTimer:
    """Pairs of HIP events recorded on the current stream; read after the timed region."""

    def __init__(self):
        self.pairs = []

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._open = e

    def end(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.pairs.append((self._open, e))

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs)


def live_traffic(args):
    """HBM bytes per launch of the dominant kernel (k_chain<bwd>) measured for THIS command: two child runs of this script
    (1 warm-up + 1 step, no CPU leg) under `rocprofv3 --kernel-trace --pmc <counter>`, FETCH_SIZE and WRITE_SIZE in separate
    passes as MI355X_MICROARCH.md prescribes, before this process touches the GPU.  Counter unit KiB; FETCH_SIZE reports half
    the bytes of wide coalesced streaming reads on gfx950 (doubled here), WRITE_SIZE is exact for 16-B/lane stores.
    Returns (bytes_per_launch | None, note)."""
    import csv, glob, re, shutil, signal, subprocess, tempfile
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 not on PATH"
    me = os.path.abspath(__file__)
    per_launch = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="afx_pmc_", dir="/tmp")
        cmd = [rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, me,
               "--no-cpu", "--no-pmc", "--no-grad-check", "--steps", "1", "--warmup", "1", "--precision", args.precision, "--res", str(args.res),
               "--samples", str(args.samples), "--layers", str(args.layers), "--width", str(args.width),
               "--workspace-gib", str(args.workspace_gib)] + (["--unfused"] if args.unfused else [])
        try:
            pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=300)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)       # exactly the process group this call started
                pr.wait()
                return None, f"{ctr} pass timed out"
            if rc != 0:
                return None, f"{ctr} pass exited with {rc}"
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == ctr and re.search(r"k_chain_bf16<\d+, \w+, \w+, true", row.get("Kernel_Name", "")):
                            tot += float(row["Counter_Value"])
                            n += 1
            if n == 0:
                return None, f"{ctr} pass: no k_chain<bwd> dispatch in the counter output"
            per_launch[ctr] = tot * 1024.0 / n
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"], \
        "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (FETCH_SIZE x2 per MI355X_MICROARCH.md)"


def self_launch(n_gpus):
    """`python bench.py --gpus N` outside torchrun: start `python -m torch.distributed.run ... bench.py <same flags>` as a CHILD
    process (one rank per GPU) and return its exit code.  This process never touches the GPU (a process that has initialised it
    must not exec another program on this pool, and a parent holding the device would count against the card's process limit);
    the child's stdout - rank 0's one JSON line - and stderr are inherited, i.e. relayed as they are."""
    import signal, socket, subprocess
    with socket.socket() as s:          # a free rendezvous port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    pr = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return pr.wait()
    except KeyboardInterrupt:
        os.killpg(pr.pid, signal.SIGTERM)      # exactly the process group this call started
        return pr.wait()


DP_PARITY_BAR = 1e-5      # all-reduced N-rank gradient vs the 1-rank gradient of the same global batch (fp32 summation order only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default=os.environ.get("AFX_BENCH_PRECISION", "f16s8"), choices=["f32", "bf16x3", "bf16", "f16", "f16s8"])
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--cpu-rays", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workspace-gib", type=float, default=128.0,
                    help="backward stash workspace per GPU (an upper limit: the engine takes what two 4 GiB-per-layer-plane chunks need, 79 GB for the 512^2x128 projection)")
    ap.add_argument("--unfused", action="store_true",
                    help="render -> mse_loss -> autograd backward (forward rendered separately) instead of the fused train step")
    ap.add_argument("--no-grad-check", action="store_true", help="skip the full-size gradient check against the fp32 kernels (about 2 s)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the live rocprofv3 --pmc passes for roofline.traffic (use the committed profile instead); "
                         "required when this command itself runs under rocprofv3")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    live = (None, "not measured")
    # (device_count() does not initialise the GPU on this image)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_pmc and args.precision != "f32" and torch.cuda.device_count() > 0:
        live = live_traffic(args)        # child processes; this process has not touched the GPU yet

    from nerf_for_angiography_amd import dist as afx_dist
    from nerf_for_angiography_amd.model.CPPN import CPPN
    from nerf_for_angiography_amd.render import render_projection, render_rays, train_step_mse, projection_spec
    from nerf_for_angiography_amd.phantomdata.helpers import capsule_tree, capsule_mu, get_ray_values

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
    sync = afx_dist.GradSync().install()
    t_allreduce, t_adam = EventTimer(), EventTimer()
    sync.on_call = (t_allreduce.begin, t_allreduce.end)
    opt = torch.optim.Adam(list(model.parameters()), lr=1e-4, fused=torch.cuda.is_available())      # PyTorch multi-tensor kernel (same update; one launch)

    # synthetic P-ANGIO phantom: 31-capsule vessel tree, mu = 0.2, voxelised once on a 192^3 grid over +-100 and
    # projected by the HIP ground-truth projector (afx_project_volume: all projections of this rank in ONE launch, rays
    # generated in the kernel from the poses) - set-up, outside the timed region
    from nerf_for_angiography_amd.phantomdata.helpers import VoxelVolume
    from nerf_for_angiography_amd.engine import project_volume
    caps = capsule_tree(levels=5, seed=0)
    n_proj = args.steps + args.warmup
    ax = np.linspace(-100.0, 100.0, 192)
    with torch.no_grad():
        tx = torch.from_numpy(ax).float().to(device)
        gx, gy, gz = torch.meshgrid(tx, tx, tx, indexing="ij")
        mu = torch.cat([capsule_mu(torch.stack([gx[i0:i0 + 16], gy[i0:i0 + 16], gz[i0:i0 + 16]], -1).reshape(-1, 3), caps)
                        for i0 in range(0, 192, 16)]).reshape(192, 192, 192)
        del gx, gy, gz
    vol = VoxelVolume(ax, ax, ax, mu.cpu().numpy(), fill_value=0.0, device=device)
    poses = []
    for i in range(n_proj):
        theta = 2.0 * (i * world + rank)
        _, _, m44, _, _ = get_ray_values(theta, 0.0, 0.0, np.array([0, 0, 1500.0]), 2, 2, focal, "cpu")
        poses.append(torch.from_numpy(m44[None]).to(device))
    z_gt = torch.linspace(0., 1., 160, device=device) * (far - near) + near
    with torch.no_grad():
        tg = project_volume(vol.values, vol.origin, vol.spacing, vol.fill_value, z_gt, poses=torch.cat(poses), width=W, height=H,
                            focal=focal, type_ct=True)
    targets = [t.contiguous() for t in tg.view(n_proj, W * H)]
    del tg, mu

    fused = not args.unfused and args.precision != "f32"
    n_global = world * W * H          # the loss is the mean over the GLOBAL batch; the all-reduce is a SUM (dist.GradSync)

    def step(i, timed=False):
        opt.zero_grad(set_to_none=True)
        if fused:       # forward + MSE + backward in one pass per ray chunk (afx_train_step_mse)
            loss, _ = train_step_mse(model, projection_spec(poses[i], W, H, focal, S, near, far), targets[i],
                                     n_global=n_global)
        else:
            out = render_projection(model, poses[i], W, H, focal, S, near, far)
            loss = ((out.rgb_map - targets[i]) ** 2).sum() / n_global
            loss.backward()
        if timed:
            t_adam.begin()
        opt.step()
        if timed:
            t_adam.end()
        return loss

    # the backward workspace is allocated before any timing (also with --warmup 0): a 128 GiB hipMalloc is setup, not a step
    from nerf_for_angiography_amd import _lib as afx_lib
    full = int(model.engine.lib.afx_query(model.engine.h, afx_lib.Q_BWD_WORKSPACE_FULL, 0, W * H * ((S + 31) // 32 * 32),
                                          afx_lib.PREC[args.precision]))
    model.engine._workspace(min(full, model.engine.max_workspace_bytes), device)
    for i in range(args.warmup):
        step(i)
    t_allreduce.pairs.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    model.engine.profile(True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        loss = step(args.warmup + i, timed=True)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    hip_ms = ev0.elapsed_time(ev1)
    model.engine.profile(False)
    sync.on_call = None
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
    stash_bytes, wgrad_bytes = stash_bytes_per_sample(args.width, args.layers, args.precision,
                                                      os.environ.get("AFX_SMALL_IN_KERNEL", "1") != "0")
    roofline = {"bound": "mfma", "kernel": "k_chain<bwd>: forward + Beer-Lambert + input-gradient chain + stash of H_l, dZ_l (+ first/output-layer group sums)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": None, "launches": dom_n, "avg_launch_ms": round(avg_s * 1e3, 3),
                "algorithmic_flop_per_sample": alg,
                "algorithmic_stash_bytes_per_sample": round(stash_bytes, 1),
                "hbm_write_GBps_algorithmic": round(stash_bytes * per_launch_samples / avg_s / 1e9, 1),
                "kernel_ms_per_step": {k: round(v[0] / args.steps, 2) for k, v in prof.items()},
                "wgrad_hbm_read_GBps_algorithmic": round(wgrad_bytes * samples_per_step / max(prof["wgrad"][0] / args.steps * 1e-3, 1e-9) / 1e9, 1),
                "adam_ms_per_step": round(t_adam.total_ms() / args.steps, 3),
                "allreduce_ms_per_step": round(t_allreduce.total_ms() / args.steps, 3) if world > 1 else 0.0,
                "step_tflops_algorithmic": round((fwd_f + dgrad_f + wgrad_f) * value / world / 1e12, 2),
                "step_frac_of_peak": round((fwd_f + dgrad_f + wgrad_f) * value / world / 1e12 / peak, 4)}
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS round's code
    # (profiles/r03_pmc_traffic.json names the commit they ran at); only valid for the configuration those passes ran
    traffic_file = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    default_cfg = (W, S, args.layers, args.width, fused, abs(args.workspace_gib - 128.0) < 1e-9) == (512, 128, 8, 256, True, True)
    if live[0] is not None:
        roofline["traffic"] = float(live[0])
        roofline["traffic_source"] = live[1]
        roofline["algorithmic_bytes_per_launch"] = float(stash_bytes * per_launch_samples)
    elif default_cfg and os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file)).get(args.precision)
            if t:
                roofline["traffic"] = float(t["bytes_per_launch"])
                roofline["traffic_source"] = t["source"] + f" [live passes: {live[1]}]"
                roofline["algorithmic_bytes_per_launch"] = float(stash_bytes * per_launch_samples)
        except Exception:
            pass
    # the dominant kernel is power-limited under real data (tools/power_probe.py): its time with all-zero operands, same binary
    probe_file = os.path.join(ROOT, "profiles", "r02_power_probe.json")
    if default_cfg and args.precision == "f16s8" and os.path.exists(probe_file):
        try:
            pz = json.load(open(probe_file))["plain_order (default build)"]
            roofline["power_note"] = {"chain_bwd_ms_per_step_real_data": pz["seeded"]["chain_bwd_ms"],
                                      "chain_bwd_ms_per_step_zero_operands": pz["zero"]["chain_bwd_ms"],
                                      "source": "profiles/r02_power_probe.json (committed measurement, not this run)"}
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
              "ms_per_step_hip_events": round(hip_ms / args.steps, 2),
              "final_loss": round(float(loss) * (world if fused else 1), 6), "roofline": roofline}

    if world > 1:
        if fused:
            result.update(dp_parity(model, rank, world, poses[0], targets[0], W, H, focal, S, near, far, train_step_mse,
                                    projection_spec, afx_dist))
            # a data-parallel line is only reported when the collective really ran over all N ranks and reproduced the 1-rank gradient
            bad = None
            if result["rccl_ranks"] != args.gpus:
                bad = f"process group has {result['rccl_ranks']} ranks, --gpus {args.gpus}"
            elif rank == 0 and not result["dp_grad_parity"] <= DP_PARITY_BAR:
                bad = f"dp_grad_parity {result['dp_grad_parity']:.3e} > {DP_PARITY_BAR:g}"
            flag = torch.tensor([1.0 if bad else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag.item()) > 0:
                if rank == 0:
                    print(json.dumps(result), file=sys.stderr, flush=True)
                dist.destroy_process_group()
                raise SystemExit(f"DATA-PARALLEL FAILURE: {bad or 'see rank 0'}")
    trained_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if rank == 0 and world == 1 and fused and not args.no_grad_check:
        # Full-size gradient check of the timed configuration (same workspace, same chunking): the fused step's weight gradient of
        # projection 0 at the seeded weights against the exact-fp32 kernels (k_chain_f32 / k_wgrad_f32: fp32 MFMA, fp32 stash).
        # A forward-only parity check cannot see a wrong backward (round 2 found 32-bit stash offsets wrapping at this size).
        model.load_state_dict(init_state)
        model.invalidate()
        model.zero_grad(set_to_none=True)
        train_step_mse(model, projection_spec(poses[0], W, H, focal, S, near, far), targets[0], n_global=W * H)
        g_timed = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).double()
        m32 = CPPN(dict(md, precision="f32")).to(device)
        m32.load_state_dict(init_state)
        m32.engine.max_workspace_bytes = 48 << 30
        out32 = render_projection(m32, poses[0], W, H, focal, S, near, far)
        (((out32.rgb_map - targets[0]) ** 2).sum() / (W * H)).backward()
        g_f32 = torch.cat([p.grad.reshape(-1) for p in m32.parameters() if p.grad is not None]).double()
        gerr = float((g_timed - g_f32).norm() / g_f32.norm())
        result["grad_parity_vs_fp32_kernels"] = {"rel_l2": gerr, "bar": GRAD_BAR[args.precision], "n_params": int(g_f32.numel()),
                                                 "what": "fused train step of projection 0 at the seeded weights, full size"}
        del m32, out32
        model.load_state_dict(trained_state)
        model.invalidate()
        if not gerr <= GRAD_BAR[args.precision]:
            print(json.dumps(result), file=sys.stderr, flush=True)
            raise SystemExit(f"GRADIENT FAILURE: {args.precision} weight gradient is {gerr:.3e} relative L2 from the fp32 kernels "
                             f"(bar {GRAD_BAR[args.precision]:g})")
    if rank == 0 and world == 1 and not args.no_cpu:
        result.update(cpu_leg(model, args, poses[0], targets[0], W, H, focal, near, far, S, device, render_rays, init_state,
                              trained_state))
        par = result["parity_vs_cpu_oracle"]
        worst = max(par["rel_l2"], par["rel_l2_trained_weights"])
        if not worst <= PARITY_BAR:
            print(json.dumps(result), file=sys.stderr, flush=True)
            raise SystemExit(f"PARITY FAILURE: {args.precision} pixels are {worst:.3e} relative L2 from the CPU oracle "
                             f"(bar {PARITY_BAR:g}); no benchmark line is reported at a precision that misses the bar")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def usable_cpus():
    """Threads for the CPU leg and how they were chosen: the CPUs this process may run on (affinity mask), clipped by the
    cgroup CPU quota when there is one and by the 16-core share a one-GPU box owns of its host (the pool's sizing rule;
    AFX_CPU_THREADS overrides) - more threads than the share only oversubscribe it."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.999))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, (q + per - 1) // per)
        except Exception:
            pass
    share = int(os.environ.get("AFX_CPU_THREADS", "16"))
    n = min(x for x in (aff, quota, share) if x)
    return n, f"threads = {n} = min(affinity mask {aff}, cgroup quota {quota}, box share {share})"


def dp_parity(model, rank, world, pose, target, W, H, focal, S, near, far, train_step_mse, projection_spec, afx_dist):
    """N ranks take contiguous shards of ONE global batch (the rays of one projection), all-reduce; rank 0 also
    computes that batch alone: relative L2 between the two flat gradients (fp32 summation order is the only difference: the stash's
    stochastic rounding is seeded by the sample's position in space, not by its place in a launch)."""
    n = W * H
    start, count = afx_dist.shard(n, rank, world)
    dist.broadcast(target, 0)
    pose = pose.clone()
    dist.broadcast(pose, 0)

    def flat_grad(ray_id0, n_rays, tgt):
        model.zero_grad(set_to_none=True)
        train_step_mse(model, projection_spec(pose, W, H, focal, S, near, far, ray_id0=ray_id0, n_rays=n_rays), tgt, n_global=n)
        return torch.cat([p.grad.reshape(-1) for p in model._hip_params()]).clone()

    g_sharded = flat_grad(start, count, target[start:start + count].contiguous())      # the hook SUM-all-reduces it
    out = {"rccl_ranks": dist.get_world_size(), "dist_backend": dist.get_backend()}
    hook, afx_dist._render._grad_hook = afx_dist._render._grad_hook, None
    if rank == 0:
        g_single = flat_grad(0, n, target)
        out["dp_grad_parity"] = float((g_sharded - g_single).norm() / g_single.norm())
    afx_dist._render._grad_hook = hook
    dist.barrier()
    return out


def cpu_leg(model, args, pose, target, W, H, focal, near, far, S, device, render_rays, init_state, trained_state):
    """CPU oracle (kind "port") timed on this host on a bounded ray sample + GPU-vs-oracle parity on those rays, at the
    seeded initial weights and at the weights the timed steps produced."""
    from oracle import angio_oracle as orc
    g = torch.Generator().manual_seed(1234)
    pick = torch.randperm(W * H, generator=g)[:args.cpu_rays]
    o_all, d_all = orc.get_rays(pose[0].cpu().numpy(), W, H, focal)
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    tgt = target.cpu()[pick]
    cfg = dict(num_early_layers=args.layers, num_filters=args.width)
    keep = model.precision
    ncpu, cpu_note = usable_cpus()
    torch.set_num_threads(ncpu)

    def gpu_pixels(state):
        model.load_state_dict(state)
        out = {}
        with torch.no_grad():
            for prec in dict.fromkeys([keep, "f16", "bf16x3", "bf16"]):      # (f16s8 renders exactly as f16)
                model.precision = prec
                out[prec] = render_rays(model, o.to(device), d.to(device), S, near, far, mode="acc").rgb_map.cpu()
        model.precision = keep
        return out

    def cpu_pixels(state):
        params = {k: v.detach().cpu().clone() for k, v in state.items()}
        with torch.no_grad():
            return orc.render_rays(o, d, cfg, params, near=near, far=far, n_samples=S, convention="acc")

    rel = lambda a, b: float((a - b).norm() / b.norm())
    pix_init_gpu, pix_init_cpu = gpu_pixels(init_state), cpu_pixels(init_state)
    pix_tr_gpu, pix_tr_cpu = gpu_pixels(trained_state), cpu_pixels(trained_state)
    model.load_state_dict(init_state)

    # timed CPU training steps (forward + backward + Adam) at the initial weights: best of 3 after one warm-up
    params = {k: v.detach().cpu().clone() for k, v in init_state.items()}
    leaves = {k: v.requires_grad_(True) for k, v in params.items() if k.startswith(("early", "output"))}
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-4)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        opt.zero_grad()
        pix = orc.render_rays(o, d, cfg, params, near=near, far=far, n_samples=S, convention="acc")
        torch.nn.functional.mse_loss(pix, tgt).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    mse = float(((pix_init_gpu[keep] - pix_init_cpu).double() ** 2).mean())
    return {"cpu_baseline": {"value": round(args.cpu_rays * S / best, 1), "unit": "ray-samples/s",
                             "cores": torch.get_num_threads(), "kind": "port",
                             "sample": f"{args.cpu_rays} rays x {S} samples of the same projection, {args.layers}x{args.width} MLP, "
                                       "fp32 PyTorch-CPU oracle, fwd+bwd+Adam, best of 3 timed steps after 1 warm-up; " + cpu_note},
            "parity_vs_cpu_oracle": {"rays": args.cpu_rays, "precision": keep, "bar": PARITY_BAR,
                                     "rel_l2": rel(pix_init_gpu[keep], pix_init_cpu),
                                     "psnr_db": round(-10 * np.log10(max(mse, 1e-30)), 2),
                                     "rel_l2_trained_weights": rel(pix_tr_gpu[keep], pix_tr_cpu),
                                     "weights": f"seeded initial weights / the weights after the {args.steps + args.warmup} benchmark steps",
                                     "rel_l2_by_precision": {k: rel(v, pix_init_cpu) for k, v in pix_init_gpu.items()},
                                     "rel_l2_by_precision_trained_weights": {k: rel(v, pix_tr_cpu) for k, v in pix_tr_gpu.items()}}}


if __name__ == "__main__":
    main()
