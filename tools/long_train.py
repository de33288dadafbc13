#!/usr/bin/env python3
"""Long-run convergence at the shipped precision: the SAME 8x256 CPPN, ray batches and Adam schedule trained
at f32 (exact-fp32 kernels, render + autograd), f16 and f16s8 (fused train step) on the C2 geometry - 256x256 projections,
30 training views (theta = 0..174 step 6) + one held-out view, 64 samples/ray, targets from the HIP ground-truth projector
of the voxelised capsule-tree phantom - from one seed to plateau.  Reported: held-out PSNR vs the GT projection along the
run, the final PSNRs, and the relative L2 between the three reconstructed density grids (101^3, split-bf16 evaluation).

    python tools/long_train.py [--iters 20000] [--rays 8192] [--out profiles/r03_long_train.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--eval-every", type=int, default=2500)
    ap.add_argument("--lr", type=float, default=1e-4, help="the reference's learning rate (nerf/run_nerf_acc.py:147)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precisions", default="f32,f16,f16s8")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from nerf_for_angiography_amd.model.CPPN import CPPN
    from nerf_for_angiography_amd.render import render_projection, train_step_mse, projection_spec, density_grid
    from nerf_for_angiography_amd.phantomdata.helpers import capsule_tree, capsule_mu, get_ray_values, VoxelVolume
    from nerf_for_angiography_amd.engine import project_volume
    dev = torch.device("cuda:0")
    W = H = 256
    S, near, far, focal = 64, 1400.0, 1600.0, 13.0 * 256
    thetas = [6.0 * i for i in range(30)] + [93.0]          # the last one is held out (phi = 5)
    poses = []
    for i, th in enumerate(thetas):
        _, _, m44, _, _ = get_ray_values(th, 5.0 if i == 30 else 0.0, 0.0, np.array([0, 0, 1500.0]), 2, 2, focal, "cpu")
        poses.append(torch.from_numpy(m44[None]))
    poses = torch.cat(poses).to(dev)
    caps = capsule_tree(levels=5, seed=0)
    ax = np.linspace(-100.0, 100.0, 192)
    with torch.no_grad():
        tx = torch.from_numpy(ax).float().to(dev)
        gx, gy, gz = torch.meshgrid(tx, tx, tx, indexing="ij")
        mu = torch.cat([capsule_mu(torch.stack([gx[i0:i0 + 16], gy[i0:i0 + 16], gz[i0:i0 + 16]], -1).reshape(-1, 3), caps)
                        for i0 in range(0, 192, 16)]).reshape(192, 192, 192)
        vol = VoxelVolume(ax, ax, ax, mu.cpu().numpy(), fill_value=0.0, device=dev)
        z_gt = torch.linspace(0., 1., 160, device=dev) * (far - near) + near
        tg = project_volume(vol.values, vol.origin, vol.spacing, vol.fill_value, z_gt, poses=poses, width=W, height=H, focal=focal,
                            type_ct=True).view(31, W * H)
    train_t, test_t = tg[:30].reshape(-1).contiguous(), tg[30].contiguous()
    n_train = 30 * W * H
    md = dict(num_early_layers=8, num_late_layers=0, num_filters=256, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5,
              num_img=1, device=dev)
    out = {"config": {"geometry": "C2: 256x256, 30 training views + 1 held-out, 64 samples/ray, acc convention", "model": "8x256 ReLU CPPN",
                      "rays_per_iteration": args.rays, "iterations": args.iters, "lr": f"{args.lr} x 0.1^(it/iters), Adam",
                      "targets": "afx_project_volume of the 192^3 voxelised capsule tree, 160 samples", "seed": args.seed}, "runs": {}}
    grids = {}
    for prec in args.precisions.split(","):
        torch.manual_seed(args.seed)
        m = CPPN(dict(md, precision=prec)).to(dev)
        with torch.no_grad():
            m.output_linear[0].bias.fill_(-5.0)
        opt = torch.optim.Adam(list(m.parameters()), lr=args.lr, fused=True)
        gen = torch.Generator().manual_seed(1234 + args.seed)
        hist = []
        t0 = time.time()
        for it in range(args.iters + 1):
            if it % args.eval_every == 0:
                keep, m.precision = m.precision, "bf16x3"
                with torch.no_grad():
                    pred = render_projection(m, poses, W, H, focal, S, near, far, ray_id0=30 * W * H, n_rays=W * H).rgb_map
                m.precision = keep
                mse = float(torch.nn.functional.mse_loss(pred, test_t))
                hist.append({"iter": it, "test_psnr": round(-10 * np.log10(max(mse, 1e-30)), 3), "sec": round(time.time() - t0, 1)})
                print(prec, hist[-1], flush=True)
                if not np.isfinite(mse):
                    break
            if it == args.iters:
                break
            ids = torch.randint(n_train, (args.rays,), generator=gen).to(dev, torch.int32)
            tgt = train_t[ids.long()]
            opt.zero_grad(set_to_none=True)
            if prec == "f32":
                pix = render_projection(m, poses, W, H, focal, S, near, far, ray_ids=ids).rgb_map
                torch.nn.functional.mse_loss(pix, tgt).backward()
            else:
                train_step_mse(m, projection_spec(poses, W, H, focal, S, near, far, ray_ids=ids), tgt)
            opt.step()
            for gparam in opt.param_groups:
                gparam["lr"] = args.lr * 0.1 ** ((it + 1) / args.iters)
        grids[prec] = density_grid(m, 100.0, 100).double()
        out["runs"][prec] = {"history": hist, "final_test_psnr": hist[-1]["test_psnr"], "train_seconds": round(time.time() - t0, 1)}
        del m, opt
        torch.cuda.empty_cache()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    if "f32" in grids:
        out["density_grid_rel_l2_vs_f32"] = {k: rel(v, grids["f32"]) for k, v in grids.items() if k != "f32"}
        out["psnr_minus_f32_db"] = {k: round(out["runs"][k]["final_test_psnr"] - out["runs"]["f32"]["final_test_psnr"], 3)
                                    for k in grids if k != "f32"}
    print(json.dumps(out), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
