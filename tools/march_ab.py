#!/usr/bin/env python3
"""Dense fused iteration vs the reference's own grid-marching iteration (run_nerf_acc.py:284-306) on the synthetic vessel phantom:
5 625 rays x 300 steps, 4x128 MLP, the reference's defaults.  Prints iterations/s and marched samples per iteration for both."""
import os, sys, json, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_for_angiography_amd.nerf.run_nerf_acc import main
out = {}
for march in ("dense", "grid", "grid_ops"):
    d = tempfile.mkdtemp(prefix="afx_march_")
    r = main(["--synthetic", "--img_size", "100", "--number_angles", "3", "--limited_size", "90", "--n_iters", "1500",
              "--display_every", "500", "--sample_size", "75", "--depth_samples", "300", "--num_layers", "4",
              "--num_hidden_units", "128", "--march", march, "--log_dir", d])
    h = r["history"]
    out[march] = [{k: rec.get(k) for k in ("iter", "train_loss", "test_psnr", "it_per_s", "marched_samples_per_iter", "sec")} for rec in h]
    print(march, json.dumps(out[march][-1]), flush=True)
print(json.dumps(out))
