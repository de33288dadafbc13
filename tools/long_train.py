import os, sys, json, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nerf_for_angiography_amd.nerf.run_nerf_acc import main
d = tempfile.mkdtemp(prefix="afx_long_")
extra = sys.argv[1:]
r = main(["--synthetic", "--img_size", "100", "--number_angles", "3", "--limited_size", "90", "--n_iters", "20000",
          "--display_every", "2500", "--sample_size", "75", "--depth_samples", "300", "--num_layers", "4",
          "--num_hidden_units", "128", "--log_dir", d] + extra)
for rec in r["history"]:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items() if k in ("iter", "train_loss", "test_psnr", "test_vessel_psnr", "it_per_s")})
