"""cProfile of the training driver's loop (run on the GPU box): python tools/prof_driver.py dense|grid.  What it was written for: the driver ran
at 646 it/s where tools/ref_iter.py claimed 890 - the tool's model had zero gradients."""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
from nerf_for_angiography_amd.nerf.run_nerf_acc import main
argv = ["--synthetic", "--img_size", "100", "--number_angles", "9", "--limited_size", "180", "--n_iters", "1500", "--display_every", "3000",
        "--sample_size", "75", "--depth_samples", "300", "--num_layers", "4", "--num_hidden_units", "128", "--sampling_strategy", "segmentation",
        "--march", sys.argv[1], "--log_dir", "/tmp/run_prof"]
pr = cProfile.Profile(); pr.enable(); main(argv); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
