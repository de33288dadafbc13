set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/t_round3.log; echo "rc=$?" >> gpurun_out/t_round3.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "grid or driver or march or occupancy or density or acc or unfused" 2>&1 | tail -30 > gpurun_out/t_subset.log
python bench.py --steps 5 --warmup 2 > gpurun_out/bench0.json 2> gpurun_out/bench0.err
AFX_DIST_BACKEND=gloo AFX_DEVICE_INDEX=0 python bench.py --gpus 2 --steps 2 --warmup 1 --workspace-gib 48 > gpurun_out/bench_dp2.json 2> gpurun_out/bench_dp2.err; echo "dp2 rc=$?" >> gpurun_out/bench_dp2.err
tail -5 gpurun_out/t_round3.log gpurun_out/t_subset.log; cat gpurun_out/bench0.json | head -c 600
