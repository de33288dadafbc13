set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "packed or tanh or grid_march" 2>&1 | tail -40 > gpurun_out/t_packed.log && \
timeout -k 10 300 python tools/march_ab.py > gpurun_out/march_ab3.log 2>&1
tail -15 gpurun_out/t_packed.log; tail -2 gpurun_out/march_ab3.log | cut -c1-1500
