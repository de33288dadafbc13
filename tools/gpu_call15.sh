set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "coarse_reuse or hierarchical or graph_capture" 2>&1 | tail -30 > gpurun_out/t_hier.log && \
FUSED=1 timeout -k 10 300 python tools/c3_only.py > gpurun_out/c3_reuse.log 2>&1
tail -12 gpurun_out/t_hier.log; grep -v Warn gpurun_out/c3_reuse.log | tail -4
