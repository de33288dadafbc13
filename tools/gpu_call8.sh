set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -q -m gpu -k "tanh or packed or grid_march" 2>&1 | tail -40 > gpurun_out/t_packed.log
python tools/march_ab.py > gpurun_out/march_ab3.log 2>&1
python tools/measure_configs.py > gpurun_out/measure_configs3.log 2>&1
tail -15 gpurun_out/t_packed.log; grep -v Warn gpurun_out/measure_configs3.log | tail -12
