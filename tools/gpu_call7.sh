set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -q -m gpu -k "tanh" 2>&1 | tail -40 > gpurun_out/t_act.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "golden or forward or bit_identical" 2>&1 | tail -8 > gpurun_out/t_fwd.log
python tools/measure_configs.py > gpurun_out/measure_configs2.log 2>&1
tail -15 gpurun_out/t_act.log; tail -3 gpurun_out/t_fwd.log
