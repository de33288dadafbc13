set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -q -m gpu 2>&1 | tail -40 > gpurun_out/t_round3.log
./tools/micro/shape_ab 8192 7 > gpurun_out/shape_ab.log 2>&1
./tools/micro/shape_ab 8192 5 1 > gpurun_out/shape_ab_zero.log 2>&1
python tools/c3_only.py > gpurun_out/c3_plain.log 2>&1
PRE=1024 python tools/c3_only.py > gpurun_out/c3_pre.log 2>&1
WS_FIRST=1 python tools/c3_only.py > gpurun_out/c3_wsfirst.log 2>&1
python tools/march_ab.py > gpurun_out/march_ab.log 2>&1
cat gpurun_out/shape_ab.log
