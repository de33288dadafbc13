set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -q -m gpu -k "split" -x 2>&1 | tail -40 > gpurun_out/t_split.log
python tools/long_train.py --iters 20000 --precisions f16,f16s8 --out gpurun_out/r03_long_train_s0.json > gpurun_out/long_train_s0.log 2>&1
python tools/long_train.py --iters 20000 --precisions f16,f16s8 --seed 1 --out gpurun_out/r03_long_train_s1.json > gpurun_out/long_train_s1.log 2>&1
python tools/measure_configs.py > gpurun_out/measure_configs.log 2>&1
tail -12 gpurun_out/t_split.log
