set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -q -m gpu -k "split or hierarchical_train" 2>&1 | tail -60 > gpurun_out/t_split.log
FUSED=1 python tools/c3_only.py > gpurun_out/c3_fused.log 2>&1
python tools/ref_iter.py 4 128 > gpurun_out/ref_iter.log 2>&1
python tools/ref_iter.py 8 256 >> gpurun_out/ref_iter.log 2>&1
python tools/march_ab.py > gpurun_out/march_ab2.log 2>&1
python tools/long_train.py --iters 20000 --precisions f32 --out gpurun_out/r03_long_train_f32_s0.json > gpurun_out/long_train_f32_s0.log 2>&1
python tools/long_train.py --iters 20000 --precisions f32 --seed 1 --out gpurun_out/r03_long_train_f32_s1.json > gpurun_out/long_train_f32_s1.log 2>&1
tail -12 gpurun_out/t_split.log
