set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/transient.py > gpurun_out/transient.log 2>&1
python tools/long_train.py --iters 20000 --out gpurun_out/r03_long_train.json > gpurun_out/long_train.log 2>&1
tail -3 gpurun_out/transient.log
