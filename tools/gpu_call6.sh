set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -q -m gpu -k "split or hierarchical_train" 2>&1 | tail -30 > gpurun_out/t_split.log
for i in 1 2; do
python tools/ref_iter.py 4 128 500 >> gpurun_out/ref_ab.log 2>&1
AFX_NO_SPLIT=1 python tools/ref_iter.py 4 128 500 >> gpurun_out/ref_ab.log 2>&1
python tools/ref_iter.py 8 256 200 >> gpurun_out/ref_ab.log 2>&1
AFX_NO_SPLIT=1 python tools/ref_iter.py 8 256 200 >> gpurun_out/ref_ab.log 2>&1
done
AFX_FORCE_SPLIT=1 python bench.py --steps 5 --warmup 2 --no-cpu --no-pmc > gpurun_out/bench_forcesplit.json 2> gpurun_out/bench_forcesplit.err
python bench.py --steps 5 --warmup 2 --no-cpu --no-pmc > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err
(time python -m pytest tests -q -m gpu 2>&1 | tail -15) > gpurun_out/t_all.log 2>&1
tail -5 gpurun_out/t_split.log; grep -v Warn gpurun_out/ref_ab.log | grep ms; tail -6 gpurun_out/t_all.log
