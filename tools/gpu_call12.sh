set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/grid_iter3.log
python tools/grid_iter.py 4 128 300 fused vessels > $O 2>&1
python tools/grid_iter.py 4 128 300 ops vessels >> $O 2>&1
python tools/grid_iter.py 4 128 300 fused all >> $O 2>&1
python tools/grid_iter.py 4 128 300 ops all >> $O 2>&1
python tools/grid_iter.py 8 256 200 fused all >> $O 2>&1
python tools/grid_iter.py 8 256 200 ops all >> $O 2>&1
python tools/grid_iter.py 8 256 200 fused vessels >> $O 2>&1
grep -v Warn $O | grep -v amdgpu
(time python -m pytest tests -q -m gpu 2>&1 | tail -8) > gpurun_out/t_all.log 2>&1
tail -8 gpurun_out/t_all.log
