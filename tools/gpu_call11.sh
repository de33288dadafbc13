set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -q -m gpu -x -k "grid or march or packed or occupancy or driver" 2>&1 | tail -15 > gpurun_out/t_grid.log && {
O=gpurun_out/grid_iter2.log
python tools/grid_iter.py 4 128 300 fused vessels > $O 2>&1
python tools/grid_iter.py 4 128 300 ops vessels >> $O 2>&1
python tools/grid_iter.py 4 128 300 fused all >> $O 2>&1
python tools/grid_iter.py 4 128 300 ops all >> $O 2>&1
python tools/grid_iter.py 8 256 200 fused all >> $O 2>&1
python tools/ref_iter.py 4 128 300 >> $O 2>&1
python tools/ref_iter.py 8 256 200 >> $O 2>&1
grep -v Warn $O | grep -v amdgpu; }
tail -6 gpurun_out/t_grid.log
