set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/grid_iter.log
python tools/grid_iter.py 4 128 300 fused vessels > $O 2>&1
python tools/grid_iter.py 4 128 300 ops vessels >> $O 2>&1
python tools/grid_iter.py 4 128 300 fused all >> $O 2>&1
python tools/grid_iter.py 4 128 300 ops all >> $O 2>&1
python tools/grid_iter.py 8 256 200 fused vessels >> $O 2>&1
python tools/grid_iter.py 8 256 200 ops vessels >> $O 2>&1
python tools/ref_iter.py 4 128 300 >> $O 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_grid -- python3 tools/grid_iter.py 4 128 300 fused all > gpurun_out/prof_grid.log 2>&1
python3 tools/pmc_reduce.py gpurun_out/prof_grid gpurun_out/grid_iter_stats.json > /dev/null
rm -rf gpurun_out/prof_grid
grep -v Warn $O | grep -v amdgpu
