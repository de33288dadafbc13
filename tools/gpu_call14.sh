cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AFX_VARIANT=stamp128 python tools/ref_iter.py 4 128 100 > gpurun_out/stamp128.log 2>&1
grep -E "stamps|ms/iter" gpurun_out/stamp128.log
