cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 10 600 python -m pytest tests -q -m gpu 2>&1 | tail -12) > gpurun_out/t_all.log 2>&1
tail -14 gpurun_out/t_all.log
