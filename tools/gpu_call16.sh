cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -v -m gpu -x -k "coarse_reuse or graph_capture_of_the_split" > gpurun_out/t_hier.log 2>&1
grep -n "PASSED\|FAILED\|Fatal\|fault\|Error\|error\|::test" gpurun_out/t_hier.log | head -20
