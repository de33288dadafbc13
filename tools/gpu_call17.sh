cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/dbg_graph.log
for cfg in "64 128 4 128" "64 300 4 128" "5625 300 4 128"; do
  echo "== $cfg" >> gpurun_out/dbg_graph.log
  timeout -k 10 120 python tools/dbg_graph.py $cfg >> gpurun_out/dbg_graph.log 2>&1 || { echo "FAILED rc=$?" >> gpurun_out/dbg_graph.log; break; }
done
grep -v Warn gpurun_out/dbg_graph.log | grep -v "^  File" | head -40
grep -q FAILED gpurun_out/dbg_graph.log || { timeout -k 10 400 python -m pytest tests/test_gpu_round3.py -v -m gpu -x -k "coarse_reuse or graph_capture_of_the_split" > gpurun_out/t_hier.log 2>&1; grep -n "PASSED\|FAILED\|Fatal\|fault\|rror" gpurun_out/t_hier.log | head -20; }
