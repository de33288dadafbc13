cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu --no-pmc --no-grad-check --steps 3 --warmup 1 > $O/prof_stats.log 2>&1
python3 tools/pmc_reduce.py $O/prof_stats $O/gap_stats.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq -- python3 bench.py --no-cpu --no-pmc --no-grad-check --steps 1 --warmup 1 > $O/prof_sq.log 2>&1
python3 tools/pmc_reduce.py $O/prof_sq $O/gap_sq.json > /dev/null
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/prof_sq2 -- python3 bench.py --no-cpu --no-pmc --no-grad-check --steps 1 --warmup 1 > $O/prof_sq2.log 2>&1
python3 tools/pmc_reduce.py $O/prof_sq2 $O/gap_sq2.json > /dev/null
rm -rf $O/prof_stats $O/prof_sq $O/prof_sq2
