#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel stats, HBM-traffic PMC passes, SQ/MFMA PMC pass,
# bench JSONs.  Outputs land under gpurun_out/ and are reduced by tools/pmc_reduce.py / tools/make_profiles.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
set -x
python3 bench.py > $O/bench_default.log 2>&1 && grep "^{" $O/bench_default.log | tail -1 > $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu > $O/prof_stats.log 2>&1
python3 tools/pmc_reduce.py $O/prof_stats $O/r01_stats.json > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --no-cpu --steps 1 --warmup 1 > $O/prof_fetch.log 2>&1
python3 tools/pmc_reduce.py $O/prof_fetch $O/r01_fetch.json > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --no-cpu --steps 1 --warmup 1 > $O/prof_write.log 2>&1
python3 tools/pmc_reduce.py $O/prof_write $O/r01_write.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/prof_sq -- python3 bench.py --no-cpu --steps 1 --warmup 1 > $O/prof_sq.log 2>&1
python3 tools/pmc_reduce.py $O/prof_sq $O/r01_sq.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/prof_sq2 -- python3 bench.py --no-cpu --steps 1 --warmup 1 > $O/prof_sq2.log 2>&1
python3 tools/pmc_reduce.py $O/prof_sq2 $O/r01_sq2.json > /dev/null
python3 bench.py --precision f32 --steps 1 --warmup 1 > $O/bench_f32.log 2>&1 && grep "^{" $O/bench_f32.log | tail -1 > $O/bench_f32.json
python3 bench.py --precision bf16x3 --unfused > $O/bench_x3u.log 2>&1 && grep "^{" $O/bench_x3u.log | tail -1 > $O/bench_x3_unfused.json
python3 tools/measure_configs.py > $O/configs.log 2>&1
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq $O/prof_sq2
ls -la $O | tail -12
