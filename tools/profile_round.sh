#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel stats, HBM-traffic PMC passes, SQ/MFMA PMC passes,
# bench JSONs.  Outputs land under gpurun_out/ and are reduced by tools/pmc_reduce.py / tools/make_profiles.py.
# usage: tools/profile_round.sh [precision=f16s8] [tag=r03]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=${1:-f16s8}
T=${2:-r03}
O=gpurun_out
set -x
python3 bench.py --precision $P > $O/${T}_bench_$P.log 2>&1 && grep "^{" $O/${T}_bench_$P.log | tail -1 > $O/${T}_bench_$P.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu --no-pmc --no-grad-check --precision $P > $O/prof_stats.log 2>&1
python3 tools/pmc_reduce.py $O/prof_stats $O/${T}_${P}_stats.json > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --no-cpu --no-pmc --no-grad-check --steps 1 --warmup 1 --precision $P > $O/prof_fetch.log 2>&1
python3 tools/pmc_reduce.py $O/prof_fetch $O/${T}_${P}_fetch.json > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --no-cpu --no-pmc --no-grad-check --steps 1 --warmup 1 --precision $P > $O/prof_write.log 2>&1
python3 tools/pmc_reduce.py $O/prof_write $O/${T}_${P}_write.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq -- python3 bench.py --no-cpu --no-pmc --no-grad-check --steps 1 --warmup 1 --precision $P > $O/prof_sq.log 2>&1
python3 tools/pmc_reduce.py $O/prof_sq $O/${T}_${P}_sq.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/prof_sq2 -- python3 bench.py --no-cpu --no-pmc --no-grad-check --steps 1 --warmup 1 --precision $P > $O/prof_sq2.log 2>&1
python3 tools/pmc_reduce.py $O/prof_sq2 $O/${T}_${P}_sq2.json > /dev/null
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq $O/prof_sq2
ls -la $O | tail -12
