#!/bin/bash
# A/B timing of library variants (dev tool): every ab/*.so is copied over the in-tree library and benched.
set -e
LIB=nerf_for_angiography_amd/csrc/libafx.so
cp $LIB /tmp/libafx_orig.so
for v in ab/*.so; do
  cp $v $LIB
  n=$(basename $v .so)
  python bench.py --no-cpu "$@" > gpurun_out/ab_$n.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab_$n.log; continue; }
  grep "^\[stamps\]" gpurun_out/ab_$n.log | head -8 || true
  grep "^{" gpurun_out/ab_$n.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('final_loss'))"
done
cp /tmp/libafx_orig.so $LIB
