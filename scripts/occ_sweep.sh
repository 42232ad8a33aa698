#!/bin/bash
# rebuilds the engine with different waves-per-SIMD targets for the fused kernel and benches each
cd $GRAFT_REPO_ROOT
for occ in ${OCCS:-8 7 6}; do
  MCQ_HIPCC_FLAGS="-DMCQ_WAVE_OCC=$occ" python3 -c "
import importlib; b=importlib.import_module('metacache-mpi_amd.build'); b.build_hip()" || exit 1
  for fl in ${FLAGS:-0}; do
    timeout -k 10 200 python3 bench.py --steps 24 --no-cpu-baseline --query-flags $fl > gpurun_out/occ_$occ.json 2> gpurun_out/occ_$occ.err || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/occ_$occ.json')); print('occ', $occ, 'flags', '$fl', 'ms/step %.3f' % d['ms_per_step'], 'kernel ms %.3f' % d['roofline']['kernel_ms_sum'])"
  done
done
