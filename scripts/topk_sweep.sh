#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" "-DMCQ_TOPK_DPP"; do
  MCQ_HIPCC_FLAGS="$v -DMCQ_DUMMY=1" python3 -c "
import importlib; b=importlib.import_module('metacache-mpi_amd.build'); b.build_hip()" || exit 1
  for cfg in "2 2" "4 4" "8 4" "8 8" "1 2"; do
    set -- $cfg
    timeout -k 10 200 python3 bench.py --steps 16 --no-cpu-baseline --emulate-ranks $1 --max-cand $2 > gpurun_out/tk.json 2> gpurun_out/tk.err || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/tk.json')); print('variant', '$v' or 'lds', 'P', $1, 'M', $2, 'kernel ms %.3f' % d['roofline']['kernel_ms_sum'])"
  done
done
