#!/bin/bash
# cumulative stage times of the fused kernel (stop-stage hook), plain runs without a profiler; ST_ARGS = extra bench.py arguments
cd $GRAFT_REPO_ROOT
# needs a profiling build: MCQ_HIPCC_FLAGS=-DMCQ_PROFILE_HOOKS (the stop-stage bits are rejected otherwise)
export MCQ_HIPCC_FLAGS="$MCQ_HIPCC_FLAGS -DMCQ_PROFILE_HOOKS"
for s in 1 2 3 4 5 0; do
  python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --stop-stage $s $ST_ARGS > gpurun_out/st_$s.json 2> gpurun_out/st_$s.err || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/st_$s.json')); print('stop', $s, 'kernel ms %.3f' % d['roofline']['kernel_ms_sum'])"
done
