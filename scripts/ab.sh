#!/bin/bash
# same-box A/B of two builds: AB_A / AB_B = extra hipcc flags, AB_W = workloads, AB_ARGS = extra bench.py arguments
cd $GRAFT_REPO_ROOT
for v in A B A B; do
  fl=$( [ $v = A ] && echo "$AB_A" || echo "$AB_B" )
  MCQ_HIPCC_FLAGS="$fl -DMCQ_AB_$v" python3 -c "
import importlib; b=importlib.import_module('metacache-mpi_amd.build'); b.build_hip()" || exit 1
  for w in ${AB_W:-c2 paired}; do
    timeout -k 10 200 python3 bench.py --steps 48 --no-cpu-baseline --workload $w $AB_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('variant $v [$fl]', '$w', 'ms/step %.4f' % d['ms_per_step'])"
  done
done
