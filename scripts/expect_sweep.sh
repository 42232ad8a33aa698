#!/bin/bash
# sketch threshold target (-DMCQ_SKETCH_EXPECT) against the default 40, same box: c2 and paired
cd $GRAFT_REPO_ROOT
for e in 40 24 28 32 36 48 40; do
  MCQ_HIPCC_FLAGS="-DMCQ_SKETCH_EXPECT=$e -DMCQ_SWEEP$e" python3 -c "
import importlib; b=importlib.import_module('metacache-mpi_amd.build'); b.build_hip()" > gpurun_out/es_build.log 2>&1 || { echo "build failed: $e"; tail -3 gpurun_out/es_build.log; continue; }
  for w in c2 paired; do
    timeout -k 10 200 python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-pcie-leg --workload $w > gpurun_out/es.json 2> gpurun_out/es.err || { echo "bench failed"; continue; }
    python3 -c "
import json; d=json.load(open('gpurun_out/es.json')); print('[expect $e]', '$w', 'ms/step %.4f' % d['ms_per_step'], 'valu/read %.1f' % d['roofline'].get('valu_issue',{}).get('insts_per_read',0))"
  done
done
