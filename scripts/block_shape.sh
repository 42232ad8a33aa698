#!/bin/bash
# shape of the workgroup kernel for long reads: threads per query, keys of LDS, resident workgroups (same-box comparison)
cd $GRAFT_REPO_ROOT
# MCQ_SHAPES="NT:LCAP:WGS ..." overrides the list
for v in ${MCQ_SHAPES:-1024:8192:512 64:4096:1024 64:3072:1536 128:4096:1024 256:4096:1024 1024:8192:512}; do
  set -- ${v//:/ }
  MCQ_HIPCC_FLAGS="-DMCQ_BLOCK_NT=$1 -DMCQ_BLOCK_LCAP=$2 -DMCQ_BLOCK_WGS_DEFAULT=$3" python3 -c "
import importlib; b=importlib.import_module('metacache-mpi_amd.build'); b.build_hip()" > gpurun_out/bs_build.log 2>&1 || { tail -5 gpurun_out/bs_build.log; exit 1; }
  timeout -k 10 300 python3 bench.py --workload long --steps 12 --warmup 2 --cpu-seconds 2 > gpurun_out/bs.json 2> gpurun_out/bs.err || { tail -3 gpurun_out/bs.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bs.json')); print('NT $1 LCAP $2 WGS $3: ms/step %.3f  reads/s %.3g  kernel_ms' % (d['ms_per_step'], d['value']), {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()}, 'parity', d['cpu_baseline']['gpu_matches_cpu_on_first_batch'])"
done
