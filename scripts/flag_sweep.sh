#!/bin/bash
# compiler scheduling flags against the default build, same box: c2 and paired
cd $GRAFT_REPO_ROOT
for fl in "" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" "-mllvm -enable-post-misched=0" "-mllvm -amdgpu-use-divergent-register-indexing=1" ""; do
  MCQ_HIPCC_FLAGS="$fl -DMCQ_SWEEP" python3 -c "
import importlib; b=importlib.import_module('metacache-mpi_amd.build'); b.build_hip()" > gpurun_out/fs_build.log 2>&1 || { echo "build failed: $fl"; tail -3 gpurun_out/fs_build.log; continue; }
  for w in c2 paired; do
    timeout -k 10 200 python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-pcie-leg --workload $w > gpurun_out/fs.json 2> gpurun_out/fs.err || { echo "bench failed"; continue; }
    python3 -c "
import json; d=json.load(open('gpurun_out/fs.json')); print('[$fl]', '$w', 'ms/step %.4f' % d['ms_per_step'])"
  done
done
