#!/bin/bash
# diagnostic: phase clocks of the second wave stage on the RefSeq-scale table (variant built with -DMCQ_PHASE_CLOCK)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export MCQ_HIP_LIB=$GRAFT_REPO_ROOT/scripts/_ab/libmcq_hip_phclk.so
timeout -k 10 400 python3 bench.py --refseq-scale --no-cpu-baseline --no-pcie-leg --steps 6 > gpurun_out/wclk_refseq.json 2> gpurun_out/wclk_refseq.err || { tail gpurun_out/wclk_refseq.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/wclk_refseq.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel_ms'])
print(d.get('DIAGNOSTIC_phase_clocks_of_the_workgroup_kernel'))
PY
