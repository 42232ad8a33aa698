#!/bin/bash
# diagnostic: phase clocks of the second wave stage (variant built with -DMCQ_PHASE_CLOCK); WCLK_ARGS = bench.py arguments (default: the RefSeq-scale table)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export MCQ_HIP_LIB=$GRAFT_REPO_ROOT/scripts/_ab/libmcq_hip_phclk.so
timeout -k 10 400 python3 bench.py ${WCLK_ARGS:---refseq-scale} --no-cpu-baseline --no-pcie-leg --steps 6 > gpurun_out/wclk.json 2> gpurun_out/wclk.err || { tail gpurun_out/wclk.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/wclk.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel_ms'])
p=d.get('DIAGNOSTIC_phase_clocks_of_the_workgroup_kernel')
print(p)
w=[p[3+i] for i in range(7)]+[p[11]]; t=sum(w) or 1
print('second wave stage:', ' '.join('%s %.1f%%'%(n,100*x/t) for n,x in zip(['front','gather','cells','compaction','distinct-table','sort+sweep','heads+lists','rest'],w)))
PY
