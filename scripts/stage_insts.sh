#!/bin/bash
# dynamic instruction counts per stage: PMC on the fused kernel with the stop-stage hook
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in ${STAGES:-1 2 3 4 5 0}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "k_query_wave" --output-format csv -d gpurun_out/si_$s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --stop-stage $s > gpurun_out/si_$s.log 2>&1
  python3 - $s <<'PY'
import csv,glob,sys,collections
s=sys.argv[1]
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/si_%s/**/*counter_collection.csv"%s, recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("stop",s, {k: round(sum(v)/len(v)/1048576,1) for k,v in sorted(acc.items())})
PY
done
