#!/bin/bash
# instruction-cache and issue-stall counters of the fused kernel (values per read, summed over SEs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0; do
 for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_IFETCH SQ_IFETCH_LEVEL"; do
  rm -rf gpurun_out/ic_tmp
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_query_wave" --output-format csv -d gpurun_out/ic_tmp -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --query-flags $v > gpurun_out/ic.log 2>&1
  python3 - $v <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/ic_tmp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("flags", sys.argv[1], {k: round(sum(v)/len(v)/1048576,2) for k,v in sorted(acc.items())}, flush=True)
PY
 done
done
