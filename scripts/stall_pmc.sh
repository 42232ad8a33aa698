#!/bin/bash
# queue-full / conflict / issue counters of the fused kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VSKIPPED"; do
  rm -rf gpurun_out/stall_tmp
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_query_wave" --output-format csv -d gpurun_out/stall_tmp -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/stall.log 2>&1
  python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/stall_tmp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)/1048576,1) for k,v in sorted(acc.items())}, flush=True)
PY
done
