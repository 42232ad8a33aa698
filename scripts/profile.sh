#!/bin/bash
# run on the GPU box from the repo root: bash scripts/profile.sh <tag> [bench args]
# kernel trace + separate PMC passes (never combined with sys/hip traces), outputs under gpurun_out/
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 6 --warmup 1 --no-cpu-baseline --no-pcie-leg $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py $ARGS > gpurun_out/${TAG}_trace.log 2>&1 || exit 1
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --kernel-include-regex "k_query|k_reduce|k_shard" --output-format csv -d gpurun_out/${TAG}_pmc$i -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 gpurun_out/${TAG}_pmc$i.log; }
  echo "pass $i done"
done
python3 scripts/summarize_prof.py $TAG gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc1 gpurun_out/${TAG}_pmc2 gpurun_out/${TAG}_pmc3 gpurun_out/${TAG}_pmc4 gpurun_out/${TAG}_pmc5 > gpurun_out/${TAG}_summary.txt 2>&1
cp profiles/${TAG}_*.csv gpurun_out/ 2>/dev/null
tail -60 gpurun_out/${TAG}_summary.txt
