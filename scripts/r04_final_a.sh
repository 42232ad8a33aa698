#!/bin/bash
# end-of-round measurements, part A (one gpurun call): kernel traces + PMC passes of the C2-table workloads and of the RefSeq-scale table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c2 paired long; do
  bash scripts/profile.sh r04_$w --workload $w > gpurun_out/r04_${w}_profile.log 2>&1 || echo "profile $w failed"
  cp profiles/r04_${w}_*.csv gpurun_out/ 2>/dev/null
  echo "profiled $w"
done
bash scripts/profile.sh r04_refseq --refseq-scale > gpurun_out/r04_refseq_profile.log 2>&1 || echo "profile refseq failed"
cp profiles/r04_refseq_*.csv gpurun_out/ 2>/dev/null
echo "profiled refseq"
rm -rf gpurun_out/r04_*_trace gpurun_out/r04_*_pmc[0-9]
