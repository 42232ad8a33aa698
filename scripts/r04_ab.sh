#!/bin/bash
# round 4, same-box comparisons: the r03 library (scripts/mk_variant.sh r03 cc4c076) against the working tree's, per workload
cd $GRAFT_REPO_ROOT
R03=scripts/_ab/libmcq_hip_r03.so
AB_ROUNDS=2 bash scripts/ab_libs.sh "c2_r03|$R03|--steps 24" "c2_r04|-|--steps 24" "long_r03|$R03|--workload long --steps 16" "long_r04|-|--workload long --steps 16" \
  "paired_r03|$R03|--workload paired --steps 24" "paired_r04|-|--workload paired --steps 24"
AB_ROUNDS=1 AB_TIMEOUT=400 bash scripts/ab_libs.sh "refseq_r03|$R03|--refseq-scale --steps 10 --warmup 3" "refseq_r04|-|--refseq-scale --steps 10 --warmup 3" \
  "refseqp_r03|$R03|--refseq-scale --workload paired --steps 10 --warmup 3" "refseqp_r04|-|--refseq-scale --workload paired --steps 10 --warmup 3"
