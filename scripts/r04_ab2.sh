#!/bin/bash
# round 4: long reads bisected on one box (r03 library, working tree, working tree without the two-barrier scan / without the
# sketch prefetch / without both), then the RefSeq-scale table with the direct mode
cd $GRAFT_REPO_ROOT
A=scripts/_ab
AB_ROUNDS=2 bash scripts/ab_libs.sh "long_r03|$A/libmcq_hip_r03.so|--workload long --steps 16" "long_r04|-|--workload long --steps 16" \
  "long_nosp|$A/libmcq_hip_nosp.so|--workload long --steps 16" "long_noscan|$A/libmcq_hip_noscan.so|--workload long --steps 16" "long_nopf|$A/libmcq_hip_nopf.so|--workload long --steps 16"
AB_ROUNDS=1 AB_TIMEOUT=400 bash scripts/ab_libs.sh "refseq_r03|$A/libmcq_hip_r03.so|--refseq-scale --steps 10 --warmup 3" "refseq_r04|-|--refseq-scale --steps 10 --warmup 3" \
  "refseqp_r04|-|--refseq-scale --workload paired --steps 10 --warmup 3"
