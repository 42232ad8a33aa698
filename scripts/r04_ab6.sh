#!/bin/bash
# round 4: the workgroup kernels' queue in two passes (long queries first) against one pass, same sources
cd $GRAFT_REPO_ROOT
A=scripts/_ab
AB_ROUNDS=2 bash scripts/ab_libs.sh "long_onepass|$A/libmcq_hip_nolf.so|--workload long --steps 16" "long_first16k|-|--workload long --steps 16" "long_first8k|$A/libmcq_hip_lf8k.so|--workload long --steps 16"
AB_ROUNDS=1 AB_TIMEOUT=400 bash scripts/ab_libs.sh "refseqp_onepass|$A/libmcq_hip_nolf.so|--refseq-scale --workload paired --steps 10 --warmup 3" "refseqp_twopass|-|--refseq-scale --workload paired --steps 10 --warmup 3"
