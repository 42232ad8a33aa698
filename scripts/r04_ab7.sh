#!/bin/bash
# round 4: threshold of the workgroup kernels' first pass (long queries first) on ONT-like reads; RefSeq-scale pairs (no long queries: no first pass)
cd $GRAFT_REPO_ROOT
A=scripts/_ab
AB_ROUNDS=2 bash scripts/ab_libs.sh "long_first12k|$A/libmcq_hip_lf12k.so|--workload long --steps 16" "long_first16k|-|--workload long --steps 16" "long_first32k|$A/libmcq_hip_lf32k.so|--workload long --steps 16"
AB_ROUNDS=1 AB_TIMEOUT=400 bash scripts/ab_libs.sh "refseqp|-|--refseq-scale --workload paired --steps 10 --warmup 3" "refseq|-|--refseq-scale --steps 10 --warmup 3"
