#!/bin/bash
# round 4: chunk size of the workgroup kernels' cursor (long reads), and the wave stages' entries from a cursor (RefSeq scale)
cd $GRAFT_REPO_ROOT
A=scripts/_ab
AB_ROUNDS=2 bash scripts/ab_libs.sh "long_div8|-|--workload long --steps 16" "long_div32|$A/libmcq_hip_qdiv32.so|--workload long --steps 16" "long_div2|$A/libmcq_hip_qdiv2.so|--workload long --steps 16"
AB_ROUNDS=1 AB_TIMEOUT=400 bash scripts/ab_libs.sh "refseq_static|-|--refseq-scale --steps 10 --warmup 3" "refseq_dynq8|$A/libmcq_hip_dynq8.so|--refseq-scale --steps 10 --warmup 3" "refseq_dynq32|$A/libmcq_hip_dynq32.so|--refseq-scale --steps 10 --warmup 3" \
   "refseqp_static|-|--refseq-scale --workload paired --steps 10 --warmup 3" "refseqp_dynq8|$A/libmcq_hip_dynq8.so|--refseq-scale --workload paired --steps 10 --warmup 3"
