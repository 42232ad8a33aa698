#!/bin/bash
# round 4: the workgroup kernels' queue -- entries from a shared cursor (working tree) against the static assignment (same sources,
# -DMCQ_BLOCK_STATIC_QUEUE) and the r03 library, on long reads and on the RefSeq-scale table (a million mostly empty entries)
cd $GRAFT_REPO_ROOT
A=scripts/_ab
AB_ROUNDS=2 bash scripts/ab_libs.sh "long_r03|$A/libmcq_hip_r03.so|--workload long --steps 16" "long_static|$A/libmcq_hip_staticq.so|--workload long --steps 16" "long_r04|-|--workload long --steps 16"
AB_ROUNDS=1 AB_TIMEOUT=400 bash scripts/ab_libs.sh "refseqp_static|$A/libmcq_hip_staticq.so|--refseq-scale --workload paired --steps 10 --warmup 3" "refseqp_r04|-|--refseq-scale --workload paired --steps 10 --warmup 3"
