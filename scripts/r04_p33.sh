#!/bin/bash
# where the 33 Gbp table (--species 800) stands with the final kernels: pairs and single-end
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_ROUNDS=1 AB_TIMEOUT=500 scripts/ab_libs.sh "p33_r03|scripts/_ab/libmcq_hip_r03.so|--species 800 --workload paired --steps 16 --warmup 2" "p33|-|--species 800 --workload paired --steps 16 --warmup 2" "s33|-|--species 800 --steps 16 --warmup 2"
