#!/bin/bash
# round 4: r03 library against the working tree on long reads (direct entry by geometry) and on the RefSeq-scale table fused; then the
# sharded path on that table at one rank and as two ranks on the one GPU over gloo
cd $GRAFT_REPO_ROOT
A=scripts/_ab
AB_ROUNDS=2 bash scripts/ab_libs.sh "long_r03|$A/libmcq_hip_r03.so|--workload long --steps 16" "long_r04|-|--workload long --steps 16"
AB_ROUNDS=1 AB_TIMEOUT=400 bash scripts/ab_libs.sh "refseq_r04|-|--refseq-scale --steps 10 --warmup 3" "refseq_sharded_n1|-|--refseq-scale --mode sharded --no-replicas-leg --steps 10 --warmup 3"
timeout -k 10 700 python3 bench.py --refseq-scale --db-fixed --gpus 2 --backend gloo --mode sharded --no-replicas-leg --no-cpu-baseline --steps 6 --warmup 2 --sharded-timeout 600 \
  > gpurun_out/r04_bench_line_refseq_2ranks_one_gpu_gloo_rehearsal.json 2> gpurun_out/r04_rehearsal.err; echo "rehearsal rc $?"
