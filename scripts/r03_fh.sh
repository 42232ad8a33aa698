#!/bin/bash
# probe hand-over to the third wave stage (library in the tree) against HEAD (scripts/_ab/libmcq_hip_head.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_refseq_scale.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/fh_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/fh_tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit 1; fi
AB_ROUNDS=2 bash scripts/ab_libs.sh "c2_new|-|--steps 24" "c2_head|scripts/_ab/libmcq_hip_head.so|--steps 24" "paired_new|-|--steps 24 --workload paired" "paired_head|scripts/_ab/libmcq_hip_head.so|--steps 24 --workload paired"
AB_ROUNDS=1 AB_TIMEOUT=500 bash scripts/ab_libs.sh "refseqp_new|-|--refseq-scale --workload paired --steps 16 --warmup 2" "refseqp_head|scripts/_ab/libmcq_hip_head.so|--refseq-scale --workload paired --steps 16 --warmup 2" \
    "refseq_new|-|--refseq-scale --steps 16 --warmup 2" "refseq_head|scripts/_ab/libmcq_hip_head.so|--refseq-scale --steps 16 --warmup 2"
