#!/bin/bash
# workgroup kernels fetching 64 queue entries at once (library in the tree) against HEAD (scripts/_ab/libmcq_hip_head.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_refseq_scale.py tests/test_gpu_shard_native.py -m gpu -x -q > gpurun_out/qscan_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/qscan_tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit 1; fi
AB_ROUNDS=2 bash scripts/ab_libs.sh "long_new|-|--steps 24 --workload long" "long_head|scripts/_ab/libmcq_hip_head.so|--steps 24 --workload long" "c2_new|-|--steps 24" "c2_head|scripts/_ab/libmcq_hip_head.so|--steps 24"
AB_ROUNDS=1 AB_TIMEOUT=500 bash scripts/ab_libs.sh "refseqp_new|-|--refseq-scale --workload paired --steps 16 --warmup 2" "refseqp_head|scripts/_ab/libmcq_hip_head.so|--refseq-scale --workload paired --steps 16 --warmup 2" \
    "refseq_new|-|--refseq-scale --steps 16 --warmup 2" "refseq_head|scripts/_ab/libmcq_hip_head.so|--refseq-scale --steps 16 --warmup 2"
