#!/bin/bash
# third wave stage with the 64-register form (libmcq_hip_w64.so) against the library in the tree: parity, then RefSeq-scale lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MCQ_HIP_LIB=$GRAFT_REPO_ROOT/scripts/_ab/libmcq_hip_w64.so timeout -k 10 900 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_refseq_scale.py tests/test_gpu_shard_native.py -m gpu -x -q > gpurun_out/w64_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/w64_tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit 1; fi
AB_ROUNDS=1 AB_TIMEOUT=500 bash scripts/ab_libs.sh "refseqp_cur|-|--refseq-scale --workload paired --steps 16 --warmup 2" "refseqp_w64|scripts/_ab/libmcq_hip_w64.so|--refseq-scale --workload paired --steps 16 --warmup 2" \
    "refseq_cur|-|--refseq-scale --steps 16 --warmup 2" "refseq_w64|scripts/_ab/libmcq_hip_w64.so|--refseq-scale --steps 16 --warmup 2"
