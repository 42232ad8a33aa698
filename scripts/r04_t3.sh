#!/bin/bash
# per-pass copies of T in the two-class passes (scripts/_ab/libmcq_hip_t3.so) against the in-tree library: same-box A/B, then parity with the variant
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_ROUNDS=1 AB_TIMEOUT=500 scripts/ab_libs.sh "refseq_head|-|--refseq-scale --steps 8" "refseq_t3|scripts/_ab/libmcq_hip_t3.so|--refseq-scale --steps 8" "refseqp_head|-|--refseq-scale --workload paired --steps 8" "refseqp_t3|scripts/_ab/libmcq_hip_t3.so|--refseq-scale --workload paired --steps 8" "p33_t3|scripts/_ab/libmcq_hip_t3.so|--species 800 --workload paired --steps 16 --warmup 2" || exit 1
export MCQ_HIP_LIB=$GRAFT_REPO_ROOT/scripts/_ab/libmcq_hip_t3.so
timeout -k 10 1000 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_refseq_scale.py -x -q > gpurun_out/t3_tests.log 2>&1 || { tail -40 gpurun_out/t3_tests.log; exit 1; }
tail -3 gpurun_out/t3_tests.log
