#!/bin/bash
# the one-block / two-register sorting networks: unit test against std::sort, parity suites, then same-box A/B against the previous commit's library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_sort_networks.py tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q > gpurun_out/sort_tests.log 2>&1 || { tail -40 gpurun_out/sort_tests.log; exit 1; }
tail -3 gpurun_out/sort_tests.log
AB_ROUNDS=2 scripts/ab_libs.sh "c2_head|scripts/_ab/libmcq_hip_head.so|--steps 32" "c2_new|-|--steps 32" "long_head|scripts/_ab/libmcq_hip_head.so|--workload long --steps 16" "long_new|-|--workload long --steps 16" "paired_head|scripts/_ab/libmcq_hip_head.so|--workload paired --steps 32" "paired_new|-|--workload paired --steps 32" || exit 1
AB_ROUNDS=1 AB_TIMEOUT=500 scripts/ab_libs.sh "p33_head|scripts/_ab/libmcq_hip_head.so|--species 800 --workload paired --steps 16 --warmup 2" "p33_new|-|--species 800 --workload paired --steps 16 --warmup 2" "refseq_head|scripts/_ab/libmcq_hip_head.so|--refseq-scale --steps 8" "refseq_new|-|--refseq-scale --steps 8" "refseqp_new|-|--refseq-scale --workload paired --steps 8"
