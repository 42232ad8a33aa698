#!/bin/bash
# parity of the 8193..16384-word path of k_query_block, then its same-box A/B against HEAD (scripts/_ab/libmcq_hip_head.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q > gpurun_out/mid_tests.log 2>&1 || { tail -30 gpurun_out/mid_tests.log; exit 1; }
tail -3 gpurun_out/mid_tests.log
AB_ROUNDS=2 scripts/ab_libs.sh "long_head|scripts/_ab/libmcq_hip_head.so|--workload long --steps 16" "long_mid|-|--workload long --steps 16" || exit 1
timeout -k 10 300 python3 scripts/long_by_length.py > gpurun_out/long_by_length_mid.log 2>&1 || { tail gpurun_out/long_by_length_mid.log; exit 1; }
cat gpurun_out/long_by_length_mid.log
