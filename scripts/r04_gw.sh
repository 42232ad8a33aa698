#!/bin/bash
# parity of the one-round-trip LocGW look-up + one-chunk lists, then RefSeq-scale A/B against HEAD~ (scripts/_ab/libmcq_hip_head.so) and phase clocks
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_shard_native.py -x -q > gpurun_out/gw_tests.log 2>&1 || { tail -30 gpurun_out/gw_tests.log; exit 1; }
tail -3 gpurun_out/gw_tests.log
AB_ROUNDS=1 AB_TIMEOUT=400 scripts/ab_libs.sh "refseq_head|scripts/_ab/libmcq_hip_head.so|--refseq-scale --steps 8" "refseq_new|-|--refseq-scale --steps 8" "refseqp_head|scripts/_ab/libmcq_hip_head.so|--refseq-scale --workload paired --steps 8" "refseqp_new|-|--refseq-scale --workload paired --steps 8" || exit 1
scripts/r04_wclk.sh
