#!/bin/bash
# the two-class passes with a lane mask on the partial row only: same-box A/B against the previous commit's library first, parity after it (RefSeq-scale tests included)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_ROUNDS=1 AB_TIMEOUT=500 scripts/ab_libs.sh "refseq_prev|scripts/_ab/libmcq_hip_prev.so|--refseq-scale --steps 8" "refseq_new|-|--refseq-scale --steps 8" "refseqp_new|-|--refseq-scale --workload paired --steps 8" "p33_new|-|--species 800 --workload paired --steps 16 --warmup 2" || exit 1
timeout -k 10 1000 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_refseq_scale.py -x -q > gpurun_out/rows_tests.log 2>&1 || { tail -40 gpurun_out/rows_tests.log; exit 1; }
tail -3 gpurun_out/rows_tests.log
