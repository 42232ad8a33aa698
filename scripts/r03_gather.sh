#!/bin/bash
# bitmap gather up to 16 registers + prefix-maximum gather for 32 / 64 (library in the tree) against the bitmap form everywhere with
# scheduling barriers (scripts/_ab/libmcq_hip_sb.so = -DMCQ_GATHER_SB)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_refseq_scale.py -m gpu -x -q > gpurun_out/gather_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/gather_tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit 1; fi
AB_ROUNDS=1 AB_TIMEOUT=500 bash scripts/ab_libs.sh "refseqp_new|-|--refseq-scale --workload paired --steps 16 --warmup 2" "refseqp_sb|scripts/_ab/libmcq_hip_sb.so|--refseq-scale --workload paired --steps 16 --warmup 2" \
    "refseq_new|-|--refseq-scale --steps 16 --warmup 2" "refseq_sb|scripts/_ab/libmcq_hip_sb.so|--refseq-scale --steps 16 --warmup 2"
