#!/bin/bash
# one selection for all ranks (OptDev::lin) against the lists + fold one by one (MCQ_FOLD_BY_LISTS = 0x8000): parity, then bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_params.py tests/test_gpu_rows.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/lin_tests.log 2>&1
rc=$?; tail -5 gpurun_out/lin_tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit 1; fi
AB_ROUNDS=2 bash scripts/ab_libs.sh "c2_lin|-|--steps 24" "c2_linonly|scripts/_ab/libmcq_hip_linonly.so|--steps 24" "c2_lists|-|--steps 24 --query-flags 0x8000" \
   "paired_lin|-|--steps 24 --workload paired" "paired_linonly|scripts/_ab/libmcq_hip_linonly.so|--steps 24 --workload paired" \
   "long_lin|-|--steps 24 --workload long" "long_lists|-|--steps 24 --workload long --query-flags 0x8000" \
   "P64M4_lin|-|--steps 24 --emulate-ranks 64 --max-cand 4" "P8M4_lin|-|--steps 24 --emulate-ranks 8 --max-cand 4"
AB_ROUNDS=1 AB_TIMEOUT=500 bash scripts/ab_libs.sh "refseq_lin|-|--refseq-scale --steps 16 --warmup 2" "refseq_linonly|scripts/_ab/libmcq_hip_linonly.so|--refseq-scale --steps 16 --warmup 2"
