#!/bin/bash
# the RefSeq-scale shape (BASELINE configs[2] on one GPU): a small rehearsal first, then the >= 100 Gbp table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MCQ_BUILD_TRACE=1
timeout -k 10 300 python3 bench.py --refseq-scale --species 100 --steps 8 --warmup 2 > gpurun_out/r03_refseq_small.json 2> gpurun_out/r03_refseq_small.err
rc=$?; echo "small rc $rc"; tail -c 600 gpurun_out/r03_refseq_small.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_refseq_small.json').read().strip().splitlines()[-1]); c=d['config']
print('small: ms/step %.3f' % d['ms_per_step'], c['db_layout'], 'build s', c['db_build_s'], 'parts', c['db_build_parts'], 'cpu', d.get('cpu_baseline'))"
timeout -k 10 ${BIG_TIMEOUT:-800} python3 bench.py --refseq-scale --steps ${BIG_STEPS:-12} --warmup 2 $BIG_ARGS > gpurun_out/r03_refseq_big.json 2> gpurun_out/r03_refseq_big.err
rc=$?; echo "big rc $rc"; tail -c 1500 gpurun_out/r03_refseq_big.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_refseq_big.json').read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('big: ms/step %.3f' % d['ms_per_step'], c['workload']); print(c['db_layout']); print('build s', c['db_build_s'], 'parts', c['db_build_parts'], 'genomes s', c['genomes_s'], 'setup', c['setup_s'])
print('kernels', r['kernel_ms'], 'frac', r['frac'], 'per launch', r['per_launch']); print('cpu', d.get('cpu_baseline'))"
