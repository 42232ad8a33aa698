#!/bin/bash
# two open questions: table load factor at RefSeq scale (MCQ_SLOTS_PER_KEY=4), and where 33 Gbp pairs stand with this round's kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MCQ_SLOTS_PER_KEY=4 timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-pcie-leg --refseq-scale --steps 16 --warmup 2 > gpurun_out/exp_refseq_spk4.json 2> gpurun_out/exp_refseq_spk4.err; echo "refseq spk4 rc $?"
timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-pcie-leg --species 800 --workload paired --steps 16 --warmup 2 > gpurun_out/exp_p33.json 2> gpurun_out/exp_p33.err; echo "33 Gbp paired rc $?"
timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-pcie-leg --species 800 --steps 16 --warmup 2 > gpurun_out/exp_s33.json 2> gpurun_out/exp_s33.err; echo "33 Gbp single rc $?"
python3 - <<'PY'
import json
for f in ('exp_refseq_spk4','exp_p33','exp_s33'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); rf=d['roofline']
        print(f, '%.3f ms'%d['ms_per_step'], ' '.join('%.3f'%v for v in rf['kernel_ms'].values()), rf['per_launch'], d['config']['db_layout']['slots_per_key'], d['config']['db_hbm_bytes']/1e9)
    except Exception as e: print(f, 'unreadable', e)
PY
