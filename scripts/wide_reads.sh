#!/bin/bash
# reads of 5..8 windows (65..128 features): second wave stage (two features per lane) against the workgroup kernel
# (--query-flags 0x800 = MCQ_NO_WAVE16); 2 x 250 bp pairs and 500 bp single-end on the C2 table
cd $GRAFT_REPO_ROOT
for cfg in "--workload paired --read-len 250" "--workload c2 --read-len 500"; do
  for fl in 0 0x800; do
    timeout -k 10 400 python3 bench.py --steps 12 --no-cpu-baseline --batch 524288 $cfg --query-flags $fl $WR_ARGS > gpurun_out/wr.json 2> gpurun_out/wr.err || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/wr.json')); print('$cfg flags $fl: %.3g reads/s  %.3f ms/step' % (d['value'], d['ms_per_step']), {k: round(v/524288,2) for k,v in d['roofline']['per_launch'].items()})"
  done
done
