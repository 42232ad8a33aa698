#!/bin/bash
# throughput against database size: the C2 generator with 50 / 250 / 400 / 800 species (2 / 10 / 16 / 33 Gbp),
# single-end and paired reads; one JSON line per run under gpurun_out/
cd $GRAFT_REPO_ROOT
for sp in ${DBS_SPECIES:-50 250 400 800}; do
  for w in c2 paired; do
    timeout -k 10 500 python3 bench.py --species $sp --steps 24 --no-cpu-baseline --workload $w > gpurun_out/dbscale_${sp}_$w.json 2> gpurun_out/dbscale_${sp}_$w.err || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/dbscale_${sp}_$w.json')); r=d['roofline']; B=1<<20
print('species $sp $w: %.2f Gbp  %.3g reads/s  %.3f ms/step  T=%.0f  overflow=%.3f  keys=%.3g locs=%.3g  db=%.1f GB  build=%.2fs' % (
  float(d['config']['workload'].split('divergence, ')[1].split(' Gbp')[0]), d['value'], d['ms_per_step'], r['per_launch']['n_locations']/B*(2 if '$w'=='paired' else 1),
  r['per_launch']['n_overflow']/B*(2 if '$w'=='paired' else 1), d['config']['db_keys'], d['config']['db_locations'], d['config']['db_hbm_bytes']/1e9, d['config']['db_build_s']))"
  done
done
