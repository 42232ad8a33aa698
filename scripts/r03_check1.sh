#!/bin/bash
# round 3, first GPU pass: whole GPU suite, then the configs[1] line in the three location forms / two layouts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
free -g > gpurun_out/r03_box.txt; nproc >> gpurun_out/r03_box.txt; rocm-smi --showmeminfo vram >> gpurun_out/r03_box.txt 2>&1
step() {   # name, timeout, command...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.out 2> gpurun_out/$name.err; local rc=$?
  echo "$name rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
}
step r03a_pytest 900 python3 -m pytest tests -q -m gpu -x
tail -5 gpurun_out/r03a_pytest.out
step r03a_c2 200 python3 bench.py --steps 48
step r03a_c2_gw 200 python3 bench.py --steps 48 --loc-format gw --no-cpu-baseline --no-pcie-leg
step r03a_c2_s16 200 python3 bench.py --steps 48 --bucket-bytes 16 --no-cpu-baseline --no-pcie-leg
step r03a_c2_gw_s16 200 python3 bench.py --steps 48 --loc-format gw --bucket-bytes 16 --no-cpu-baseline --no-pcie-leg
step r03a_c2_refseqlike 300 python3 bench.py --steps 48 --contigs 132 --long-genome-mbp 16
step r03a_c2_f64 200 python3 bench.py --steps 24 --loc-format fields64 --no-cpu-baseline --no-pcie-leg
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a_c2*.out')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']; c=d['config']
        print(f.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'frac %.4f' % r['frac'], c['db_layout']['loc_format'], c['db_layout']['bucket_bytes'], 'targets', c['db_targets'],
              'db GB %.1f' % (c['db_hbm_bytes']/1e9), 'cpu ok', d.get('cpu_baseline',{}).get('gpu_matches_cpu_on_first_batch'), 'kernel_ms', {k:round(v,4) for k,v in r['kernel_ms'].items()})
    except Exception as e:
        print(f, 'unreadable', e)
PY
