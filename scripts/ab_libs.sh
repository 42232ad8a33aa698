#!/bin/bash
# same-box comparison of bench lines: every argument is "label|MCQ_HIP_LIB or -|bench.py arguments"; the whole list is run
# AB_ROUNDS times (default 2) in order, so that drift of the box shows.  Lines go to gpurun_out/ab_<label>_<round>.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in $(seq 1 ${AB_ROUNDS:-2}); do
  for spec in "$@"; do
    IFS='|' read -r label lib args <<< "$spec"
    if [ "$lib" = "-" ]; then unset MCQ_HIP_LIB; else export MCQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
    timeout -k 10 ${AB_TIMEOUT:-300} python3 bench.py --no-cpu-baseline --no-pcie-leg $args > gpurun_out/ab_${label}_$r.json 2> gpurun_out/ab_${label}_$r.err
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$label killed"; exit 1; fi
    python3 - "$label" "$r" "$rc" <<'PY'
import json,sys
label,r,rc=sys.argv[1:4]
try:
    d=json.loads(open('gpurun_out/ab_%s_%s.json'%(label,r)).read().strip().splitlines()[-1])
    rf=d['roofline']; c=d['config']; lay=c.get('db_layout',{})
    print('%-22s round %s  ms/step %.4f  kernels %s  T/read %.1f ovf %.3f  %s/%s  db %.1f GB' % (label, r, d['ms_per_step'],
          ' '.join('%.3f'%v for v in rf['kernel_ms'].values()) + ' two-class %s retry %s narrow %s' % (rf['per_launch'].get('n_two_class'), rf['per_launch'].get('n_two_class_retry'), rf['per_launch'].get('n_narrow_queued')), rf['per_launch']['n_locations']/ (c['reads_total']/d['steps']/d['n_gpus']),
          rf['per_launch']['n_overflow']/(c['reads_total']/d['steps']/d['n_gpus']), lay.get('loc_format'), lay.get('bucket_bytes'), c['db_hbm_bytes']/1e9))
except Exception as e:
    print(label, 'round', r, 'rc', rc, 'unreadable:', e)
PY
  done
done
