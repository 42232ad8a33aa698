#!/bin/bash
# end-of-round measurements on one GPU box: kernel traces + PMC passes per workload, bench lines per workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c2 paired long; do
  bash scripts/profile.sh r02_$w --workload $w > gpurun_out/r02_${w}_profile.log 2>&1 || echo "profile $w failed"
  cp profiles/r02_${w}_*.csv gpurun_out/ 2>/dev/null
  echo "profiled $w"
done
python3 bench.py > gpurun_out/r02_bench_line_c2.json 2> gpurun_out/r02_bench_c2.err; echo "c2 rc $?"
python3 bench.py --workload paired --steps 24 > gpurun_out/r02_bench_line_paired.json 2> gpurun_out/r02_bench_paired.err; echo "paired rc $?"
python3 bench.py --workload long --steps 24 > gpurun_out/r02_bench_line_long.json 2> gpurun_out/r02_bench_long.err; echo "long rc $?"
python3 bench.py --mode sharded --steps 24 > gpurun_out/r02_bench_line_sharded_n1.json 2> gpurun_out/r02_bench_sharded.err; echo "sharded rc $?"
# keep only the summaries (the raw rocprofv3 output is large)
rm -rf gpurun_out/r02_*_trace gpurun_out/r02_*_pmc[0-9]
