#!/bin/bash
# end-of-round measurements on one GPU box: kernel traces + PMC passes per workload, bench lines per workload (round 3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c2 paired long; do
  bash scripts/profile.sh r03_$w --workload $w > gpurun_out/r03_${w}_profile.log 2>&1 || echo "profile $w failed"
  cp profiles/r03_${w}_*.csv gpurun_out/ 2>/dev/null
  echo "profiled $w"
done
bash scripts/profile.sh r03_refseq --refseq-scale > gpurun_out/r03_refseq_profile.log 2>&1 || echo "profile refseq failed"
cp profiles/r03_refseq_*.csv gpurun_out/ 2>/dev/null
echo "profiled refseq"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_line_c2_driver_form.json 2> gpurun_out/r03_bench_c2d.err; echo "c2 (driver form) rc $?"
python3 bench.py > gpurun_out/r03_bench_line_c2.json 2> gpurun_out/r03_bench_c2.err; echo "c2 rc $?"
python3 bench.py --workload paired --steps 24 > gpurun_out/r03_bench_line_paired.json 2> gpurun_out/r03_bench_paired.err; echo "paired rc $?"
python3 bench.py --workload long --steps 24 > gpurun_out/r03_bench_line_long.json 2> gpurun_out/r03_bench_long.err; echo "long rc $?"
python3 bench.py --mode sharded --steps 24 > gpurun_out/r03_bench_line_sharded_n1.json 2> gpurun_out/r03_bench_sharded.err; echo "sharded rc $?"
python3 bench.py --emulate-ranks 8 --max-cand 4 --steps 24 --no-pcie-leg > gpurun_out/r03_bench_line_c2_P8_M4.json 2> gpurun_out/r03_bench_p8m4.err; echo "P8 M4 rc $?"
python3 bench.py --contigs 132 --long-genome-mbp 16 --steps 24 --no-pcie-leg > gpurun_out/r03_bench_line_c2_refseqlike_targets.json 2> gpurun_out/r03_bench_rl.err; echo "refseq-like targets rc $?"
python3 bench.py --refseq-scale --steps 16 --warmup 2 > gpurun_out/r03_bench_line_refseq.json 2> gpurun_out/r03_bench_refseq.err; echo "refseq rc $?"
python3 bench.py --refseq-scale --workload paired --steps 16 --warmup 2 --no-pcie-leg > gpurun_out/r03_bench_line_refseq_paired.json 2> gpurun_out/r03_bench_refseq_p.err; echo "refseq paired rc $?"
rm -rf gpurun_out/r03_*_trace gpurun_out/r03_*_pmc[0-9]
