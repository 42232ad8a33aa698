#!/bin/bash
# part B: RefSeq-scale pairs profiled, then the bench lines (profiles/pmc_traffic_*.json of part A are in the tree by now)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/profile.sh r04_refseqp --refseq-scale --workload paired > gpurun_out/r04_refseqp_profile.log 2>&1 || echo "profile refseqp failed"
cp profiles/r04_refseqp_*.csv gpurun_out/ 2>/dev/null
python3 scripts/pmc_traffic.py r04_refseqp refseqp ALL 1048576 > /dev/null; cp profiles/pmc_traffic_refseqp.json gpurun_out/
echo "profiled refseqp"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line_c2_driver_form.json 2> gpurun_out/r04_bench_c2d.err; echo "c2 (driver form) rc $?"
python3 bench.py --workload paired --steps 24 > gpurun_out/r04_bench_line_paired.json 2> gpurun_out/r04_bench_paired.err; echo "paired rc $?"
python3 bench.py --workload long --steps 24 > gpurun_out/r04_bench_line_long.json 2> gpurun_out/r04_bench_long.err; echo "long rc $?"
python3 bench.py --mode sharded --steps 24 > gpurun_out/r04_bench_line_sharded_n1.json 2> gpurun_out/r04_bench_sharded.err; echo "sharded rc $?"
python3 bench.py --refseq-scale --steps 16 --warmup 2 > gpurun_out/r04_bench_line_refseq.json 2> gpurun_out/r04_bench_refseq.err; echo "refseq rc $?"
python3 bench.py --refseq-scale --workload paired --steps 16 --warmup 2 --no-pcie-leg > gpurun_out/r04_bench_line_refseq_paired.json 2> gpurun_out/r04_bench_refseq_p.err; echo "refseq paired rc $?"
python3 bench.py --refseq-scale --mode sharded --steps 12 --warmup 3 > gpurun_out/r04_bench_line_refseq_sharded_n1.json 2> gpurun_out/r04_bench_refseq_sh.err; echo "refseq sharded rc $?"
rm -rf gpurun_out/r04_*_trace gpurun_out/r04_*_pmc[0-9]
