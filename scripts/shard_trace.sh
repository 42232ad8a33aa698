#!/bin/bash
# kernel timeline of the sharded leg in its steady state (which kernels overlap): rocprofv3 --kernel-trace, condensed
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/st && mkdir -p /tmp/st
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 bench.py --mode sharded --steps 8 --warmup 3 --no-cpu-baseline --no-pcie-leg --no-replicas-leg $@ > gpurun_out/st.json 2> gpurun_out/st.err
f=$(find /tmp/st -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows
      ]
ks.sort()
t0 = ks[0][0]
out = open("gpurun_out/shard_timeline.txt", "w")
for s, e, n, q in ks[-60:]:
    out.write("%10.3f %10.3f %8.3f  q%s %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
out.close()
mins = {}
for s_, e, n, q in ks:
    mins[n] = min(mins.get(n, 1e18), e - s_)
print("shortest run of each kernel (ms):", {n: round(v / 1e6, 3) for n, v in mins.items() if v > 50000})
PY
tail -45 gpurun_out/shard_timeline.txt
