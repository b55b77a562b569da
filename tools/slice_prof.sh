#!/bin/bash
# per-kernel split of the layer-loop slice step: rocprofv3 --kernel-trace --stats of tools/slice_bench.py; usage: bash tools/slice_prof.sh <outdir> [workload]
OUT=${1:-gpurun_out/slice_prof}; WL=${2:-cfg2}; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -- python tools/slice_bench.py $WL 60 > $OUT/trace_$WL.log 2>&1
grep images_per $OUT/trace_$WL.log | cut -c1-120
S=$(find $OUT/trace_$WL -name "*kernel_stats.csv")
python - "$S" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=0
for r in rows[:22]:
    n=r["Name"].replace("void mgacbam::","").replace("mgacbam::","")[:64]
    if int(r["Calls"])>=60:
        tot+=float(r["AverageNs"])/1e3
        print("%-66s calls=%5s avg=%8.1f us"%(n,r["Calls"],float(r["AverageNs"])/1e3))
print("sum of per-step kernels: %.1f us"%tot)
PY
