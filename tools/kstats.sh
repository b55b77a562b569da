#!/bin/bash
# usage (GPU box): bash tools/kstats.sh <tag> [ENV=VAL ...]  -- rocprofv3 kernel stats of a short bench run, printed compactly
TAG=$1; shift
export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 2 > gpurun_out/$TAG.log 2>&1
S=$(find gpurun_out/$TAG -name "*kernel_stats.csv")
python - "$S" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].split("(")[0].replace("void mgacbam::","")[:40]
    if int(r["Calls"])>50: print("%-42s calls=%4s avg=%8.1f us min=%8.1f max=%8.1f"%(n,r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
