#!/bin/bash
# HBM traffic per kernel of the layer-loop slice step: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md's
# recipe: FETCH_SIZE is in 32-byte units x2 on gfx950 ... see tools/summarize_profile.py); usage: bash tools/slice_pmc.sh <outdir> [workload]
OUT=${1:-gpurun_out/slice_pmc}; WL=${2:-cfg2}; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$tag -- python tools/slice_bench.py $WL 20 > $OUT/pmc_$tag.log 2>&1
  F=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$F" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r["Kernel_Name"].replace("void mgacbam::","").replace("mgacbam::","")[:50]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,d in sorted(acc.items()):
    if "head" in n or "k_" in n:
        print("%-52s"%n, "  ".join("%s=%.4g (n=%d)"%(k,sum(v)/len(v),len(v)) for k,v in d.items()))
PY
done
