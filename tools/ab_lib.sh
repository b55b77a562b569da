#!/bin/bash
# usage: ab_lib.sh <workload> <name>:<lib> ...   interleaved A/B of library builds on the CBAM step, per-kernel times included
WL=$1; shift
for i in 1 2; do
for v in "$@"; do
  name=${v%%:*}; lib=${v#*:}
  out=$(MGACBAM_LIB=$lib timeout -k 10 300 python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-eager --no-harness 2>gpurun_out/ab_err.log | tail -1)
  echo "$WL $name $(echo $out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], {k: v["us"] for k, v in d["kernels"].items()})')"
done; done
