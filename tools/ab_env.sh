#!/bin/bash
# usage: ab_env.sh <workload> "<name>:<ENV=VAL ...>" ...   interleaved A/B of environment settings on the CBAM step, per-kernel times included
WL=$1; shift
for i in 1 2; do
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  out=$(env $envs timeout -k 10 300 python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-eager --no-harness 2>gpurun_out/ab_err.log | tail -1)
  echo "$WL $name $(echo $out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], {k: v["us"] for k, v in d["kernels"].items()})')"
done; done
