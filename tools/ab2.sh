#!/bin/bash
# usage: ab2.sh "<bench args>" "<name>:<env>" ...
ARGS=$1; shift
for i in 1 2; do
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  out=$(env $envs timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline 2>gpurun_out/ab_err.log | tail -1)
  echo "$ARGS | $name $(echo $out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k: v["us"] for k, v in d["kernels"].items()})')"
done; done
