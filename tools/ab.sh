#!/bin/bash
# usage: ab.sh "<name>:<env assignments>" ...   (each variant twice, interleaved)
for i in 1 2; do
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  out=$(env $envs timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>gpurun_out/ab_err.log | tail -1)
  echo "$name $(echo $out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
