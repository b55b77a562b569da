#!/bin/bash
# usage: ab_slice.sh <name>:<lib> ...   interleaved A/B of library builds on the layer-loop slice step (bench.py's layer_loop_slice)
for i in 1 2; do
for v in "$@"; do
  name=${v%%:*}; lib=${v#*:}
  out=$(MGACBAM_LIB=$lib timeout -k 10 300 python bench.py --no-harness --no-cpu-baseline --steps 50 2>gpurun_out/ab_err.log | tail -1)
  echo "$name $(echo $out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["layer_loop_slice"]["ms_per_step"], d["ms_per_step"])')"
done; done
