#!/bin/bash
run() { echo "## $*"; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 20 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
run A=1
run MGACBAM_POOL_CPT=1
run MGACBAM_POOL_CPT=4
run MGACBAM_POOL_TX=64
run MGACBAM_POOL_TX=64 MGACBAM_POOL_CPT=1
run MGACBAM_POOL_TX=128
