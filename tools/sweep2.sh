#!/bin/bash
run() { echo "## $*"; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 20 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
run MGACBAM_LEVEL_ORDER=0
run MGACBAM_LEVEL_ORDER=1
run MGACBAM_LEVEL_ORDER=1 MGACBAM_CHAN_MINTX=8
run MGACBAM_LEVEL_ORDER=1 MGACBAM_CHAN_MINTX=4
run MGACBAM_LEVEL_ORDER=0 MGACBAM_CHAN_MINTX=4
run MGACBAM_LEVEL_ORDER=1 MGACBAM_CHAN_MINTX=4 MGACBAM_POOL_CPT=1
run MGACBAM_LEVEL_ORDER=1 MGACBAM_CHAN_MINTX=4 MGACBAM_POOL_CPT=4
