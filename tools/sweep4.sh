#!/bin/bash
run() { echo "## $*"; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 20 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
for pf in 1 2 4; do for cpt in 1 2 4; do run MGACBAM_LIB=$PWD/mga_yolo_amd/libmgacbam_pf$pf.so MGACBAM_POOL_CPT=$cpt; done; done
