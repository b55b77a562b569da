#!/bin/bash
run() { echo "## $*"; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 20 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
for r in 1 2; do for n in base e4 e8 e16; do run MGACBAM_LIB=$PWD/mga_yolo_amd/libmgacbam_$n.so; done; done
