#!/bin/bash
run() { echo "## $*"; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 20 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
run A=1
for n in u1 u2 u8; do run MGACBAM_LIB=$PWD/mga_yolo_amd/libmgacbam_$n.so; done
run A=2
