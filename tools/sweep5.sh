#!/bin/bash
run() { echo "## $*"; env "$@" python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline --kernel-reps 10 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
run A=1
run MGACBAM_POOL_CPT=1
run MGACBAM_POOL_CPT=2
run MGACBAM_LIB=$PWD/mga_yolo_amd/libmgacbam_pf2.so
run MGACBAM_LIB=$PWD/mga_yolo_amd/libmgacbam_pf4.so
run MGACBAM_LIB=$PWD/mga_yolo_amd/libmgacbam_pf4.so MGACBAM_POOL_CPT=2
run MGACBAM_LIB=$PWD/mga_yolo_amd/libmgacbam_pf4.so MGACBAM_POOL_CPT=1
