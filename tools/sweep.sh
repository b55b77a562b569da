#!/bin/bash
# Launch-geometry / build-variant sweep on the GPU box: one line of per-kernel in-step durations per setting.
#   bash tools/sweep.sh [workload]          (default cfg2)
# Settings are environment overrides read by the library (MGACBAM_POOL_TX, MGACBAM_POOL_CPT, MGACBAM_CHAN_TX,
# MGACBAM_CHAN_MINTX, MGACBAM_LEVEL_ORDER, MGACBAM_NT, MGACBAM_HALF_VEC) or MGACBAM_LIB=<path> for an A/B build made with
# mga_yolo_amd.build.build(defines=[...], out=...).  Differences below ~2 % are run-to-run / box-to-box noise.
WL=${1:-cfg2}
run() { echo "## $*"; env "$@" python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 20 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
run BASE=1
for tx in 16 32 64; do run MGACBAM_CHAN_TX=$tx; done
for cpt in 1 2 4; do run MGACBAM_POOL_CPT=$cpt; done
for tx in 32 64 128 256; do run MGACBAM_POOL_TX=$tx; done
run MGACBAM_LEVEL_ORDER=0
run MGACBAM_CHAN_MINTX=8
run MGACBAM_NT=0
run BASE=2
