#!/bin/bash
# geometry sweep on the GPU box: prints one line per setting (per-kernel us)
run() { echo "## $*"; env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --kernel-reps 20 2>/dev/null | python tools/show_bench.py /dev/stdin | grep -E "value|f.pool"; }
run A=1
for tx in 16 32 64; do run MGACBAM_CHAN_TX=$tx; done
for cpt in 1 2 4; do run MGACBAM_POOL_CPT=$cpt MGACBAM_APPLY_CPT=$cpt; done
for tx in 32 64 128 256; do run MGACBAM_POOL_TX=$tx MGACBAM_APPLY_TX=$tx; done
for tx in 64 256; do for cpt in 2 4; do run MGACBAM_POOL_TX=$tx MGACBAM_APPLY_TX=$tx MGACBAM_POOL_CPT=$cpt MGACBAM_APPLY_CPT=$cpt; done; done
