#!/bin/bash
# A/B of the staggered sweep start and the k_chan tile width at cfg4 and cfg2
OUT=gpurun_out/${1:-sweep5}; mkdir -p $OUT
run() { wl=$1; tag=$2; shift 2; env "$@" python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-eager --kernel-reps 20 > $OUT/${wl}_$tag.json 2> $OUT/${wl}_$tag.err || echo "$wl $tag failed"; }
for wl in cfg4 cfg2 cfg3; do
run $wl rot1 MGACBAM_POOL_ROT=1
run $wl rot0 MGACBAM_POOL_ROT=0
run $wl cf32 MGACBAM_CHANF_TX=32
run $wl cf16 MGACBAM_CHANF_TX=16
run $wl rot1b MGACBAM_POOL_ROT=1
done
python tools/show_bench.py $OUT/*.json
