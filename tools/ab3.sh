#!/bin/bash
# A/B: kGateR = 8 build and k_bwd_reduce2's channels per row
OUT=gpurun_out/${1:-ab3}; mkdir -p $OUT
run() { wl=$1; tag=$2; shift 2; env "$@" python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-eager --kernel-reps 20 > $OUT/${wl}_$tag.json 2> $OUT/${wl}_$tag.err || echo "$wl $tag failed"; }
for wl in cfg2 cfg3 cfg4; do
run $wl base X=1
run $wl r8 MGACBAM_LIB=build/variants/libmgacbam_r8.so
run $wl r2cpt2 MGACBAM_R2_CPT=2
run $wl base2 X=1
done
python tools/show_bench.py $OUT/*.json | grep -v "dominant\|cpu"
