#!/bin/bash
OUT=gpurun_out/${1:-ab4}; mkdir -p $OUT
run() { wl=$1; tag=$2; shift 2; env "$@" python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-eager --kernel-reps 30 > $OUT/${wl}_$tag.json 2> $OUT/${wl}_$tag.err || echo "$wl $tag failed"; }
for wl in cfg2 cfg3; do
run $wl fat1 MGACBAM_WSA_FAT=1
run $wl fat0 MGACBAM_WSA_FAT=0
run $wl fat1b MGACBAM_WSA_FAT=1
run $wl fat0b MGACBAM_WSA_FAT=0
done
python tools/show_bench.py $OUT/*.json | grep -v "dominant\|cpu"
