#!/bin/bash
# launch-geometry sweep at a workload (default cfg4): one bench line per variant, per-kernel table printed by tools/show_bench.py
WL=${1:-cfg4}; OUT=gpurun_out/${2:-sweep4}; mkdir -p $OUT
run() { tag=$1; shift; env "$@" python bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-eager --kernel-reps 20 > $OUT/$tag.json 2> $OUT/$tag.err || echo "$tag failed"; }
run base X=1
run cpt1 MGACBAM_POOL_CPT=1
run cpt4 MGACBAM_POOL_CPT=4
run ptx128 MGACBAM_POOL_TX=128
run ptx64 MGACBAM_POOL_TX=64
run ctx32 MGACBAM_CHAN_TX=32
run ctx16 MGACBAM_CHAN_TX=16
run ctx64 MGACBAM_CHAN_TX=64
run order0 MGACBAM_LEVEL_ORDER=0
run nofuse MGACBAM_FUSE_FWD=0 MGACBAM_FOLD_BWD=0
run pf2 MGACBAM_LIB=build/variants/libmgacbam_pf2.so
run pf4 MGACBAM_LIB=build/variants/libmgacbam_pf4.so
run pf2cpt1 MGACBAM_LIB=build/variants/libmgacbam_pf2.so MGACBAM_POOL_CPT=1
run pf4cpt1 MGACBAM_LIB=build/variants/libmgacbam_pf4.so MGACBAM_POOL_CPT=1
python tools/show_bench.py $OUT/*.json
