#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace + stats, then HBM traffic counters in SEPARATE passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (on the GPU box): bash tools/profile.sh <tag> [workload]
set -e
TAG=${1:-prof}; WL=${2:-cfg2}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-eager --kernel-reps 10"
python bench.py --workload $WL --no-cpu-baseline --no-eager > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -- python $ARGS > $OUT/trace_$WL.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$WL -- python $ARGS > $OUT/pmc_fetch_$WL.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$WL -- python $ARGS > $OUT/pmc_write_$WL.log 2>&1
find $OUT -name "*.csv" | head -20
