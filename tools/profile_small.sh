#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic (PMC, separate passes) of the operator kernels at ndet 112 and 128 (tools/prof_small.py) and the
# per-kernel stats of the DEFAULT bench command.  Usage: bash tools/profile_small.sh <tag>
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/profsmall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_small.py > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_small.py > /dev/null 2> $OUT/pmc_write.err
echo "write rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_default -- python3 bench.py --no-cpu > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"
find $OUT -name "*kernel_stats.csv"
