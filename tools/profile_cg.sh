#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats of bench.py INCLUDING the 50-iteration CG run.
# Usage: bash tools/profile_cg.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/profcg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
find $OUT -name "*kernel_stats.csv" | head
