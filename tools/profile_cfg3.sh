#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel stats of the CG loops at configs[1] (tools/cg_screened.py) and configs[2]
# (tools/cfg3_cg.py), and the SQ wait / issue counters of the configs[2] loop (separate pass).  Usage: bash tools/profile_cfg3.sh <tag>
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/profcfg3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cg4096 -- python3 tools/cg_screened.py > $OUT/cg4096.log 2> $OUT/cg4096.err
echo "cg4096 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3 -- python3 tools/cfg3_cg.py 4 > $OUT/cfg3.log 2> $OUT/cfg3.err
echo "cfg3 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/cfg3_sq -- python3 tools/cfg3_cg.py 2 > /dev/null 2> $OUT/cfg3_sq.err
echo "cfg3 sq rc=$?"
find $OUT -name "*kernel_stats.csv"
du -sh $OUT
