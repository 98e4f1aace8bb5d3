#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic counters (separate passes) for bench.py with a
# short CG run, so that the fused CG kernels appear.  Usage: bash tools/profile_cg_pmc.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/profcgpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-cfg3 --cg-iters 8"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/pmc_write.err
echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > /dev/null 2> $OUT/pmc_sq.err
echo "sq rc=$?"
du -sh $OUT
