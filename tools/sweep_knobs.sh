#!/bin/bash
# Sweep of the launch-geometry / cache-policy knobs of the experiments build on the configs[1] pair (one box, alternating).
export PTYCHO_HIP_LIB=tools/build/libptychohip_exp.so
run() { echo "== $*"; env "$@" python tools/pair_time.py | tail -1; }
run A=0
for nt in 3 7 11 15 0; do run PTYCHO_HIP_NT=$nt; done
for g in 8 16 64 128; do run PTYCHO_HIP_ROWGRID=$g; done
for m in 8 24 32; do run PTYCHO_HIP_MINSEG=$m; done
for c in 32 48 96 128; do run PTYCHO_HIP_COLSEGS=$c; done
run A=0
