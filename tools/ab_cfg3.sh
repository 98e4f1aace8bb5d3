#!/bin/bash
# A/B of kernel variants on one box at configs[2]'s CG: the shipped library against tools/build/libptychohip_ab<mask>.so, alternating
for rep in 1 2; do
  echo "== shipped"; python tools/cfg3_cg.py 4 | tail -1
  for lib in tools/build/libptychohip_ab*.so; do
    echo "== $lib"; PTYCHO_HIP_LIB=$lib python tools/cfg3_cg.py 4 | tail -1
  done
done
