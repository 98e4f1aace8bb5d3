#!/bin/bash
# A/B of kernel variants on one box at configs[2]'s operator pair (4096 x 512^2): shipped library against tools/build/libptychohip_ab<mask>.so
for rep in 1 2; do
  echo "== shipped"; python tools/pair_time512.py | tail -1
  for lib in tools/build/libptychohip_ab*.so; do
    echo "== $lib"; PTYCHO_HIP_LIB=$lib python tools/pair_time512.py | tail -1
  done
done
