#!/bin/bash
# A/B of kernel variants on one box: the shipped library against tools/build/libptychohip_ab<mask>.so, alternating
for rep in 1 2; do
  echo "== shipped"; python tools/pair_time.py "$@" | tail -2
  for lib in tools/build/libptychohip_ab*.so; do
    echo "== $lib"; PTYCHO_HIP_LIB=$lib python tools/pair_time.py "$@" | tail -2
  done
done
