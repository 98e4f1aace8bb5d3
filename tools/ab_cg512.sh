#!/bin/bash
for rep in 1 2; do
  echo "== shipped"; python tools/cg512.py | tail -1
  for lib in tools/build/libptychohip_ab*.so; do
    echo "== $lib"; PTYCHO_HIP_LIB=$lib python tools/cg512.py | tail -1
  done
done
