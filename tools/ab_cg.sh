#!/bin/bash
# A/B of library variants on the CG loop at 4096 positions (screened probe: tools/cg_screened.py)
for rep in 1 2; do
  echo "== shipped"; python tools/cg_screened.py | tail -1
  for lib in tools/build/libptychohip_ab*.so; do
    echo "== $lib"; PTYCHO_HIP_LIB=$lib python tools/cg_screened.py | tail -1
  done
done
