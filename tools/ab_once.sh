#!/bin/bash
echo "== shipped"; python tools/pair_time.py | tail -1
for lib in tools/build/libptychohip_ab*.so; do
  echo "== $lib"; PTYCHO_HIP_LIB=$lib python tools/pair_time.py | tail -1
done
echo "== shipped"; python tools/pair_time.py | tail -1
