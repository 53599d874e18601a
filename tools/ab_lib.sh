#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <other .so> <bench.py args...>   (alternates other / product, twice)
OTHER=$1; shift
for i in 1 2; do
  for lib in $OTHER ""; do
    if [ -n "$lib" ]; then export RGM_LIB_PATH=$PWD/$lib; else unset RGM_LIB_PATH; fi
    python bench.py "$@" --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-product}', d['ms_per_step'])"
  done
done
