#!/bin/bash
# same-box A/B of two builds: ab.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
for rep in 1 2 3; do for L in $A $B; do
  v=$(RGM_LIB_PATH=$PWD/rule-guided-music_amd/rgm/$L python bench.py --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$L $v"
done; done
