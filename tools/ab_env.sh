#!/bin/bash
# same-box A/B of an environment switch of the library: tools/ab_env.sh VAR a b [bench.py args...]   (alternates VAR=a / VAR=b, three rounds)
VAR=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v python bench.py "$@" --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['config']['repeats_ms_per_step'])"
  done
done
