#!/bin/bash
# same-box A/B/C of an environment switch: tools/ab_env3.sh VAR "a b c" [bench.py args...]   (alternates the values, three rounds)
VAR=$1; VALS=$2; shift 2
for i in 1 2 3; do
  for v in $VALS; do
    env $VAR=$v python bench.py "$@" --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['config']['repeats_ms_per_step'])"
  done
done
