#!/bin/bash
# usage: ab_env.sh VAR A B [bench args...] : ABBA of `bench.py` (headline precision only, 100 steps) with VAR=A and VAR=B on one box
var=$1; a=$2; b=$3; shift 3
for v in $a $b $b $a $a $b; do
  echo -n "$var=$v: "
  env $var=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stream8 --extra-dtypes= "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'ms', round(d['value']), 'frames/s')"
done
