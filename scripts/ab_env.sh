#!/bin/bash
# ABBA of one environment switch on the bf16 bench step in ONE gpurun call: ab_env.sh VAR A_VALUE B_VALUE [bench args]
var=$1; a=$2; b=$3; shift 3
for v in $a $b $b $a; do
  env $var=$v python3 bench.py --steps 60 --warmup 10 --extra-dtypes '' --no-cpu-baseline --no-stream8 "$@" 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$var=$v', 'ms_per_step %.3f' % d['ms_per_step'], 'patch avg us %.1f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'])"
done
