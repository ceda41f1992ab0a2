#!/bin/bash
# ABBA of one environment switch: isolated fire-module times (scripts/fire_times.py) and the bf16 bench step, in ONE gpurun call.
# usage: ab_env_fire.sh VAR A_VALUE B_VALUE
var=$1; a=$2; b=$3
for v in $a $b $b $a; do
  echo "== fire modules, $var=$v"
  env $var=$v python3 scripts/fire_times.py 2>&1 | grep -v amdgpu.ids | grep "64x64\|32x32\|16x16\|sum"
done
for v in $a $b $b $a; do
  env $var=$v python3 bench.py --steps 60 --warmup 10 --extra-dtypes '' --no-cpu-baseline --no-stream8 --no-cups --no-host-fed --no-probes 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$var=$v', 'ms_per_step %.3f' % d['ms_per_step'], 'patch avg us %.1f' % d['roofline']['avg_launch_us'])"
done
