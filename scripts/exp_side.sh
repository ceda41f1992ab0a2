#!/bin/bash
B="python3 bench.py --no-cpu-baseline --extra-dtypes '' --steps 100 --warmup 5"
for rep in 1 2; do
for cfg in "1 1" "1 8" "2 8" "3 8" "4 8" "5 8"; do
  set -- $cfg
  r=$(OKP_SIDE_MIN_LEVEL=$1 OKP_BENCH_TIMER_EVERY=$2 python3 bench.py --no-cpu-baseline --extra-dtypes "" --steps 100 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches_timed'])")
  echo "side_min_level=$1 timer_every=$2: $r"
done; done
