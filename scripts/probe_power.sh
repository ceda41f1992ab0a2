#!/bin/bash
# Board power and shader clock while the bf16 step runs back to back (is the step power-limited?): rocm-smi samples next to a long bench run.
python3 bench.py --steps ${1:-3000} --warmup 20 --extra-dtypes '' --no-cpu-baseline --no-stream8 > /tmp/power_bench.json 2>/tmp/power_bench.err &
pid=$!
while kill -0 $pid 2>/dev/null; do
  echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E -i "Package Power|sclk" | sed -E 's/.*: //' | tr '\n' ' ')"
  sleep 0.25
done > /tmp/power_samples.txt
wait $pid
sort -k4 -n -r -t' ' /tmp/power_samples.txt | head -0
awk '{gsub(/[()A-Za-z]/,"",$2); print $1, $2, $NF}' /tmp/power_samples.txt | sort -k3 -n | tail -12
python3 -c "import json;d=json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],'patch avg us',d['roofline']['avg_launch_us'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i power
