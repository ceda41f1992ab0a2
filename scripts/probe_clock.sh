#!/bin/bash
# sample sclk / power with rocm-smi while one convolution shape runs in a loop (is the L kernel power-limited?)
tile=${1:-6}
python scripts/bench_conv.py hw=128 tile=$tile iters=4000 > /tmp/bc.log 2>&1 &
pid=$!
sleep 6
for i in 1 2 3; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.7
done
wait $pid
tail -1 /tmp/bc.log | sed -E 's/.*: ([0-9.]+ us\/launch.*)/\1/'
