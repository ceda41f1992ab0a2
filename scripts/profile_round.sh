#!/bin/bash
# usage: profile_round.sh <tag>   (on the GPU box)  -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,last_forward.txt}
# rocprofv3 --kernel-trace --stats of the default bench command; the summaries are copied into profiles/ by hand.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_profiled.json 2> $out/bench_profiled.err
cd $GRAFT_REPO_ROOT
stats=$(ls $out/raw/*/*kernel_stats.csv | head -1); trace=$(ls $out/raw/*/*kernel_trace.csv | head -1)
cp $stats $out/kernel_stats.csv
python3 scripts/trace_summary.py $trace 100 > $out/last_forward.txt
rm -rf $out/raw
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-400
tail -12 $out/last_forward.txt
