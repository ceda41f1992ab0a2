#!/bin/bash
# usage: run_spawn_ws1.sh <tag>   (on the GPU box)  ->  gpurun_out/<tag>/bench_spawn_ws1.{json,err}
# The one-GPU rehearsal of the driver's multi-GPU command: `python bench.py --gpus 1 --spawn` goes through the launcher (no HIP
# call in the parent), torchrun starts ONE rank, the rank creates its RCCL process group (backend "nccl") and runs the
# barrier / all_gather_into_tensor / all_reduce(MAX) of the N > 1 path on a one-rank communicator.  NCCL_DEBUG=VERSION puts the
# RCCL version line into the log.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
NCCL_DEBUG=VERSION timeout 600 python3 bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline --no-stream8 --extra-dtypes= > $out/bench_spawn_ws1.json 2> $out/bench_spawn_ws1.err
echo "rc=$?" | tee -a $out/bench_spawn_ws1.err
cut -c1-400 $out/bench_spawn_ws1.json
grep -i -m3 "rccl\|nccl version" $out/bench_spawn_ws1.err
tail -5 $out/bench_spawn_ws1.err
