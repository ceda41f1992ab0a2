#!/bin/bash
# usage: ab_tiles.sh "<bench_conv args>" t1 t2 ...   -> one line per tile code, run in the order given
cfg=$1; shift
for t in "$@"; do echo -n "$cfg tile=$t: "; python scripts/bench_conv.py $cfg tile=$t 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+ us\/launch.*)/\1/'; done
