#!/bin/bash
# usage: kstat.sh <pattern>  -- average duration per kernel matching <pattern> in a profiled 10-step bench, builds A and B
cd /tmp; export TMPDIR=/tmp
for v in A B B A; do
  rm -rf /tmp/ks$v
  OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --extra-dtypes "" > /dev/null 2>&1
  echo "$v $(grep -h "$1" /tmp/ks$v/*/*kernel_stats.csv | cut -d, -f1-4 | tr '\n' ' ' | cut -c1-300)"
done
