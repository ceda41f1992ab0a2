#!/bin/bash
# every streaming fire-module instance of the step, alone on the chip at N=64 (in the step they overlap with other streams: rocprof
# durations there include the contention); effective rate = (input + output bytes) / time
for args in "c=256 hw=64" "c=256 hw=64 stride=2" "c=256 hw=32" "c=384 hw=16" "c=384 hw=16 stride=2" "c=384 hw=8" "c=384 co=256 hw=16" ; do
  python scripts/probe_fire2_time.py $args 2>/dev/null | tail -1
done
