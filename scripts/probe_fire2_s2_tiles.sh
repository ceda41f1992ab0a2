#!/bin/bash
# The stride-2 one-launch fire modules of the bf16 step on forced interior rectangles (OKP_F2_S2_TILE="ih,iw"; '' = the launcher's choice).
# usage (on the GPU box): bash scripts/probe_fire2_s2_tiles.sh
for t in "" "2,12" "2,11" "3,8" "1,16" "2,8" "4,4" "3,6" "2,10"; do
  echo "== S2 tile '$t'"
  OKP_F2_S2_TILE=$t python3 scripts/fire_times.py 2>&1 | grep " 2   \| 2  "
done
