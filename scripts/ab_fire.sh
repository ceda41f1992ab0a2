#!/bin/bash
# ABBA of two builds (lib/libokp_hip_A.so, _B.so) on the fire-module shapes of the step
for a in "hw=64" "hw=32" "hw=64 stride=2" "c=384 hw=16" "c=384 hw=16 stride=2" "c=512 hw=8"; do
  for v in A B B A; do
    echo "$v $(OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so python3 scripts/probe_fire2_time.py $a 2>&1 | tail -1)"
  done
done
