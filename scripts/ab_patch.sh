#!/bin/bash
# ABBA of two builds (lib/libokp_hip_A.so, _B.so) on the patch-kernel shapes of the step
for a in "hw=64" "hw=128" "hw=128 cin=128 stride=2" "hw=64 stride=2" "hw=64 res=1"; do
  for v in A B B A; do
    echo "$v [$a] $(OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so python3 scripts/probe_patch.py $a 2>&1 | grep 'tile 13' | tr '\n' ' ')"
  done
done
for a in "hw=32" "hw=16"; do
  for v in A B B A; do
    echo "$v [unpool $a] $(OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so python3 scripts/probe_unpool.py $a 2>&1 | tail -1)"
  done
done
