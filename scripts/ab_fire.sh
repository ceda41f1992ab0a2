#!/bin/bash
# ABBA of two builds (lib/libokp_hip_A.so, _B.so) on every one-launch fire module shape of the step (scripts/fire_times.py)
for v in A B B A; do
  echo "== $v"; OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so python3 scripts/fire_times.py 2>&1 | grep -v amdgpu.ids
done
