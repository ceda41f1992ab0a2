#!/bin/bash
# ABBA of two builds (lib/libokp_hip_A.so, _B.so) on the split-product patch kernel's shapes of the float32x3 step (one gpurun call)
for a in "hw=64 tiles=13" "hw=64 res=1 tiles=13" "hw=128 n=32 skip=128 tiles=13" "hw=128 n=32 stride=2 cin=128 tiles=13" "hw=64 stride=2 tiles=13"; do
  echo "== $a"
  for v in A B B A; do
    echo "$v: $(OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so python scripts/probe_patch_x3.py $a 2>&1 | grep -v amdgpu | tail -1)"
  done
done
