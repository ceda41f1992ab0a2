#!/bin/bash
# Several builds (lib/libokp_hip_<tag>.so) in ONE gpurun call, palindromic order: isolated fire-module times only (scripts/fire_times.py).
# usage: ab_fire_only.sh A B ...
L=$GRAFT_REPO_ROOT/object_keypoints_amd/lib
order="$@"; rev=$(echo $order | tr ' ' '\n' | tac | tr '\n' ' ')
for v in $order $rev; do
  echo "== fire modules, build $v"
  OKP_LIB=$L/libokp_hip_$v.so python3 scripts/fire_times.py 2>&1 | grep -v amdgpu.ids | grep "64x64\|32x32\|16x16\|sum"
done
