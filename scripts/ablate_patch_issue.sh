for a in "hw=64" "hw=128 cin=128 stride=2" "hw=64 stride=2"; do
  for v in A NODMA CHEAPADDR B A; do
    echo "$v [$a] $(OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so python3 scripts/probe_patch.py $a 2>&1 | grep 'tile 13' | tr '\n' ' ')"
  done
done
