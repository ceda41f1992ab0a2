#!/bin/bash
# Several builds (lib/libokp_hip_<tag>.so) in ONE gpurun call, palindromic order (clock drift cancels): isolated fire-module times
# (scripts/fire_times.py) and the bf16 bench step.  usage: ab_libs.sh A B C D
L=$GRAFT_REPO_ROOT/object_keypoints_amd/lib
order="$@"; rev=$(echo $order | tr ' ' '\n' | tac | tr '\n' ' ')
for v in $order $rev; do
  echo "== fire modules, build $v"
  OKP_LIB=$L/libokp_hip_$v.so python3 scripts/fire_times.py 2>&1 | grep -v amdgpu.ids | grep "64x64\|32x32\|16x16\|sum"
done
for v in $order $rev; do
  OKP_LIB=$L/libokp_hip_$v.so python3 bench.py --steps 60 --warmup 10 --extra-dtypes '' --no-cpu-baseline --no-stream8 --no-cups --no-host-fed --no-probes 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('build $v', 'ms_per_step %.3f' % d['ms_per_step'], 'patch avg us %.1f' % d['roofline']['avg_launch_us'])"
done
