for cfg in "k=1 cin=256 cout=128 hw=64" "k=1 cin=384 cout=192 hw=32" "k=3 cin=256 cout=256 hw=16" "k=3 cin=256 cout=256 hw=8" "k=1 cin=512 cout=256 hw=16"; do
 for t in 2 7 7 2 1 8 8 1; do echo -n "$cfg tile=$t: "; python scripts/bench_conv.py $cfg tile=$t iters=50 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+ us\/launch.*)/\1/'; done; done
