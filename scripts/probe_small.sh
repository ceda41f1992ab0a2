for cfg in "k=1 cin=512 cout=256 hw=4" "k=1 cin=128 cout=64 hw=4" "k=1 cin=128 cout=64 hw=2" "k=3 cin=256 cout=256 hw=4" "k=1 cin=384 cout=192 hw=8"; do
 for t in 8 1 2; do echo -n "$cfg tile=$t: "; python scripts/bench_conv.py $cfg tile=$t iters=200 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+ us\/launch.*)/\1/'; done; done
