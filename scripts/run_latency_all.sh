for d in bf16 f32mix f32x3 f32; do timeout 300 python scripts/latency_sweep.py dtype=$d batches=1,2,4,8,16,32,64 ; done > gpurun_out/r03r_latency.txt 2>&1
