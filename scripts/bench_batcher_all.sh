#!/bin/bash
# serving-shaped batcher benches: fp16 10Mx768 (K2, Q<=64 per pass) and fp32 10Mx768 (K1b, 8 queries per pass)
mkdir -p gpurun_out
timeout 600 python scripts/bench_batcher.py --dtype f16 --threads 128 > gpurun_out/bench_batcher_f16.json 2> gpurun_out/bench_batcher_f16.err
timeout 600 python scripts/bench_batcher.py --dtype f32 --threads 32 --per-thread 10 --max-batch 16 > gpurun_out/bench_batcher_f32.json 2> gpurun_out/bench_batcher_f32.err
tail -c 300 gpurun_out/bench_batcher_f32.err
cat gpurun_out/bench_batcher_f32.json
