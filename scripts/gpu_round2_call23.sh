#!/bin/bash
# GPU call 23 of round 2: refreshed single-batch sweep (ef 64 / 256) and c5 after the winner pre-screen
O=gpurun_out; mkdir -p $O
SWEEP_BS=256,2048 SWEEP_LA=4 timeout 300 python scripts/latency_sweep.py > $O/r2_c23_latency_sweep.jsonl 2> $O/r2_c23_latency_sweep.err; cut -c1-200 $O/r2_c23_latency_sweep.jsonl
timeout 400 python scripts/bench_c5.py > $O/r2_c23_c5.json 2> $O/r2_c23_c5.err
python -c "
import json;d=json.load(open('$O/r2_c23_c5.json'));print({k:d[k] for k in ('bm25_stream_ms_per_batch_hip_events','bm25_frac_of_hbm_8TBps','hybrid_one_call_ms_per_batch','hybrid_ms_per_batch','parity_vs_oracle')})"
