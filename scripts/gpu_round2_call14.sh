#!/bin/bash
# GPU call 14 of round 2: wave-owned BM25 kernel, waves per query sweep (re-used for call 15: sentinel accumulators, both kernels)
O=gpurun_out; mkdir -p $O
for nb in 4096 2048 1024; do
COS_BM25_BLOCKS=$nb timeout 400 python scripts/bench_c5.py > $O/r2_c14_c5_wave_$nb.json 2> $O/r2_c14_c5_wave_$nb.err
python -c "
import json,sys;d=json.load(open('$O/r2_c14_c5_wave_$nb.json'));print('wave $nb',{k:d[k] for k in ('bm25_stream_ms_per_batch_hip_events','bm25_frac_of_hbm_8TBps','hybrid_one_call_ms_per_batch','parity_vs_oracle')})"
done
