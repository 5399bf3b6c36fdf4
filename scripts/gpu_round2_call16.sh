#!/bin/bash
# GPU call 16 of round 2: BM25 block kernel with sentinel accumulators, blocks per launch sweep
O=gpurun_out; mkdir -p $O
for nb in 2048 4096 16384; do
  COS_BM25_KERNEL=block COS_BM25_BLOCKS=$nb timeout 400 python scripts/bench_c5.py > $O/r2_c16_c5_block_$nb.json 2> $O/r2_c16_c5_block_$nb.err
  python -c "
import json,sys;d=json.load(open('$O/r2_c16_c5_block_$nb.json'));print('block $nb',{k:d[k] for k in ('bm25_stream_ms_per_batch_hip_events','bm25_frac_of_hbm_8TBps','hybrid_one_call_ms_per_batch','parity_vs_oracle')})"
done
