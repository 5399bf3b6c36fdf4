#!/bin/bash
# GPU call 12 of round 2: BM25 launch shape (1-D grid, heaviest queries first, blocks per launch 2048 / 8192 / 32768)
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py -m gpu -q --timeout 600 --tb=short > $O/r2_c12_pytest.log 2>&1; tail -5 $O/r2_c12_pytest.log
for nb in 2048 8192 32768; do
  COS_BM25_BLOCKS=$nb timeout 400 python scripts/bench_c5.py > $O/r2_c12_c5_blocks$nb.json 2> $O/r2_c12_c5_blocks$nb.err; tail -1 $O/r2_c12_c5_blocks$nb.err
  python -c "
import json,sys;d=json.load(open('$O/r2_c12_c5_blocks$nb.json'));print($nb,{k:d[k] for k in ('bm25_stream_ms_per_batch_hip_events','bm25_frac_of_hbm_8TBps','hybrid_one_call_ms_per_batch','hybrid_ms_per_batch','parity_vs_oracle')})"
done
