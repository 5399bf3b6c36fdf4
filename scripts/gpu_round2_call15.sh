#!/bin/bash
# GPU call 15 of round 2: BM25 with sentinel accumulators (no touched bitmap, no LDS atomics in the scoring loop): both kernels, parity
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py -m gpu -q --timeout 600 --tb=short > $O/r2_c15_pytest.log 2>&1; tail -5 $O/r2_c15_pytest.log
for cfg in "wave 32768" "wave 8192" "block 32768" "block 8192"; do
  set -- $cfg
  COS_BM25_KERNEL=$1 COS_BM25_BLOCKS=$2 timeout 400 python scripts/bench_c5.py > $O/r2_c15_c5_$1_$2.json 2> $O/r2_c15_c5_$1_$2.err
  python -c "
import json,sys;d=json.load(open('$O/r2_c15_c5_$1_$2.json'));print('$1 $2',{k:d[k] for k in ('bm25_stream_ms_per_batch_hip_events','bm25_frac_of_hbm_8TBps','hybrid_one_call_ms_per_batch','parity_vs_oracle')})"
done
