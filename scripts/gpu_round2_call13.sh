#!/bin/bash
# GPU call 13 of round 2: wave-owned BM25 kernel (no workgroup barrier, FT-granular directory) against the block kernel, same run
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py -m gpu -q --timeout 600 --tb=short > $O/r2_c13_pytest.log 2>&1; tail -5 $O/r2_c13_pytest.log
for kern in wave block; do
  COS_BM25_KERNEL=$kern timeout 400 python scripts/bench_c5.py > $O/r2_c13_c5_$kern.json 2> $O/r2_c13_c5_$kern.err; tail -1 $O/r2_c13_c5_$kern.err
  python -c "
import json,sys;d=json.load(open('$O/r2_c13_c5_$kern.json'));print('$kern',{k:d[k] for k in ('bm25_stream_ms_per_batch_hip_events','bm25_frac_of_hbm_8TBps','hybrid_one_call_ms_per_batch','hybrid_ms_per_batch','parity_vs_oracle')})"
done
COS_BM25_BLOCKS=8192 timeout 400 python scripts/bench_c5.py > $O/r2_c13_c5_wave_8192.json 2> $O/r2_c13_c5_wave_8192.err
python -c "
import json,sys;d=json.load(open('$O/r2_c13_c5_wave_8192.json'));print('wave 8192',{k:d[k] for k in ('bm25_stream_ms_per_batch_hip_events','bm25_frac_of_hbm_8TBps','hybrid_one_call_ms_per_batch','parity_vs_oracle')})"
