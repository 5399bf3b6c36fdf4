#!/bin/bash
# GPU call 3 of round 2: fused flat-scan epilogue (tests + c3), on-disk reader test, default bench with PB8 + walk chain.
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_flat.py tests/test_ref_index_format.py -m gpu -q --timeout 900 > $O/r2_c3_pytest.log 2>&1; tail -15 $O/r2_c3_pytest.log
python scripts/bench_c3.py --walk-n 0 > $O/r2_c3_c3_fused.json 2> $O/r2_c3_c3_fused.err; tail -2 $O/r2_c3_c3_fused.err; cat $O/r2_c3_c3_fused.json
COS_FLAT_UNFUSED=1 python scripts/bench_c3.py --walk-n 0 --reps 2 > $O/r2_c3_c3_unfused.json 2> $O/r2_c3_c3_unfused.err; cat $O/r2_c3_c3_unfused.json
python bench.py --steps 20 --warmup 5 > $O/r2_c3_bench_default.json 2> $O/r2_c3_bench_default.err; tail -2 $O/r2_c3_bench_default.err; head -c 300 $O/r2_c3_bench_default.json; echo
python - <<'P'
import json
b=json.load(open('gpurun_out/r2_c3_bench_default.json'))
print({k:b[k] for k in ['value','ms_per_step','recall_at_10','single_batch_qps','build_seconds']}, b['roofline']['frac'], b['roofline']['aggregate'], b['roofline']['per_launch'])
P
