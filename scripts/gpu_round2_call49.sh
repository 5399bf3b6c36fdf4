#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q --timeout 1200 -x > $O/r2_c49_pytest.log 2>&1; tail -3 $O/r2_c49_pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_c49_bench.json 2> $O/r2_c49_bench.err; python -c "
import json; b=json.load(open('$O/r2_c49_bench.json')); print({k:b[k] for k in ['value','ms_per_step','recall_at_10','single_batch_qps','parity_vs_oracle','setup_seconds']}, b['roofline']['frac'], b['roofline']['per_launch']['prep_ms'], b['roofline']['per_launch']['finalize_ms'])"
python scripts/bench_c3.py --walk-n 0 --reps 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read())['flat']; print('c3', j['gemm_ms'], j['wall_s_incl_select_rerank_copies'], j['upload_quantize_s'], j['same_answer_as_tile_kernel'])"
