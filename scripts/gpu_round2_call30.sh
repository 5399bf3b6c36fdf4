#!/bin/bash
O=gpurun_out; mkdir -p $O
export COS_SCAN_WAVES4=1
python scripts/dbg_areg.py 2>&1 | tail -8 | tr "\n" ";"; echo
timeout 900 python -m pytest tests/test_gpu_flat.py -m gpu -q --timeout 600 -x 2>&1 | tail -3
timeout 300 python scripts/bench_c3.py --walk-n 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('4 waves: gemm_ms', round(j['flat']['gemm_ms'],3), 'wall', round(j['flat']['wall_s_incl_select_rerank_copies']*1e3,3))"
unset COS_SCAN_WAVES4
timeout 300 python scripts/bench_c3.py --walk-n 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('8 waves: gemm_ms', round(j['flat']['gemm_ms'],3), 'wall', round(j['flat']['wall_s_incl_select_rerank_copies']*1e3,3))"
