#!/bin/bash
# timing decomposition: DBG 1 = no epilogue, 2 = epilogue VALU without the survivor path, 3 = no sched_group_barrier, 4 = no sgb in expansion steps
O=gpurun_out; mkdir -p $O
for d in 0 1 2 3 4; do
  if [ $d = 0 ]; then unset COS_FLAT_DBG; else export COS_FLAT_DBG=$d; fi
  timeout 300 python scripts/bench_c3.py --walk-n 0 --reps 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('DBG $d gemm_ms', round(j['flat']['gemm_ms'],3), 'wall', round(j['flat']['wall_s_incl_select_rerank_copies']*1e3,3))"
done | tee $O/r2_c12_dbg.txt
