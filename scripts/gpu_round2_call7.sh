#!/bin/bash
# GPU call 7 of round 2: i8 scan with the register prefetch ring (PF = 1, 2, 3), BM25 with 4 postings in flight, one-call hybrid.
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_flat.py tests/test_gpu_hybrid.py tests/test_golden.py -m gpu -q --timeout 900 > $O/r2_c7_pytest.log 2>&1; tail -5 $O/r2_c7_pytest.log
for pf in 2 1 3; do
  COS_FLAT_PF=$pf python scripts/bench_c3.py --walk-n 0 > $O/r2_c7_c3_pf$pf.json 2> $O/r2_c7_c3_pf$pf.err; tail -1 $O/r2_c7_c3_pf$pf.err; cat $O/r2_c7_c3_pf$pf.json
done
python scripts/bench_c5.py > $O/r2_c7_c5.json 2> $O/r2_c7_c5.err; tail -1 $O/r2_c7_c5.err; cat $O/r2_c7_c5.json
