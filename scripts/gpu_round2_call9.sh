#!/bin/bash
# GPU call 9 of round 2: query-resident quaternary scan (parity + c3 timing against the tile kernel)
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flat.py tests/test_gpu_hybrid.py -m gpu -q --timeout 600 -x > $O/r2_c9_pytest.log 2>&1; tail -8 $O/r2_c9_pytest.log
timeout 300 python scripts/bench_c3.py --walk-n 0 > $O/r2_c9_c3_areg.json 2> $O/r2_c9_c3_areg.err; tail -1 $O/r2_c9_c3_areg.err; cat $O/r2_c9_c3_areg.json
COS_FLAT_TILE_KERNEL=1 timeout 300 python scripts/bench_c3.py --walk-n 0 > $O/r2_c9_c3_tile.json 2> $O/r2_c9_c3_tile.err; cat $O/r2_c9_c3_tile.json
