#!/bin/bash
# GPU call 20 of round 2: flat_gemm_f32 with the LDS-staged, estimate-screened fused epilogue (parity + kernel time under rocprofv3)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flat.py -m gpu -q --timeout 600 --tb=short > $O/r2_c20_pytest.log 2>&1; tail -4 $O/r2_c20_pytest.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_fg -o fg -- python $R/scripts/bench_flat_gemm.py > $O/r2_c20_flat_gemm.json 2> $O/r2_c20_flat_gemm.err
python $R/scripts/rocprof_summary.py /tmp/p_fg/fg_results.db > $O/r2_c20_flat_gemm_kernel_trace.txt
cat $O/r2_c20_flat_gemm.json; grep "flat_" $O/r2_c20_flat_gemm_kernel_trace.txt | head
