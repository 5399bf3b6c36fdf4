R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/final_pytest.log
bash scripts/final_bench_only.sh
