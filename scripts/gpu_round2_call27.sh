#!/bin/bash
# GPU call 27 of round 2: cos_sparse_create_from_vectors (host-built CSR) against the index created from the CSR
O=gpurun_out; mkdir -p $O
timeout 200 python -m pytest tests/test_sparse.py -m gpu -q --timeout 200 --tb=short > $O/r2_c27_pytest.log 2>&1; tail -4 $O/r2_c27_pytest.log
