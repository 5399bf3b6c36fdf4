#!/usr/bin/env python3
"""Kernel-level check of the f32 MFMA scan (cos_bruteforce_topk): B queries x n x d, agreement with a chunked torch
matmul+topk, wall time.  Run under `rocprofv3 --kernel-trace --stats` to get flat_gemm_f32's own duration
(MFMA rate = 2*B*n*d / kernel time; dense f32 matrix peak on gfx950 = 157 TFLOP/s)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cosdata_amd as ca
from bench import mixture, bruteforce_top10
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(41)
c = torch.randn(max(64, n // 1000), d, generator=g, device=dev); c /= c.norm(dim=1, keepdim=True)
X = mixture(n, d, 42, dev, c); Q = mixture(B, d, 43, dev, c)
ix = ca.HNSWIndex(d); ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
Qh = Q.cpu().numpy()
ix.bruteforce_topk(Qh[:256], 10)   # warm-up
t = time.perf_counter(); ids, sc = ix.bruteforce_topk(Qh, 10); wall = time.perf_counter() - t
ref = bruteforce_top10(X, Q, 10).cpu().numpy()
agree = float(np.mean([len(set(ids[i].tolist()) & set(ref[i].tolist())) / 10 for i in range(B)]))
print(json.dumps({"n": n, "dim": d, "queries": B, "wall_s": wall, "tflops_end_to_end": 2.0 * B * n * d / wall / 1e12,
                  "flops_per_call": 2.0 * B * n * d, "agreement_with_torch": agree}))
