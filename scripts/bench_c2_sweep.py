#!/usr/bin/env python3
"""Config c2 breadth: 1M x 768, storage u8 AND f32, clustered AND uniform corpora, ef sweep
{32, 64, 128, 256, 512}: recall@10 vs the exact f32 scan and QPS (8192 queries per launch, 2 launches in flight,
inputs resident in HBM).  One JSON object per (corpus, storage) on stdout.  Not the driver's bench (bench.py)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cosdata_amd as ca
from bench import mixture

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--efs", default="32,64,128,256,512")
ap.add_argument("--cases", default="clustered:u8,clustered:f32,uniform:u8")
ap.add_argument("--launches", type=int, default=24)
a = ap.parse_args()
dev = torch.device("cuda:0")
n, d, k = a.n, a.dim, 10
B = 8192
NRQ = 2048


def corpus(kind):
    if kind == "clustered":   # bench.py's corpus
        g = torch.Generator(device=dev); g.manual_seed(41)
        c = torch.randn(max(64, n // 1000), d, generator=g, device=dev)
        c = c / c.norm(dim=1, keepdim=True)
        return mixture(n, d, 42, dev, c), mixture(2 * B, d, 43, dev, c)
    g = torch.Generator(device=dev); g.manual_seed(42)   # tests/test.py:88 of the reference: uniform(-1, 1)
    X = torch.rand(n, d, generator=g, device=dev) * 2 - 1
    idx = torch.randint(0, n, (2 * B,), generator=g, device=dev)
    Q = (X[idx] + 0.05 * torch.randn(2 * B, d, generator=g, device=dev)).clamp_(-1, 1)   # rps-test.py style perturbation
    return X, Q


for case in a.cases.split(","):
    kind, storage = case.split(":")
    X, Q = corpus(kind)
    torch.cuda.synchronize()
    if storage == "u8":
        vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
        st = ca.StorageType.UnsignedByte()
    else:
        vr = (-1.0, 1.0)
        st = ca.StorageType.FullPrecisionFP()
    ix = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, st, vr, 64, device=0, seed=42)
    ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
    t0 = time.time(); ix.build(4096); build_s = time.time() - t0
    gt, _ = ix.bruteforce_topk(Q[:NRQ].cpu().numpy(), k)
    gt = torch.from_numpy(gt.astype(np.int64)).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    o_i = torch.zeros(2, B, k, dtype=torch.int32, device=dev); o_s = torch.zeros(2, B, k, dtype=torch.float32, device=dev)
    o_c = torch.zeros(2, B, dtype=torch.int32, device=dev); o_t = torch.zeros(2, B, dtype=torch.int32, device=dev)
    rows = []
    for ef in [int(x) for x in a.efs.split(",")]:
        ix.set_ef_search(ef)
        ix.batch_search_device(Q.data_ptr(), NRQ, k, o_i[0].data_ptr(), o_s[0].data_ptr(), o_c[0].data_ptr(), o_t[0].data_ptr(), streams[0].cuda_stream)
        torch.cuda.synchronize()
        ann = o_i[0][:NRQ].to(torch.int64) & 0xFFFFFFFF
        valid = torch.arange(k, device=dev)[None, :] < o_c[0][:NRQ, None]
        hit = ((ann.unsqueeze(2) == gt.unsqueeze(1)) & valid.unsqueeze(2)).any(dim=1).float().mean().item()
        def step(i):
            s = i % 2
            ix.batch_search_device(Q[(i % 2) * B:].data_ptr(), B, k, o_i[s].data_ptr(), o_s[s].data_ptr(), o_c[s].data_ptr(), o_t[s].data_ptr(),
                                   streams[s].cuda_stream)
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.launches):
            step(i)
        torch.cuda.synchronize()
        rows.append({"ef_search": ef, "recall_at_10": hit, "qps": a.launches * B / (time.perf_counter() - t1)})
    print(json.dumps({"config": f"c2 sweep: {n} x {d}, corpus {kind}, storage {storage}, values_range {vr}, default hyper-params "
                                f"(M 32 / M0 64, ef_construction 128, 9 layers), {B} queries per launch x 2 in flight",
                      "build_seconds": build_s, "sweep": rows}), flush=True)
    del ix, X, Q, gt
    torch.cuda.empty_cache()
