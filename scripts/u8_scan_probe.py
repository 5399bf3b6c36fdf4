#!/usr/bin/env python3
"""Probe of the query-resident u8 exhaustive scan (flat_scan_u8_areg) against the 256 x 128 tile kernel (tuning knob flat_tile_kernel = 1) on
n x dim u8 codes with a 256-query batch: scan-kernel time inside the call (HIP events), wall per call, i8 ops / s, equality of the answers."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import cosdata_amd as ca
    from cosdata_amd import _lib
    dev = torch.device("cuda:0")
    n, d, B = a.n, a.dim, a.batch
    g = torch.Generator(device=dev); g.manual_seed(7)
    nc = max(64, n // 1000)
    centers = torch.rand(nc, d, generator=g, device=dev) * 1.6 - 0.8

    def draw(m, seed):
        gg = torch.Generator(device=dev); gg.manual_seed(seed)
        out = torch.empty(m, d, device=dev)
        for s in range(0, m, 1 << 18):
            k = min(1 << 18, m - s)
            idx = torch.randint(0, nc, (k,), generator=gg, device=dev)
            out[s:s + k] = (centers[idx] + 0.2 * torch.randn(k, d, generator=gg, device=dev)).clamp_(-0.999, 0.999)
        return out
    X = draw(n, 42); Q = draw(B, 43)
    torch.cuda.synchronize()
    ix = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), (-1.0, 1.0), device=0)
    ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
    Qh = Q.cpu().numpy()
    ref = None
    for name, knobs in (("query_resident", {}), ("tile_kernel", {"flat_tile_kernel": 1}), ("query_resident_again", {}), ("tile_kernel_again", {"flat_tile_kernel": 1})):
        with _lib.tuning(**knobs):
            ix.flat_search(Qh, 10)
            runs = []
            for _ in range(a.reps):
                t = time.time()
                ids, sc, cnt, st = ix.flat_search(Qh, 10, with_stats=True)
                runs.append((st.gemm_ms, (time.time() - t) * 1e3))
        if ref is None:
            ref = (ids.copy(), sc.copy(), cnt.copy())
        same = bool(np.array_equal(ids, ref[0]) and np.array_equal(sc.view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(cnt, ref[2]))
        gm = float(np.median([r[0] for r in runs])); wl = float(np.median([r[1] for r in runs]))
        print(json.dumps({"variant": name, "n": n, "dim": d, "batch": B, "gemm_ms_all_launches": gm, "ms_per_call": wl, "gemm_launches": int(st.gemm_launches),
                          "i8_tops": st.int8_ops / gm / 1e9, "frac_of_5PF_i8": st.int8_ops / gm / 1e9 / 5000.0, "code_GBps": st.code_bytes / gm / 1e6,
                          "same_answer_as_first": same}), flush=True)


if __name__ == "__main__":
    main()
