#!/usr/bin/env python3
"""Probe of the FP4 exhaustive scan with two waves per SIMD (tuning knob flat_fp4_w8; kernels_scan.hip flat_scan_q2_fp4_w8) against
the one-wave-per-SIMD kernel (flat_fp4_w8 = 0) on config c3's shape (n x 768 quaternary codes, 256-query batch): scan-kernel time inside the
call (HIP events), wall time per call, and bit-for-bit equality of ids / score bits / counts.  One JSON line per variant."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    import torch
    import cosdata_amd as ca
    from cosdata_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(7)
    n, d, B = a.n, a.dim, a.batch
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
    ix = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, ca.StorageType.SubByte(2), (-1.0, 1.0), device=0)
    ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
    Qh = Q.cpu().numpy()
    ref = None
    variants = (("one_wave_per_simd", {"flat_fp4_w8": 0}), ("w8", {"flat_fp4_w8": 1}), ("w8_even_deal", {"flat_fp4_w8": 2}), ("w8_even_deal_alternating", {"flat_fp4_w8": 3}),
                ("w8_even_deal_alternating_128col", {"flat_fp4_w8": 4}))
    for name, knobs in (variants + tuple((n_ + "_again", k_) for n_, k_ in variants)):
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
        print(json.dumps({"variant": name, "n": n, "dim": d, "batch": B, "gemm_ms_all_launches": gm, "gemm_ms_min": float(min(r[0] for r in runs)),
                          "ms_per_call": wl, "gemm_launches": int(st.gemm_launches), "pops": st.int8_ops / gm / 1e12, "frac_of_10PF": st.int8_ops / gm / 1e12 / 10.0,
                          "same_answer_as_first_variant": same}), flush=True)


if __name__ == "__main__":
    main()
