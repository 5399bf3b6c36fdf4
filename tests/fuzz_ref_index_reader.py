"""Seeded damage to a written dense-HNSW directory, fed to the library's host-only reader (no device needed).  Run by
tests/test_ref_index_format.py in a child process, so that a crash of the reader is a failed test and not a dead pytest.
Usage: python -m tests.fuzz_ref_index_reader <scratch dir> <mutations> <seed>; prints "accepted A rejected R"."""
import ctypes as C
import os
import shutil
import sys

import numpy as np

from cosdata_amd import _lib
from oracle import oracle as O
from tests import helpers as H
from tests.ref_index_writer import write_dense_hnsw_dir

HP = dict(num_layers=3, ef_construction=24, ef_search=24, neighbors_count=8, level0_neighbors_count=16)


def damage(raw, rng):
    raw = bytearray(raw)
    kind = int(rng.integers(4))
    if kind == 0 and raw:                                   # a few bytes replaced
        for _ in range(int(rng.integers(1, 6))):
            raw[int(rng.integers(len(raw)))] = int(rng.integers(256))
    elif kind == 1:                                         # truncated
        raw = raw[:int(rng.integers(len(raw) + 1))]
    elif kind == 2 and len(raw) > 8:                        # a run of 0x00 / 0xFF (0xFF.. is the format's own "null" pattern)
        a, n = int(rng.integers(len(raw) - 4)), int(rng.integers(1, 64))
        raw[a:a + n] = bytes([int(rng.choice([0, 255]))]) * len(raw[a:a + n])
    else:                                                   # garbage appended
        raw += bytes(rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8))
    return bytes(raw)


def main(base, n_iter, seed):
    L = _lib.lib()
    X = H.uniform_corpus(300, 24, seed=3) * 0.9
    oix = H.oracle_index(X, O.STORAGE_U8, 0, **HP)
    d0 = os.path.join(base, "orig")
    write_dense_hnsw_dir(d0, oix.export_graph(), oix.codes(), oix.mags(), O.STORAGE_U8, 0, X.shape[1], index_file_min_size=4096)
    files = sorted(os.listdir(d0))
    rng = np.random.default_rng(seed)
    Ltop, M, M0 = HP["num_layers"], HP["neighbors_count"], HP["level0_neighbors_count"]
    accepted = rejected = 0
    for _ in range(n_iter):
        d = os.path.join(base, "damaged")
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(d0, d)
        f = os.path.join(d, files[int(rng.integers(len(files)))])
        raw = open(f, "rb").read()
        open(f, "wb").write(damage(raw, rng))
        counts = np.zeros(Ltop + 1, np.uint32)
        rc = L.cos_reference_dir_level_counts(d.encode(), Ltop, M, M0, counts.ctypes.data_as(C.c_void_p))
        if rc != 0:
            assert rc in (_lib.ERR_INVALID, _lib.ERR_UNIMPLEMENTED), rc
            rejected += 1
            continue
        accepted += 1                                       # damage to a distance / an id the graph rules cannot see: still a well-formed directory
        assert counts.sum() <= 4 * (len(X) + 1)
        for l in range(Ltop + 1):
            ids = np.zeros(int(counts[l]), np.uint32)
            nbr = np.zeros((int(counts[l]), M0 if l == 0 else M), np.uint32)
            assert L.cos_reference_dir_read_level(d.encode(), Ltop, M, M0, l, ids.ctypes.data_as(C.c_void_p), nbr.ctypes.data_as(C.c_void_p)) == 0
            assert np.all(np.diff(ids.astype(np.int64)) > 0) and ids[-1] == 0xFFFFFFFF      # ascending, the root last
    print("accepted", accepted, "rejected", rejected)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
