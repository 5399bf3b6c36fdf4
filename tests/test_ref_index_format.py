"""The reference's on-disk dense index: tests/ref_index_writer.py (writer, from the serializer sources) ->
libcosdata_hip's reader.  The host-only views run without a GPU; loading into a device handle is a GPU test."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.ref_index_writer import cbor_f32, write_dense_hnsw_dir


def _oracle(storage, res, n=700, dim=40, **kw):
    X = H.uniform_corpus(n, dim, seed=3) * 0.9
    hp = dict(num_layers=3, ef_construction=24, ef_search=24, neighbors_count=8, level0_neighbors_count=16)
    hp.update(kw)
    return X, H.oracle_index(X, storage, res, **hp), hp


def _read_dir(path, hp):
    from cosdata_amd import _lib
    L = _lib.lib()
    Ltop, M, M0 = hp["num_layers"], hp["neighbors_count"], hp["level0_neighbors_count"]
    counts = np.zeros(Ltop + 1, np.uint32)
    _lib.check(L.cos_reference_dir_level_counts(path.encode(), Ltop, M, M0, counts.ctypes.data_as(C.c_void_p)))
    out = []
    for l in range(Ltop + 1):
        ids = np.zeros(counts[l], np.uint32)
        nbr = np.zeros((counts[l], M0 if l == 0 else M), np.uint32)
        _lib.check(L.cos_reference_dir_read_level(path.encode(), Ltop, M, M0, l, ids.ctypes.data_as(C.c_void_p), nbr.ctypes.data_as(C.c_void_p)))
        out.append((ids, nbr))
    return out


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2), (O.STORAGE_F16, 0), (O.STORAGE_F32, 0)])
@pytest.mark.parametrize("min_size", [1 << 62, 4096])
def test_written_directory_reads_back_identically(tmp_path, storage, res, min_size):
    X, oix, hp = _oracle(storage, res)
    g = oix.export_graph()
    d = str(tmp_path / "dense_hnsw")
    write_dense_hnsw_dir(d, g, oix.codes(), oix.mags(), storage, res, X.shape[1], index_file_min_size=min_size)
    if min_size == 4096:
        assert os.path.exists(os.path.join(d, "3.index"))       # records spread over several index files
    got = _read_dir(d, hp)
    for (gi, gn), (ei, en) in zip(got, g):
        assert np.array_equal(gi, ei) and np.array_equal(gn, en)


def test_reader_rejects_damaged_directories(tmp_path):
    from cosdata_amd import _lib
    X, oix, hp = _oracle(O.STORAGE_U8, 0, n=120)
    d = str(tmp_path / "dense_hnsw")
    write_dense_hnsw_dir(d, oix.export_graph(), oix.codes(), oix.mags(), O.STORAGE_U8, 0, X.shape[1])
    counts = np.zeros(4, np.uint32)
    args = lambda M=8, M0=16, L=3: (d.encode(), L, M, M0, counts.ctypes.data_as(C.c_void_p))
    L = _lib.lib()
    assert L.cos_reference_dir_level_counts(*args()) == 0 and counts[0] == 121
    assert L.cos_reference_dir_level_counts(*args(M0=32)) == _lib.ERR_INVALID           # other hyper-parameters than the file's
    assert L.cos_reference_dir_level_counts(*args(L=2)) == _lib.ERR_INVALID             # a node above num_layers
    raw = open(os.path.join(d, "0.index"), "rb").read()
    open(os.path.join(d, "0.index"), "wb").write(raw[:len(raw) // 2])                   # truncated index file
    assert L.cos_reference_dir_level_counts(*args()) == _lib.ERR_INVALID
    open(os.path.join(d, "0.index"), "wb").write(raw)
    p = bytearray(open(os.path.join(d, "prop.data"), "rb").read())
    p[0] = 0x1F                                                                          # not a CBOR map
    open(os.path.join(d, "prop.data"), "wb").write(bytes(p))
    assert L.cos_reference_dir_level_counts(*args()) == _lib.ERR_INVALID
    assert L.cos_reference_dir_level_counts(b"/nonexistent", 3, 8, 16, counts.ctypes.data_as(C.c_void_p)) == _lib.ERR_INVALID


def _craft_first_prop(d, tail_of):
    """overwrite the FIRST prop record (the root's: id u32::MAX) in place, keeping its length: {"id": .., "value": <tail>}"""
    from tests.ref_index_writer import cbor_text, cbor_uint
    p = bytearray(open(os.path.join(d, "prop.data"), "rb").read())
    head = b"\xa2" + cbor_text("id") + cbor_uint(0xFFFFFFFF) + cbor_text("value")
    # length of the first record = offset of the second one: records are back to back and start with the map header 0xA2
    first_len = p.index(b"\xa2" + cbor_text("id") + cbor_uint(0), 1)
    tail = tail_of(first_len - len(head))
    assert len(head) + len(tail) == first_len
    p[:first_len] = head + tail
    open(os.path.join(d, "prop.data"), "wb").write(bytes(p))


def test_reader_survives_crafted_cbor_lengths_and_nesting(tmp_path):
    """array lengths from the file were trusted (resize before any element is read) and skip() recursed without a
    limit; a corrupt prop.data must come back as COS_ERR_INVALID through the C ABI, never as an exception / stack overflow"""
    from cosdata_amd import _lib
    X, oix, hp = _oracle(O.STORAGE_U8, 0, n=120)
    L = _lib.lib()
    counts = np.zeros(4, np.uint32)
    for tail_of in (lambda m: b"\x9b" + (1 << 40).to_bytes(8, "big") + b"\x00" * (m - 9),       # array claiming 2^40 elements
                    lambda m: b"\xbb" + (1 << 63).to_bytes(8, "big") + b"\x00" * (m - 9),       # map claiming 2^63 pairs
                    lambda m: b"\x81" * m,                                                        # arrays nested to the record's end
                    lambda m: b"\xc1" * m):                                                       # tags nested to the record's end
        d = str(tmp_path / f"dense_hnsw_{len(os.listdir(tmp_path))}")
        write_dense_hnsw_dir(d, oix.export_graph(), oix.codes(), oix.mags(), O.STORAGE_U8, 0, X.shape[1])
        assert L.cos_reference_dir_level_counts(d.encode(), 3, 8, 16, counts.ctypes.data_as(C.c_void_p)) == 0
        _craft_first_prop(d, tail_of)
        assert L.cos_reference_dir_level_counts(d.encode(), 3, 8, 16, counts.ctypes.data_as(C.c_void_p)) == _lib.ERR_INVALID


def test_reader_survives_random_damage(tmp_path):
    """200 seeded mutations (bytes replaced, truncation, runs of 0x00 / 0xFF, garbage appended) of one of the directory's files: the
    host-only reader answers COS_OK with a well-formed level list or COS_ERR_INVALID — never a crash (child process: a dead reader
    fails the test instead of pytest)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "tests.fuzz_ref_index_reader", str(tmp_path), "200", "7"], capture_output=True, text=True,
                         cwd=root, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stderr[-1500:])
    words = out.stdout.split()
    assert words[0] == "accepted" and int(words[1]) + int(words[3]) == 200 and int(words[1]) >= 20 and int(words[3]) >= 20      # both outcomes occur


def test_cbor_float_encoding_follows_serde_cbor():
    assert cbor_f32(1.0) == b"\xf9\x3c\x00"                 # lossless as half
    assert cbor_f32(0.1)[0] == 0xFA and len(cbor_f32(0.1)) == 5
    assert cbor_f32(np.inf) == b"\xf9\x7c\x00"
    assert cbor_f32(65504.0)[0] == 0xF9 and cbor_f32(65505.0)[0] == 0xFA


@pytest.mark.gpu
@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2), (O.STORAGE_SUBBYTE, 3), (O.STORAGE_F16, 0), (O.STORAGE_F32, 0)])
def test_loaded_directory_searches_like_the_oracle(tmp_path, storage, res):
    """write the oracle's index in the reference's format, load it into a device handle (vectors uploaded separately, stored
    codes verified against the device's own quantization) and search: ids/scores bit-exact vs the oracle"""
    import cosdata_amd as ca
    from tests.test_gpu_parity import _assert_same_search, _assert_same_walk
    X, oix, hp = _oracle(storage, res, n=2500, dim=96, num_layers=4, neighbors_count=32, level0_neighbors_count=64, ef_construction=48, ef_search=64)
    d = str(tmp_path / "dense_hnsw")
    root_ptr = write_dense_hnsw_dir(d, oix.export_graph(), oix.codes(), oix.mags(), storage, res, 96, index_file_min_size=1 << 20)
    h = ca.HNSWHyperParams(num_layers=4, ef_construction=48, ef_search=64)
    dix = ca.HNSWIndex(96, h, ca.DistanceMetric.Cosine, ca.StorageType(ca.StorageKind(storage), res), (-1.0, 1.0))
    dix.upload_vectors(X).load_reference_dir(d, root_ptr, verify_codes=True)
    for (gi, gn), (ei, en) in zip(dix.download_graph(), oix.export_graph()):
        assert np.array_equal(gi, ei) and np.array_equal(gn, en)
    Q = H.queries_from(X, 24, seed=5)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)
    # a stored code that differs from the device's quantization is reported, not ignored
    X2 = X.copy()
    X2[17] *= -1.0
    dix2 = ca.HNSWIndex(96, h, ca.DistanceMetric.Cosine, ca.StorageType(ca.StorageKind(storage), res), (-1.0, 1.0))
    with pytest.raises(ca.CosdataError) as ei:
        dix2.upload_vectors(X2).load_reference_dir(d, root_ptr, verify_codes=True)
    assert ei.value.status == 1
    with pytest.raises(ca.CosdataError):                       # a ptr offset that is not the root's
        dix2.load_reference_dir(d, root_ptr + 8)
    # a stored vector whose CBOR array claims more elements than the record holds / than any code of this index has
    from tests.ref_index_writer import cbor_map, cbor_text, cbor_f32
    variant = {0: "UnsignedByte", 1: "SubByte", 2: "HalfPrecisionFP", 3: "FullPrecisionFP"}[storage]
    key = "vec" if storage == 3 else "quant_vec"
    for claimed in (1 << 40, 96 * 4 + 1):
        def tail_of(m, claimed=claimed):
            t = b"\xa1" + cbor_text(variant) + b"\xa2" + cbor_text("mag") + cbor_f32(1.0) + cbor_text(key) + b"\x9b" + claimed.to_bytes(8, "big")
            return t + b"\x01" * (m - len(t))
        d3 = str(tmp_path / f"crafted_{claimed}")
        rp = write_dense_hnsw_dir(d3, oix.export_graph(), oix.codes(), oix.mags(), storage, res, 96, index_file_min_size=1 << 20)
        _craft_first_prop(d3, tail_of)
        with pytest.raises(ca.CosdataError) as ei:
            ca.HNSWIndex(96, h, ca.DistanceMetric.Cosine, ca.StorageType(ca.StorageKind(storage), res), (-1.0, 1.0)).upload_vectors(X).load_reference_dir(d3, rp)
        assert ei.value.status == 3


@pytest.mark.gpu
@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2), (O.STORAGE_F32, 0)])
def test_loaded_directory_takes_appends_and_deletes(tmp_path, storage, res):
    """round 6: a collection served from the Rust server's files keeps taking inserts and deletes — load the directory, restore the link
    state the reference has after a reload (cos_index_restore_link_state), append and delete: the oracle that imported the same graph and
    did the same gives the same graph, slot for slot, and the same answers"""
    import cosdata_amd as ca
    X, oix, hp = _oracle(storage, res, n=2400, dim=96, num_layers=4, neighbors_count=32, level0_neighbors_count=64, ef_construction=48, ef_search=64)
    d = str(tmp_path / "dense_hnsw")
    root_ptr = write_dense_hnsw_dir(d, oix.export_graph(), oix.codes(), oix.mags(), storage, res, 96, index_file_min_size=1 << 20)
    h = ca.HNSWHyperParams(num_layers=4, ef_construction=48, ef_search=64)
    dix = ca.HNSWIndex(96, h, ca.DistanceMetric.Cosine, ca.StorageType(ca.StorageKind(storage), res), (-1.0, 1.0), seed=oix.params.seed)
    dix.upload_vectors(X).load_reference_dir(d, root_ptr, verify_codes=True)
    dix.restore_link_state()
    oix.restore_link_state()                                     # (the oracle built it sequentially: it, too, continues from the reload state)
    Xn = H.uniform_corpus(500, 96, seed=11) * 0.9
    oix.append(Xn, 64)
    dix.append(Xn, 64)
    dele = np.arange(4, 2900, 61, dtype=np.uint32)
    oix.delete(dele)
    dix.delete(dele)
    for (gi, gn), (ei, en) in zip(dix.download_graph(), oix.export_graph()):
        assert np.array_equal(gi, ei) and np.array_equal(gn, en)
    Q = H.queries_from(np.concatenate([X, Xn]), 100, seed=6)
    ids, sc, cnt = dix.batch_search(Q, 10)
    oids, osc, ocnt = oix.search_batch(Q, 10, threads=4)[:3]
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
