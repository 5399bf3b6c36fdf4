"""text -> BM25 terms, host code of libcosdata_hip: XXH32 against the `xxhash` package, and tokenizer /
stopwords / max-token-length / counting / stored-tf arithmetic against a Python restatement of indexes/tf_idf/mod.rs:282-389.
The stemmer is a callback (cos_stem_english or the host's own): tests/test_stem_english.py covers the built-in one."""
import numpy as np
import pytest

STOPWORDS = {"a", "and", "are", "as", "at", "be", "but", "by", "for", "if", "in", "into", "is", "it", "no", "not", "of", "on", "or", "s", "such",
             "t", "that", "the", "their", "then", "there", "these", "they", "this", "to", "was", "will", "with", "www"}


def _tokenize(text):
    out, cur = [], ""
    for ch in text:
        if ch.isalnum() or ch == "_":
            cur += ch
        elif cur:
            out.append(cur)
            cur = ""
    if cur:
        out.append(cur)
    return out


def _py_process(text, max_token_len, avg, k1, b, stem=lambda s: s):
    import xxhash
    kept = [t.lower() for t in _tokenize(text) if len(t.encode()) <= max_token_len]
    kept = [t for t in kept if t not in STOPWORDS]
    freq = {}
    for t in kept:
        h = xxhash.xxh32(stem(t).encode(), seed=0).intdigest()
        freq[h] = freq.get(h, 0) + 1
    f = np.float32
    out = {}
    for h, c in freq.items():
        inner = f(f(f(1.0) - f(b)) + f(f(b) * f(f(len(kept)) / f(avg))))
        out[h] = f(f(f(c) * f(f(k1) + f(1.0))) / f(f(c) + f(f(k1) * inner)))
    return len(kept), out


TEXTS = [
    "The quick brown fox jumps over the lazy dog; the dog was NOT amused, and the fox_42 ran to www.example.com!",
    "",
    "   ...   ",
    "a the and of",                                                                         # only stopwords
    "Supercalifragilisticexpialidocious_and_then_some_more_characters_to_exceed_forty short ok",   # a token above max_token_len
    "naïve café Ünïcode straße ÉCOLE 123abc x_y_z __ 9",
    "repeat repeat Repeat REPEAT other Other",
    "tabs\tand\nnewlines\r\nmixed,punctuation;everywhere:ok?yes!(no)[maybe]{x}",
]


def test_xxhash32_matches_the_xxhash_package():
    import xxhash
    from cosdata_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1)
    for n in list(range(0, 40)) + [63, 64, 65, 255, 1000]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 0xDEADBEEF):
            assert L.cos_xxhash32(data, n, seed) == xxhash.xxh32(data, seed=seed).intdigest(), (n, seed)


@pytest.mark.parametrize("text", TEXTS)
def test_process_text_matches_python_restatement(text):
    import cosdata_amd as ca
    for max_len, avg, k1, b in ((40, 1.0, 1.5, 0.75), (5, 17.5, 1.2, 0.5), (40, 120.0, 2.0, 0.0)):
        n_tokens, want = _py_process(text, max_len, avg, k1, b)
        assert ca.count_tokens(text, max_len) == n_tokens
        hashes, tfs = ca.process_text(text, max_len, avg, k1, b, stemmer=None)
        assert hashes.tolist() == sorted(want)                                   # ascending hash
        assert all(np.float32(tfs[i]).tobytes() == np.float32(want[int(h)]).tobytes() for i, h in enumerate(hashes))


def test_stemmer_callback_is_applied_before_hashing():
    import cosdata_amd as ca
    text = "running runs runner ran quickly"
    stem = lambda s: s[:3]                                                        # a stand-in, NOT the reference's stemmer
    _, want = _py_process(text, 40, 3.0, 1.5, 0.75, stem)
    hashes, tfs = ca.process_text(text, 40, 3.0, 1.5, 0.75, stemmer=stem)
    assert hashes.tolist() == sorted(want) and len(hashes) == 3                  # run x3, ran, qui
    assert all(np.float32(tfs[i]).tobytes() == np.float32(want[int(h)]).tobytes() for i, h in enumerate(hashes))


def test_stored_tf_equals_the_oracle_formula():
    from cosdata_amd import _lib
    from oracle import oracle as O
    L = _lib.lib()
    for c, dl, avg, k1, b in ((1, 10, 12.5, 1.5, 0.75), (7, 300, 120.0, 1.2, 0.6), (3, 1, 1.0, 2.0, 1.0)):
        assert np.float32(L.cos_bm25_term_frequency(c, dl, avg, k1, b)).tobytes() == np.float32(O.bm25_tf(c, dl, avg, k1, b)).tobytes()


def test_text_side_survives_arbitrary_bytes_and_tiny_buffers():
    """cos_text_process / cos_text_count_tokens / cos_stem_english on 4000 seeded inputs the Rust host's `&str` could never hold
    (invalid UTF-8, truncated sequences) and the ones it can (multi-byte case mappings, apostrophes, tokens of hundreds of bytes),
    with output capacities of 0..1024 terms and stem buffers of 0..64 bytes: COS_OK with ascending hashes within the capacity,
    or COS_ERR_INVALID — no overrun of the stem buffer (guard bytes), no crash"""
    import ctypes as C
    from cosdata_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    stem = C.cast(L.cos_stem_english, C.c_void_p)
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'  .,-yYsSeEdDgGiInN", np.uint8)
    points = [0x41, 0x61, 0x20, 0xE9, 0x130, 0x1E9E, 0x10400, 0x3A3, 0xDF, 0x27]
    n_ok = n_invalid = 0
    for _ in range(4000):
        kind, n = int(rng.integers(4)), int(rng.integers(0, 400))
        if kind == 0:
            raw = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 1:
            raw = bytes(rng.choice(letters, n))
        elif kind == 2:
            raw = "".join(chr(int(c)) for c in rng.choice(points, n)).encode()
        else:
            raw = bytes(rng.choice(letters, n)) * int(rng.integers(1, 4)) + bytes(rng.integers(128, 256, int(rng.integers(0, 5)), dtype=np.uint8))
        cap, max_len = int(rng.choice([0, 1, 2, 8, 1024])), int(rng.choice([0, 1, 5, 40, 4000]))
        h, t, out = np.zeros(max(cap, 1), np.uint32), np.zeros(max(cap, 1), np.float32), C.c_uint32()
        rc = L.cos_text_process(raw, len(raw), max_len, 1.0, 1.5, 0.75, stem if rng.integers(2) else None, None,
                                h.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), cap, C.byref(out))
        if rc == 0:
            n_ok += 1
            assert out.value <= cap and np.all(np.diff(h[:out.value].astype(np.int64)) > 0)
        else:
            n_invalid += 1
            assert rc == _lib.ERR_INVALID
        L.cos_text_count_tokens(raw, len(raw), max_len)
        tok, oc = raw[:int(rng.integers(0, 60))], int(rng.choice([0, 1, 3, 64]))
        buf = C.create_string_buffer(oc + 8)
        buf.raw = b"\xAA" * (oc + 8)
        L.cos_stem_english(None, tok, len(tok), buf, oc)
        assert buf.raw[oc:] == b"\xAA" * 8, (tok, oc)
    assert n_ok > 500 and n_invalid > 500
