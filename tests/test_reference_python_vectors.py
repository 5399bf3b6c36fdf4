"""Known answers produced by the REFERENCE'S OWN Python restatement of the BM25 text side (tests/golden/reference_python_bm25_text.json,
made by tests/golden/make_reference_python_vectors.py from the pure functions of /root/reference/tests/test-tf-idf-bm25.py — the script the
reference's authors check the server's TF-IDF index with).  The oracle's BM25 formulas and the library's host-side `process_text` /
`count_tokens` / `compute_bm25_term_frequency` (text_terms.hip) must reproduce them: hashes, counts and lengths exactly,
f32 values within 2 ulp of the reference's f64 value rounded to f32 (the Rust path computes in f32: tolerance stated here, nowhere else).
The stemmer stays outside (py_rust_stemmers is not installed: the vectors were made with an identity stemmer and `stemmer=None`)."""
import json
import os

import numpy as np

from oracle import oracle as O

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_python_bm25_text.json")))


def _ulps(a, b):
    a, b = np.float32(a), np.float32(b)
    return abs(int(a.view(np.int32)) - int(b.view(np.int32)))


def test_oracle_bm25_formulas_match_the_references_python():
    for e in FIX["idf"]:            # get_idf (sparse_ann_query.rs:298-302): ln_1p((N - n + 0.5) / (n + 0.5)) in f32
        assert _ulps(O.bm25_idf(e["documents"], e["containing"]), e["idf_f64"]) <= 2, e
    for e in FIX["term_frequency"]:  # compute_bm25_term_frequency (indexes/tf_idf/mod.rs:362-371)
        assert _ulps(O.bm25_tf(e["count"], e["document_length"], e["average_document_length"], e["k1"], e["b"]), e["tf_f64"]) <= 2, e


def test_library_text_side_matches_the_references_python():
    """cos_text_process with no stemmer == transform_sentence_to_vector with an identity stemmer: the same term hashes (xxhash32 of the
    lowercased token), the same counts behind the same stored term frequencies, the same document length — stopwords, the 40-byte limit,
    '_' inside tokens, digits, punctuation and empty input included"""
    from cosdata_amd import _lib
    from cosdata_amd import hybrid as HY
    L = _lib.lib()
    for e in FIX["term_frequency"]:
        assert _ulps(L.cos_bm25_term_frequency(e["count"], e["document_length"], e["average_document_length"], e["k1"], e["b"]), e["tf_f64"]) <= 2, e
    seen = 0
    for e in FIX["text"]:
        assert HY.count_tokens(e["text"]) == e["length"], e["text"]
        hashes, tfs = HY.process_text(e["text"], 40, e["average_document_length"], e["k1"], e["b"], stemmer=None)
        want = sorted(zip(e["hashes"], e["term_frequencies_f64"]))
        assert [int(h) for h in hashes] == [h for h, _ in want], e["text"]
        for got, (_, w) in zip(tfs, want):
            assert _ulps(got, w) <= 2, (e["text"], float(got), w)
        seen += len(want)
    assert seen > 300
