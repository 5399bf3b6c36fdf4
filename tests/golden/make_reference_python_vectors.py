#!/usr/bin/env python3
"""Golden vectors from the REFERENCE'S OWN Python restatement of the BM25 text side (run once in the build container; the fixture is committed,
/root/reference does not exist on the GPU box).

/root/reference/tests/test-tf-idf-bm25.py is the script the reference's authors check the server's TF-IDF index with: it mirrors
`process_text` (indexes/tf_idf/mod.rs:282-389: tokenize, lowercase, the 35 stopwords, the 40-byte token limit, xxhash32 of the stemmed
token, term counts, document length) and the BM25 formulas (`compute_bm25_term_frequency`, `get_idf`: sparse_ann_query.rs:298-302) in Python
to compute the expected answers.  Its top level needs the cosdata SDK, bm25s and py_rust_stemmers, none of which is installed here, so this
script compiles ONLY the pure functions it names out of the file's syntax tree (nothing of the file's text is kept) and runs them on inputs
of its own: the committed fixture holds those inputs and the functions' outputs.  The stemmer is the one piece that cannot be run
(py_rust_stemmers is absent): the sentences are passed through with an identity stemmer, which pins everything around it —
`cos_text_process(..., stemmer = NULL)` must reproduce the hashes, counts, lengths and stored term frequencies."""
import ast
import json
import math
import os
import re
import sys
import unicodedata
from collections import defaultdict
from typing import Dict, List, Set, Tuple

import xxhash

SRC = "/root/reference/tests/test-tf-idf-bm25.py"
WANT = {"compute_bm25_idf", "compute_bm25_term_frequency", "get_all_punctuation", "remove_non_alphanumeric", "SimpleTokenizer", "raw_term_frequencies",
        "hash_token", "construct_sparse_vector", "transform_sentence_to_vector"}
tree = ast.parse(open(SRC).read())
mod = ast.Module(body=[n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in WANT], type_ignores=[])
assert {n.name for n in mod.body} == WANT, sorted(WANT - {n.name for n in mod.body})
ns = {"math": math, "re": re, "sys": sys, "unicodedata": unicodedata, "xxhash": xxhash, "defaultdict": defaultdict, "Dict": Dict, "List": List, "Set": Set, "Tuple": Tuple}
exec(compile(mod, SRC, "exec"), ns)


class IdentityStemmer:
    def stem_word(self, w):
        return w


SENTENCES = [
    "The quick brown fox jumps over the lazy dog",
    "a and are as at be but by for if in into is it no not of on or s such t that the their then there these they this to was will with www",
    "HNSW graphs: layered, navigable small-world graphs -- built top-down; searched top-down!",
    "vector_store.rs calls traverse_find_nearest 1112 times; ef_search=256 ef_search=256 EF_SEARCH=256",
    "supercalifragilisticexpialidociousandevenlongerthanthat fits_not but_this_one_is_exactly_forty_bytes_long__",
    "It's a dog-eat-dog world; don't panic.  www.example.com/index.html?q=1&r=2",
    "MI355X MI355X mi355x Mi355X gfx950 gfx950 CDNA4",
    # (a lone "_" is the one ASCII case where the Python mirror and the Rust differ: the mirror drops single-character tokens that are Unicode
    # punctuation — "_" is category Pc —, the Rust tokenizer keeps every run of alphanumerics and '_' (indexes/tf_idf/mod.rs:288-306);
    # the library follows the Rust, the fixture stays on common ground)
    "", "   ", "the the THE The", "__ ___ a_b A_B x_1 _lead trail_",
    "2024 2025 3.14159 1,000,000 0xFFFFFFFF 1e-6",
    "tabs\tand\nnewlines\r\nseparate tokens   just    like   spaces",
    "one", "one one one one one one one one one one",
    "x " * 50, "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho sigma tau upsilon phi chi psi omega",
]
punct = ns["get_all_punctuation"]()
text = []
for i, s in enumerate(SENTENCES):
    v = ns["transform_sentence_to_vector"](i, s, punct, IdentityStemmer())
    for avg_len, k1, b in ((1.0, 1.5, 0.75), (7.25, 1.2, 0.75), (120.0, 1.5, 0.0), (3.0, 2.0, 1.0)):
        tfs = [ns["compute_bm25_term_frequency"](c, v["length"], avg_len, k1, b) for c in v["raw_term_frequencies"]]
        text.append({"text": s, "average_document_length": avg_len, "k1": k1, "b": b, "length": v["length"], "hashes": v["indices"],
                     "counts": v["raw_term_frequencies"], "term_frequencies_f64": tfs})
idf = [{"documents": n, "containing": c, "idf_f64": ns["compute_bm25_idf"](n, c)}
       for n in (1, 2, 10, 1000, 1_000_000, 12_345_678) for c in sorted({1, 2, 3, n // 1000 + 1, n // 10 + 1, n // 2, max(1, n - 1), n}) if c <= n]
tf = [{"count": c, "document_length": dl, "average_document_length": al, "k1": k1, "b": b, "tf_f64": ns["compute_bm25_term_frequency"](c, dl, al, k1, b)}
      for c in (1, 2, 3, 7, 50, 1000) for dl in (1, 5, 120, 4000) for al in (1.0, 96.5, 120.0) for k1, b in ((1.5, 0.75), (1.2, 0.5), (2.0, 1.0), (0.9, 0.0))]
out = {"source": "functions of /root/reference/tests/test-tf-idf-bm25.py run by tests/golden/make_reference_python_vectors.py (identity stemmer)",
       "text": text, "idf": idf, "term_frequency": tf}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_python_bm25_text.json")
json.dump(out, open(path, "w"))
print(path, len(text), len(idf), len(tf))
