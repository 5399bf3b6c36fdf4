"""cosdata_amd — MI355X-native ANN query engine for cosdata's dense/hybrid search path."""
from ._lib import CosdataError, build  # noqa: F401
from .index import sample_values_range  # noqa: F401
from .index import (DistanceMetric, HNSWHyperParams, HNSWIndex, ScalarQuantization, StorageKind, StorageType,  # noqa: F401
                    ROOT_ID, QUERY_ID, SLOT_EMPTY, VISITED_REF, VISITED_EXACT)
from .hybrid import BM25Index, InvertedIndex, count_tokens, distance_batch, hybrid_search_batch, process_text, rrf_fuse_batch, sparse_build_csr, stem_english  # noqa: F401,E402
