"""MI355X-native BM25 index build and top-k scoring path (gfx950 HIP kernels behind a C ABI).

The directory name carries a hyphen, so Python code imports this package through the
``search_engine_amd`` alias package at the repository root.
"""
from .native import (  # noqa: F401
    DOC_DONE, DOC_WC_NULL, FIELD_BODY, FIELD_DESCRIPTION, FIELD_TITLE, PAD_UNIT, ROWS_UNIT_MAJOR, SRC_NEWS, SRC_WEB,
    CorpusSpec, DocTotals, Index, IndexInfo, Sebm25Error, gen_queries, make_spec, meta, zipf_cdf,
)
