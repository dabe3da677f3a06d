"""MI355X-native BM25 index build and top-k scoring path (gfx950 HIP kernels behind a C ABI).

The directory name carries a hyphen, which a module name cannot; ``search_engine_amd`` at the repository
root is a symbolic link to this directory, so ``import search_engine_amd`` loads this one package.
"""
from .native import (  # noqa: F401
    DOC_DONE, DOC_WC_NULL, FIELD_BODY, FIELD_DESCRIPTION, FIELD_TITLE, PAD_UNIT, ROWS_UNIT_MAJOR, SRC_NEWS, SRC_WEB,
    CorpusSpec, DocTotals, Index, IndexInfo, Sebm25Error, gen_queries, make_spec, meta, zipf_cdf,
)
