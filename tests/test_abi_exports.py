"""The C-ABI library loads and exports every function include/sebm25.h declares.  No GPU and
no compute calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    text = open(os.path.join(ROOT, "include", "sebm25.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sebm25_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from search_engine_amd import native

    names = declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(native.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(native.EXPORTS) == names      # the Python binding covers the whole header


def test_host_only_entry_points_work_without_a_gpu():
    import numpy as np

    from oracle import oracle as O
    from search_engine_amd import native

    np.testing.assert_array_equal(native.zipf_cdf(3000), O.zipf_cdf(3000))
    np.testing.assert_array_equal(native.gen_queries(7, 3000, 20, 3, 100), O.gen_queries(7, 3000, O.zipf_cdf(3000), 100, 20, 3))


def test_no_gpu_fails_loudly():
    import pytest
    import torch

    from search_engine_amd import native

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(native.Sebm25Error) as e:
        native.Index(0)
    assert e.value.code == 3      # SEBM25_E_HIP: no fallback path exists


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "search-engine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
