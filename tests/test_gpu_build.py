"""GPU parity: index build through the C ABI against the CPU oracle (bit-exact integers)."""
import numpy as np
import pytest

import search_engine_amd as se
from oracle import oracle as O

from helpers import assert_build_matches, gpu_build, oracle_index, small_corpus

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_docs,n_terms,n_articles", [(300, 2000, 0), (2000, 5000, 150), (64, 37, 5), (1, 3, 0)])
def test_build_matches_oracle(gpu_index_factory, n_docs, n_terms, n_articles):
    c = small_corpus(n_docs=n_docs, n_terms=n_terms, n_articles=n_articles)
    orc = oracle_index(c)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    assert_build_matches(ix, orc)
    info = ix.info()
    assert info.n_rows == len(c["term_id"])
    assert info.n_postings == orc.totals.n_postings


def test_build_10k_docs_two_passes(gpu_index_factory):
    # benchmark configuration 1 size: 10k docs, 500k-term vocabulary -> 19 term bits, two radix passes
    c = small_corpus(n_docs=10_000, n_terms=500_000, n_articles=500)
    orc = oracle_index(c)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    assert ix.info().sort_passes == 2
    assert_build_matches(ix, orc)


def test_unsorted_rows_full_key_sort(gpu_index_factory):
    c = small_corpus(n_docs=500, n_terms=3000, n_articles=40)
    orc = oracle_index(c)
    rng = np.random.default_rng(7)
    perm = rng.permutation(len(c["term_id"]))
    shuffled = dict(c)
    for key in ("term_id", "unit_id", "tf", "meta"):
        shuffled[key] = np.ascontiguousarray(c[key][perm])
    ix = gpu_index_factory()
    with pytest.raises(se.Sebm25Error) as e:
        gpu_build(ix, shuffled, unit_major=True)
    assert e.value.code == 6  # SEBM25_E_ORDER
    gpu_build(ix, shuffled, unit_major=False)
    # weights are exact integers (tenths), so the index does not depend on the input order at all
    assert_build_matches(ix, orc)


def test_doc_flags_and_null_word_count(gpu_index_factory):
    c = small_corpus(n_docs=400, n_terms=1500, n_articles=30)
    rng = np.random.default_rng(3)
    flags = np.where(rng.random(400) < 0.8, se.DOC_DONE, 0).astype(np.uint8)
    flags |= np.where(rng.random(400) < 0.1, se.DOC_WC_NULL, 0).astype(np.uint8)
    orc = oracle_index(c, doc_flags=flags)
    ix = gpu_index_factory()
    gpu_build(ix, c, doc_flags=flags)
    assert_build_matches(ix, orc)
    # df counts docs of any status, the index keeps only 'done' docs
    assert int(orc.df.sum()) > ix.info().n_web_postings


def test_word_count_from_rows_when_absent(gpu_index_factory):
    c = small_corpus(n_docs=200, n_terms=800)
    orc = oracle_index(c, word_count=None)
    ix = gpu_index_factory()
    gpu_build(ix, c, word_count=None)
    assert_build_matches(ix, orc)
    np.testing.assert_array_equal(ix.doc_len(), c["word_count"])  # synthetic rule: word_count = sum tf over body rows


def test_empty_and_degenerate(gpu_index_factory):
    ix = gpu_index_factory()
    z32, z8 = np.zeros(0, np.uint32), np.zeros(0, np.uint8)
    ix.build(z32, z32, z32, z8, n_terms=5, n_docs=0, n_articles=0)
    st = ix.stats()
    assert st["n_docs"] == 1 and st["avgdl"] == 0.0           # `or 1`, `or 0.0` (bm25_stats.py:10,13)
    np.testing.assert_array_equal(st["df"], np.zeros(5, np.uint32))
    np.testing.assert_allclose(st["idf"], np.log(2 * 1 + 2), rtol=1e-15)  # df = 0: ln(2N + 2)
    out = ix.search([[0, 1]], k=5)
    assert out["count"][0] == 0


def test_range_errors(gpu_index_factory):
    ix = gpu_index_factory()
    one = np.ones(1, np.uint32)
    m = np.array([se.meta(1, 4)], np.uint8)
    with pytest.raises(se.Sebm25Error) as e:
        ix.build(np.array([9], np.uint32), np.zeros(1, np.uint32), one, m, n_terms=5, n_docs=3)
    assert e.value.code == 2
    with pytest.raises(se.Sebm25Error) as e:
        ix.build(np.zeros(1, np.uint32), np.array([3], np.uint32), one, m, n_terms=5, n_docs=3)
    assert e.value.code == 2
    with pytest.raises(se.Sebm25Error) as e:
        ix.build(np.zeros(1, np.uint32), np.zeros(1, np.uint32), one, np.array([3], np.uint8), n_terms=5, n_docs=3)
    assert e.value.code == 2
    with pytest.raises(se.Sebm25Error) as e:
        ix.stats()
    assert e.value.code == 5


def test_device_resident_inputs(gpu_index_factory):
    import torch

    c = small_corpus(n_docs=600, n_terms=4000, n_articles=20)
    orc = oracle_index(c)
    dev = {k: torch.from_numpy(c[k].view(np.int32) if c[k].dtype == np.uint32 else c[k]).cuda()
           for k in ("term_id", "unit_id", "tf", "meta", "word_count")}
    ix = gpu_index_factory()
    ix.build(dev["term_id"], dev["unit_id"], dev["tf"], dev["meta"], c["n_terms"], c["n_docs"], c["n_articles"],
             dev["word_count"])
    assert_build_matches(ix, orc)


def test_gpu_generator_matches_cpu_generator(gpu_index_factory):
    import torch

    ix = gpu_index_factory()
    spec = se.make_spec(n_terms=5000, n_docs=700, n_articles=90)
    n = ix.gen_corpus_rows(spec)
    ospec = O.make_spec(n_terms=5000, n_docs=700, n_articles=90)
    c = O.gen_corpus(ospec)
    assert n == len(c["term_id"])
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    u = torch.empty_like(t)
    f = torch.empty_like(t)
    m = torch.empty(n, dtype=torch.uint8, device="cuda")
    wc = torch.empty(700, dtype=torch.int32, device="cuda")
    assert ix.gen_corpus(spec, t, u, f, m, wc) == n
    np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32), c["term_id"])
    np.testing.assert_array_equal(u.cpu().numpy().view(np.uint32), c["unit_id"])
    np.testing.assert_array_equal(f.cpu().numpy().view(np.uint32), c["tf"])
    np.testing.assert_array_equal(m.cpu().numpy(), c["meta"])
    np.testing.assert_array_equal(wc.cpu().numpy().view(np.uint32), c["word_count"])
    np.testing.assert_array_equal(se.zipf_cdf(5000), O.zipf_cdf(5000))
    np.testing.assert_array_equal(se.gen_queries(20260228, 5000, 50, 3, 100),
                                  O.gen_queries(20260228, 5000, O.zipf_cdf(5000), 100, 50, 3))

def test_large_host_arrays_are_staged_like_device_arrays(gpu_index_factory):
    """Host arrays of 64 MB and more are copied in by several threads, each a slice of every array:
    the index must come out the same as from the device-resident arrays."""
    import torch

    ix = gpu_index_factory()
    spec = se.make_spec(n_terms=40_000, n_docs=25_000, n_articles=500)
    n = ix.gen_corpus_rows(spec)
    assert n * 13 > 64 << 20
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    u = torch.empty_like(t)
    f = torch.empty_like(t)
    m = torch.empty(n, dtype=torch.uint8, device="cuda")
    wc = torch.empty(25_000, dtype=torch.int32, device="cuda")
    ix.gen_corpus(spec, t, u, f, m, wc)
    ix.build(t, u, f, m, 40_000, 25_000, 500, wc)
    host = [x.cpu().numpy().view(np.uint32) for x in (t, u, f)] + [m.cpu().numpy(), wc.cpu().numpy().view(np.uint32)]
    ix2 = gpu_index_factory()
    ix2.build(host[0], host[1], host[2], host[3], 40_000, 25_000, 500, host[4])
    a, b = ix.stats(), ix2.stats()
    for key in ("df", "term_rows"):
        np.testing.assert_array_equal(a[key], b[key])
    np.testing.assert_array_equal(a["idf"].view(np.uint64), b["idf"].view(np.uint64))
    np.testing.assert_array_equal(ix.doc_len(), ix2.doc_len())
    assert ix.info().n_postings == ix2.info().n_postings
    qs = se.gen_queries(5, 40_000, 200, 3, 100)
    for src in (se.SRC_WEB, se.SRC_NEWS):
        ra, rb = ix.search(qs, 10, source=src), ix2.search(qs, 10, source=src)
        np.testing.assert_array_equal(ra["unit"], rb["unit"])
        np.testing.assert_array_equal(ra["score"].view(np.uint64), rb["score"].view(np.uint64))
    ix2.close()


def test_long_runs_and_huge_tf_take_the_exact_path(gpu_index_factory):
    """More than three rows of one (term, doc) (fields the reference's writers never produce) and
    term frequencies of 2^24 and more leave the streaming kernel's fast path: same exact weights."""
    c = small_corpus(n_docs=300, n_terms=900, n_articles=20)
    term, unit, tf, meta = (c[k].copy() for k in ("term_id", "unit_id", "tf", "meta"))
    extra_t, extra_u, extra_f, extra_m = [], [], [], []
    for doc, t in ((5, 3), (5, 4), (120, 3), (299, 899)):
        for field, f in ((1, 7), (2, 5), (3, 2), (4, 11), (5, 1), (6, 9)):   # six rows of one (term, doc)
            extra_t.append(t); extra_u.append(doc); extra_f.append(f); extra_m.append(se.meta(se.SRC_WEB, field))
    for doc, t, f in ((7, 10, 1 << 24), (7, 11, (1 << 26) + 5), (200, 0, (1 << 25) + 1)):   # huge tf, single rows
        extra_t.append(t); extra_u.append(doc); extra_f.append(f); extra_m.append(se.meta(se.SRC_WEB, 4))
    term = np.concatenate([term, np.array(extra_t, np.uint32)])
    unit = np.concatenate([unit, np.array(extra_u, np.uint32)])
    tf = np.concatenate([tf, np.array(extra_f, np.uint32)])
    meta = np.concatenate([meta, np.array(extra_m, np.uint8)])
    c2 = dict(c, term_id=term, unit_id=unit, tf=tf, meta=meta)
    orc = oracle_index(c2, word_count=None)
    ix = gpu_index_factory()
    gpu_build(ix, c2, word_count=None, unit_major=False)
    assert_build_matches(ix, orc)


def _random_unit_major_rows(rng, n_rows, n_terms, n_docs, n_articles):
    """n_rows rows in unit-major order (web docs first, then news articles), several rows per unit,
    some (term, unit) pairs repeated across fields."""
    n_news = n_rows // 7 if n_articles else 0
    n_web = n_rows - n_news
    unit_web = np.sort(rng.integers(0, n_docs, n_web)).astype(np.uint32)
    unit_news = np.sort(rng.integers(0, max(n_articles, 1), n_news)).astype(np.uint32)
    # a skewed term distribution: many repeats of small ids, a long tail
    term = np.minimum((n_terms * rng.random(n_rows) ** 3).astype(np.uint32), n_terms - 1)
    tf = rng.integers(1, 40, n_rows).astype(np.uint32)
    field = rng.choice(np.array([1, 2, 4], np.uint8), n_rows)
    meta = np.concatenate([se.SRC_WEB | (field[:n_web] << 2), np.full(n_news, se.SRC_NEWS | (4 << 2), np.uint8)]).astype(np.uint8)
    return dict(term_id=term, unit_id=np.concatenate([unit_web, unit_news]), tf=tf, meta=meta, n_terms=n_terms,
                n_docs=n_docs, n_articles=n_articles, word_count=None)


@pytest.mark.parametrize("n_terms", [5, 1500, 300_000])
def test_row_counts_around_tile_chunk_and_group_edges(gpu_index_factory, n_terms):
    """Row counts just below, at and above the sizes the kernels are tiled by (64-row rounds, 512-row
    batches, 4096-row tiles, 2-tile chunks, 16-chunk groups), with one, two and two-wide sort passes."""
    rng = np.random.default_rng(n_terms)
    ix = gpu_index_factory()
    for n_rows in (1, 2, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 8191, 8192, 8193, 12289, 65535, 65536, 65537,
                   131071, 131073, 200_003):
        c = _random_unit_major_rows(rng, n_rows, n_terms, n_docs=max(3, n_rows // 9), n_articles=max(2, n_rows // 50))
        orc = oracle_index(c, word_count=None)
        gpu_build(ix, c, word_count=None)
        assert_build_matches(ix, orc)
        orc.close()


def test_mid_size_with_news_and_unfinished_docs(gpu_index_factory):
    """50k docs + 5k news articles, a fifth of the docs not 'done', some word counts NULL: the largest
    case the CPU oracle still answers in seconds; build and both kinds of search bit for bit."""
    from oracle import oracle as O

    c = small_corpus(n_docs=50_000, n_terms=100_000, n_articles=5_000)
    rng = np.random.default_rng(5)
    flags = np.where(rng.random(50_000) < 0.8, se.DOC_DONE, 0).astype(np.uint8)
    flags |= np.where(rng.random(50_000) < 0.05, se.DOC_WC_NULL, 0).astype(np.uint8)
    orc = oracle_index(c, doc_flags=flags)
    ix = gpu_index_factory()
    gpu_build(ix, c, doc_flags=flags)
    assert_build_matches(ix, orc)
    ix.set_idf(orc.idf)  # same idf bits on both sides
    qs = O.gen_queries(99, c["n_terms"], O.zipf_cdf(c["n_terms"]), 1, 60, 3)
    for source in (se.SRC_WEB, se.SRC_NEWS):
        out = ix.search(qs, 25, source=source)
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, 25, source=source, level="csr")
            n = int(out["count"][qi])
            assert n == len(ru)
            np.testing.assert_array_equal(out["unit"][qi, :n], ru)
            np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(out["matched"][qi, :n], rm)
    orc.close()
