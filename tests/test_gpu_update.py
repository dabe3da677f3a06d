"""Incremental update against a full rebuild: after every batch of crawler-style changes (delete-then-insert of a
document's rows, app/crawler/worker.py:229-239; removed documents; new documents, articles and terms; status flips)
the updated index must equal, bit for bit, an index built from the final table — df, row counts, N, avgdl, idf,
doc lengths, the postings, and the top-k of both searches — and both must equal the CPU oracle on that table."""
import os

import numpy as np
import pytest

from oracle import oracle as O

import search_engine_amd as se
from helpers import assert_build_matches

pytestmark = pytest.mark.gpu

WEB, NEWS = 1, 2


class Table:
    """The `tokens` table (+ the two `documents` columns the job reads) held per unit on the host."""

    def __init__(self, c):
        self.n_terms = c["n_terms"]
        self.docs, self.arts = {}, {}
        src = c["meta"] & 3
        for arr, s, store in ((None, WEB, self.docs), (None, NEWS, self.arts)):
            m = src == s
            units = c["unit_id"][m]
            order = np.argsort(units, kind="stable")
            u_s, t_s, f_s, m_s = units[order], c["term_id"][m][order], c["tf"][m][order], c["meta"][m][order]
            cuts = np.flatnonzero(np.diff(u_s)) + 1
            for u, t, f, mm in zip(np.split(u_s, cuts), np.split(t_s, cuts), np.split(f_s, cuts), np.split(m_s, cuts)):
                if len(u):
                    store[int(u[0])] = (t.copy(), f.copy(), mm.copy())
        self.n_docs, self.n_articles = c["n_docs"], c["n_articles"]
        self.word_count = c["word_count"].astype(np.uint32).copy()
        self.flags = np.ones(self.n_docs, np.uint8)

    def arrays(self, docs=None, arts=None):
        """Row arrays of the given units (default: the whole table), web docs first, then articles."""
        parts = []
        for store, keys in ((self.docs, docs), (self.arts, arts)):
            for u in sorted(store if keys is None else keys):
                if u in store:
                    t, f, m = store[u]
                    parts.append((t, np.full(len(t), u, np.uint32), f, m))
        if not parts:
            z = np.zeros(0, np.uint32)
            return z, z.copy(), z.copy(), np.zeros(0, np.uint8)
        return tuple(np.ascontiguousarray(np.concatenate([p[i] for p in parts])) for i in range(4))


def random_doc_rows(rng, n_terms, hot):
    """Rows of one web doc: a few title / description / body terms (some shared between fields)."""
    body = np.unique(np.concatenate([rng.integers(0, n_terms, rng.integers(5, 40)), rng.choice(hot, 3)]))
    title = rng.choice(body, min(3, len(body)), replace=False)
    desc = rng.choice(body, min(5, len(body)), replace=False)
    t = np.concatenate([title, desc, body]).astype(np.uint32)
    f = np.concatenate([np.full(len(title), 1), np.full(len(desc), 2), np.full(len(body), 4)])
    tf = rng.integers(1, 6, len(t)).astype(np.uint32)
    return t, tf, (WEB | (f << 2)).astype(np.uint8), int(tf[f == 4].sum())


def apply_random_batch(rng, tb, hot):
    """Mutates the table like a crawl cycle; -> (changed docs, changed articles)."""
    cd, ca = set(), set()
    # new vocabulary now and then
    if rng.random() < 0.7:
        tb.n_terms += int(rng.integers(1, 50))
    # new docs and articles at the end of the id spaces
    n_new = int(rng.integers(0, 6))
    if n_new:
        tb.word_count = np.concatenate([tb.word_count, np.zeros(n_new, np.uint32)])
        tb.flags = np.concatenate([tb.flags, np.ones(n_new, np.uint8)])
        for d in range(tb.n_docs, tb.n_docs + n_new):
            cd.add(d)
        tb.n_docs += n_new
    n_new_a = int(rng.integers(0, 3))
    for a in range(tb.n_articles, tb.n_articles + n_new_a):
        ca.add(a)
    tb.n_articles += n_new_a
    # re-crawled docs (some of them changed in an earlier batch already), removed docs, status flips
    for d in rng.integers(0, tb.n_docs, int(rng.integers(1, 30))).tolist():
        cd.add(int(d))
    for d in sorted(cd):
        r = rng.random()
        if r < 0.15:                                   # the page is gone: no rows, no longer 'done'
            tb.docs.pop(d, None)
            tb.flags[d], tb.word_count[d] = 0, 0
        else:
            t, tf, m, wc = random_doc_rows(rng, tb.n_terms, hot)
            tb.docs[d] = (t, tf, m)
            tb.word_count[d] = wc
            tb.flags[d] = 1 if r < 0.9 else (0 if r < 0.95 else 3)     # not done / done with NULL word_count
    for a in rng.integers(0, max(tb.n_articles, 1), int(rng.integers(0, 6))).tolist():
        if tb.n_articles:
            ca.add(int(a))
    for a in sorted(ca):
        if rng.random() < 0.2:
            tb.arts.pop(a, None)
        else:
            t = np.unique(np.concatenate([rng.integers(0, tb.n_terms, rng.integers(3, 25)), rng.choice(hot, 2)])).astype(np.uint32)
            tb.arts[a] = (t, rng.integers(1, 5, len(t)).astype(np.uint32), np.full(len(t), NEWS | (4 << 2), np.uint8))
    return sorted(cd), sorted(ca)


def full_build(ix, tb):
    t, u, f, m = tb.arrays()
    ix.build(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, unit_major=True)


def assert_same_index(a, b, tb, rng, hot, k_list=(10, 210)):
    sa, sb = a.stats(), b.stats()
    for key in ("df", "term_rows", "idf"):
        np.testing.assert_array_equal(sa[key], sb[key], err_msg=key)          # idf too: same kernel, same integers in
    assert sa["n_docs"] == sb["n_docs"] and sa["avgdl"] == sb["avgdl"]
    np.testing.assert_array_equal(a.doc_len(), b.doc_len())
    for x, y in zip(a.postings(), b.postings()):
        np.testing.assert_array_equal(x, y)
    ia, ib = a.info(), b.info()
    assert (ia.n_rows, ia.n_postings, ia.n_web_postings) == (ib.n_rows, ib.n_postings, ib.n_web_postings)
    queries = [np.unique(np.concatenate([rng.choice(hot, rng.integers(1, 3)), rng.integers(0, tb.n_terms, rng.integers(0, 3))])).astype(np.uint32)
               for _ in range(40)]
    queries.append(np.array([tb.n_terms - 1], np.uint32))
    for source in (WEB, NEWS):
        for k in k_list:
            oa, ob = a.search(queries, k, source=source), b.search(queries, k, source=source)
            for key in ("count", "unit", "score", "matched"):
                np.testing.assert_array_equal(oa[key], ob[key], err_msg=f"{key} source {source} k {k}")
    return queries


def test_update_batches_equal_full_rebuild(gpu_index_factory):
    rng = np.random.default_rng(20260301)
    c = O.gen_corpus(O.make_spec(n_terms=3000, n_docs=1200, n_articles=150))
    tb = Table(c)
    hot = np.arange(20, 60)
    upd, ref = gpu_index_factory(), gpu_index_factory()
    full_build(upd, tb)
    for batch in range(6):
        cd, ca = apply_random_batch(rng, tb, hot)
        t, u, f, m = tb.arrays(cd, ca)
        upd.update(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=cd, changed_articles=ca)
        assert upd.info().n_segments == 2          # the base and one update segment, however many updates
        full_build(ref, tb)
        queries = assert_same_index(upd, ref, tb, rng, hot)
    # the CPU oracle on the final table agrees with both
    t, u, f, m = tb.arrays()
    orc = O.OracleIndex(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags)
    assert_build_matches(upd, orc)
    upd.set_idf(orc.idf)
    out = upd.search(queries, 10)
    for qi, q in enumerate(queries):
        ru, rs, rm = orc.search(q, 10, level="csr")
        n = int(out["count"][qi])
        assert n == len(ru)
        np.testing.assert_array_equal(out["unit"][qi, :n], ru)
        np.testing.assert_array_equal(out["score"][qi, :n], rs)
        np.testing.assert_array_equal(out["matched"][qi, :n], rm)
    # compaction: one segment again, nothing changes
    upd.finalize()
    dead = upd.info().n_dead_rows
    assert dead > 0
    upd.compact()
    inf = upd.info()
    assert inf.n_segments == 1 and inf.n_dead_rows == 0
    full_build(ref, tb)
    assert_same_index(upd, ref, tb, rng, hot)
    # and it can be updated again afterwards
    cd, ca = apply_random_batch(rng, tb, hot)
    t, u, f, m = tb.arrays(cd, ca)
    upd.update(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=cd, changed_articles=ca)
    full_build(ref, tb)
    assert_same_index(upd, ref, tb, rng, hot)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SEBM25_FUZZ_UPDATE_SEEDS", "2"))))   # more for a longer sweep
def test_update_and_compaction_cycles_reuse_segment_storage(gpu_index_factory, seed):
    """Several cycles of updates and compactions on one handle: the storage of dropped segments is used again, a
    compaction may find nothing but dead rows to drop (units removed, none added), one or many update segments to
    merge, grown term / doc id spaces (the rows are repacked); after each the index equals a full rebuild."""
    rng = np.random.default_rng(77 + seed)
    c = O.gen_corpus(O.make_spec(seed=20260227 + seed, n_terms=2500, n_docs=5000, n_articles=300))   # several merge tiles of base rows
    tb = Table(c)
    hot = np.arange(10, 50)
    upd, ref = gpu_index_factory(), gpu_index_factory()
    full_build(upd, tb)
    for cycle, n_batches in enumerate((1, 3, 0, 5, 2) if seed == 0 else rng.integers(0, 5, 5).tolist()):
        if n_batches == 0:                              # removal only: docs and articles disappear, no rows arrive
            cd = sorted(set(rng.integers(0, tb.n_docs, 40).tolist()))
            ca = sorted(set(rng.integers(0, tb.n_articles, 10).tolist()))
            for d in cd:
                tb.docs.pop(d, None)
                tb.flags[d], tb.word_count[d] = 0, 0
            for a in ca:
                tb.arts.pop(a, None)
            z = np.zeros(0, np.uint32)
            upd.update(z, z, z, np.zeros(0, np.uint8), tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags,
                       changed_docs=cd, changed_articles=ca)
        for _ in range(n_batches):
            cd, ca = apply_random_batch(rng, tb, hot)
            t, u, f, m = tb.arrays(cd, ca)
            upd.update(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=cd, changed_articles=ca)
        upd.compact()
        inf = upd.info()
        assert inf.n_segments == 1 and inf.n_dead_rows == 0, cycle
        full_build(ref, tb)
        assert_same_index(upd, ref, tb, rng, hot, k_list=(10,))


def test_update_rejects_rows_of_units_that_are_not_listed(gpu_index_factory):
    c = O.gen_corpus(O.make_spec(n_terms=500, n_docs=100))
    tb = Table(c)
    ix, ref = gpu_index_factory(), gpu_index_factory()
    full_build(ix, tb)
    t, u, f, m = tb.arrays([3, 4], [])
    with pytest.raises(se.Sebm25Error) as e:
        ix.update(t, u, f, m, tb.n_terms, tb.n_docs, 0, tb.word_count, tb.flags, changed_docs=[3])
    assert e.value.code == 1
    with pytest.raises(se.Sebm25Error):
        ix.update(t, u, f, m, tb.n_terms - 1, tb.n_docs, 0, tb.word_count, tb.flags, changed_docs=[3, 4])     # id spaces only grow
    full_build(ref, tb)                     # the failed calls left the index as it was
    rng = np.random.default_rng(1)
    assert ix.info().n_segments == 1
    assert_same_index(ix, ref, tb, rng, np.arange(5, 30), k_list=(10,))


def test_update_of_a_larger_index(gpu_index_factory):
    """300k rows in the base, three updates of a few hundred docs each: equal to the rebuild, and the update
    itself only reads the changed rows from the caller."""
    rng = np.random.default_rng(7)
    c = O.gen_corpus(O.make_spec(n_terms=20000, n_docs=4000, n_articles=300))
    tb = Table(c)
    hot = np.arange(50, 150)
    upd, ref = gpu_index_factory(), gpu_index_factory()
    full_build(upd, tb)
    for _ in range(3):
        cds, cas = set(), set()
        for _ in range(8):
            cd, ca = apply_random_batch(rng, tb, hot)
            cds.update(cd)
            cas.update(ca)
        cd, ca = sorted(cds), sorted(cas)
        t, u, f, m = tb.arrays(cd, ca)
        upd.update(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=cd, changed_articles=ca)
    full_build(ref, tb)
    assert_same_index(upd, ref, tb, rng, hot)
    upd.compact()
    assert_same_index(upd, ref, tb, rng, hot)


def test_sharded_update_equals_unsharded_rebuild(gpu_index_factory):
    """Two document shards on one GPU, each updated with sebm25_update_local; df and the totals summed as the
    all-reduce would; both searches merged across the shards equal an unsharded index built from the final table."""
    n_docs, n_art, V, G = 1600, 120, 4000, 2
    rng = np.random.default_rng(11)
    hot = np.arange(10, 50)
    final_t, final_u, final_f, final_m, final_wc, final_fl = [], [], [], [], [], []
    shards, dfs, tots, bases = [], [], [], []
    for g in range(G):
        d0, d1 = n_docs * g // G, n_docs * (g + 1) // G
        a0, a1 = n_art * g // G, n_art * (g + 1) // G
        tb = Table(O.gen_corpus(O.make_spec(n_terms=V, n_docs=d1 - d0, n_articles=a1 - a0, doc_begin=d0, article_begin=a0)))
        ix = gpu_index_factory()
        t, u, f, m = tb.arrays()
        ix.build_local(t, u, f, m, V, tb.n_docs, tb.n_articles, tb.word_count, tb.flags)
        ix.set_unit_base(d0, a0)
        # a crawl cycle inside the shard: docs re-crawled or removed, articles replaced (id spaces fixed so that the
        # global numbering of the other shard stays put)
        cd = sorted(set(rng.integers(0, tb.n_docs, 40).tolist()))
        ca = sorted(set(rng.integers(0, tb.n_articles, 8).tolist()))
        for d in cd:
            if rng.random() < 0.2:
                tb.docs.pop(d, None)
                tb.flags[d], tb.word_count[d] = 0, 0
            else:
                tt, tf, mm, wc = random_doc_rows(rng, V, hot)
                tb.docs[d], tb.word_count[d] = (tt, tf, mm), wc
        for a in ca:
            tt = np.unique(np.concatenate([rng.integers(0, V, 12), rng.choice(hot, 2)])).astype(np.uint32)
            tb.arts[a] = (tt, rng.integers(1, 5, len(tt)).astype(np.uint32), np.full(len(tt), NEWS | (4 << 2), np.uint8))
        t, u, f, m = tb.arrays(cd, ca)
        ix.update(t, u, f, m, V, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=cd, changed_articles=ca, local=True)
        df, tt_ = ix.local_stats()
        shards.append(ix)
        dfs.append(df.astype(np.uint64))
        tots.append(tt_)
        t, u, f, m = tb.arrays()
        web = (m & 3) == WEB
        final_t.append(t); final_f.append(f); final_m.append(m)
        final_u.append(np.where(web, u + d0, u + a0).astype(np.uint32))
        final_wc.append(tb.word_count); final_fl.append(tb.flags)
    df_global = sum(dfs).astype(np.uint32)
    tg = se.DocTotals(sum(t.n_done for t in tots), sum(t.n_wc for t in tots), sum(t.sum_wc for t in tots), sum(t.n_postings for t in tots))
    rows = [np.concatenate(x) for x in (final_t, final_u, final_f, final_m)]
    order = np.lexsort((rows[1], rows[3] & 3))
    orc = O.OracleIndex(*[x[order] for x in rows], V, n_docs, n_art, np.concatenate(final_wc), np.concatenate(final_fl))
    np.testing.assert_array_equal(df_global, orc.df)
    for ix in shards:
        ix.finalize(df_global, tg)
        st = ix.stats()
        np.testing.assert_array_equal(st["df"], orc.df)
        assert st["n_docs"] == orc.n_docs_stat and st["avgdl"] == orc.avgdl
        ix.set_idf(orc.idf)
    qs = [np.unique(np.concatenate([rng.choice(hot, 2), rng.integers(0, V, 1)])).astype(np.uint32) for _ in range(30)]
    for source in (WEB, NEWS):
        k = 15
        parts = [ix.search(qs, k, source=source, want_tie=True) for ix in shards]
        cat = {key: np.ascontiguousarray(np.stack([p[key] for p in parts])) for key in ("unit", "score", "matched", "tie", "count")}
        merged = shards[0].merge_topk(G, len(qs), k, cat["unit"], cat["score"], cat["matched"], cat["tie"], cat["count"])
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, k, source=source, level="csr")
            n = int(merged["count"][qi])
            assert n == len(ru)
            np.testing.assert_array_equal(merged["unit"][qi, :n], ru)
            np.testing.assert_array_equal(merged["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(merged["matched"][qi, :n], rm)


def test_failed_update_leaves_the_index_as_it_was(gpu_index_factory):
    """An update that cannot be applied — here: id spaces grown so far that the packed row has too few bits left for the
    frequencies the update segment already holds (SEBM25_E_RANGE) — is refused before a row of the index changes:
    statistics, postings and searches are those of before, the handle is still finalized, and the next valid update
    works."""
    rng = np.random.default_rng(7)
    c = O.gen_corpus(O.make_spec(n_terms=2000, n_docs=300, n_articles=20))
    tb = Table(c)
    hot = np.arange(10, 40)
    upd, ref = gpu_index_factory(), gpu_index_factory()
    full_build(upd, tb)
    cd, ca = apply_random_batch(rng, tb, hot)          # a first update: the index now holds an update segment
    t, u, f, m = tb.arrays(cd, ca)
    upd.update(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=cd, changed_articles=ca)
    full_build(ref, tb)
    assert_same_index(upd, ref, tb, rng, hot)
    before = upd.info()
    # 2^21 terms and 2^21 docs leave 61 - 21 - 21 = 19 bits for tf; the update segment was packed with 41
    big = 1 << 21
    wc = np.zeros(big, np.uint32)
    wc[:tb.n_docs] = tb.word_count
    fl = np.zeros(big, np.uint8)
    fl[:tb.n_docs] = tb.flags
    rows = tb.arrays([0], [])
    with pytest.raises(se.Sebm25Error) as e:
        upd.update(*rows, big, big, tb.n_articles, wc, fl, changed_docs=[0])
    assert e.value.code == 2
    upd.n_terms, upd.n_docs, upd.n_articles = tb.n_terms, tb.n_docs, tb.n_articles     # the sizes the index still has
    after = upd.info()
    assert (after.n_segments, after.n_rows, after.n_postings, after.n_dead_rows, after.n_terms, after.n_docs) == \
           (before.n_segments, before.n_rows, before.n_postings, before.n_dead_rows, before.n_terms, before.n_docs)
    assert_same_index(upd, ref, tb, rng, hot)          # searches run: the handle is still finalized
    # a row of a unit that is not listed: refused as well, same state
    with pytest.raises(se.Sebm25Error) as e:
        upd.update(*tb.arrays([1, 2], []), tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=[1])
    assert e.value.code == 1
    assert_same_index(upd, ref, tb, rng, hot)
    # and the next valid update is applied on top of the unchanged state
    cd, ca = apply_random_batch(rng, tb, hot)
    t, u, f, m = tb.arrays(cd, ca)
    upd.update(t, u, f, m, tb.n_terms, tb.n_docs, tb.n_articles, tb.word_count, tb.flags, changed_docs=cd, changed_articles=ca)
    full_build(ref, tb)
    assert_same_index(upd, ref, tb, rng, hot)
