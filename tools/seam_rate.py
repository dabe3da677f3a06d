"""Requests per second through the reference's search seam (`perform_web_search`) with the GPU engine behind it:
a synthetic sqlite database with the reference's table shapes, `SearchEngine.rebuild()`, then waves of concurrent
coroutines (the micro-batcher turns each wave into a few `sebm25_search` calls and metadata statements).
Usage: python tools/seam_rate.py [n_docs] [concurrency] [requests]"""
import asyncio
import os
import sqlite3
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
import search_engine_amd as se  # noqa: E402
from search_engine_amd.api import search_service as S  # noqa: E402
from search_engine_amd.engine import SearchEngine  # noqa: E402
from search_engine_amd.store import SqlStore  # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
conc = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_req = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
path = "/tmp/seam_rate.db"
if os.path.exists(path):
    os.remove(path)
# term strings that the tokenizer leaves alone: letters only, no stop words, no suffix the stemmer strips
ALPHA = "bcdfghjklmnpqrtvwxz"
def word(i):
    s = ""
    i += 19 ** 3
    while i:
        s += ALPHA[i % 19]
        i //= 19
    return "q" + s + "k"
db = sqlite3.connect(path)
db.executescript("""
CREATE TABLE documents (id INTEGER PRIMARY KEY, url TEXT UNIQUE NOT NULL, title TEXT, description TEXT, content TEXT, word_count INT, status TEXT DEFAULT 'done');
CREATE TABLE news_feeds (feed_url TEXT PRIMARY KEY, home_url TEXT, name TEXT, link TEXT, image TEXT, discovered_by_url TEXT, last_published TEXT, last_fetched TEXT, next_fetch_at TEXT, publish_rate_per_hour REAL);
CREATE TABLE news_articles (url TEXT PRIMARY KEY, feed_url TEXT, title TEXT, description TEXT, image TEXT, content TEXT, author TEXT, published_at TEXT);
CREATE TABLE tokens (doc_id INTEGER, term TEXT NOT NULL, field INTEGER NOT NULL, frequency INT NOT NULL, article_url TEXT, source_type INTEGER DEFAULT 1);
CREATE TABLE term_statistics (term TEXT PRIMARY KEY, doc_frequency BIGINT NOT NULL, idf DOUBLE PRECISION NOT NULL, avg_doc_len DOUBLE PRECISION NOT NULL);
""")
V = 50_000
c = synth.token_rows(n_docs, V)
words = [word(i) for i in range(V)]
db.executemany("INSERT INTO documents(id, url, title, description, content, word_count) VALUES (?,?,?,?,?,?)",
               [(i + 1, "https://s.test/%09d" % i, "title %d" % i, "description of page %d" % i, "", int(w)) for i, w in enumerate(c["word_count"].tolist())])
db.executemany("INSERT INTO tokens VALUES (?,?,?,?,NULL,1)",
               zip((c["unit_id"].astype(np.int64) + 1).tolist(), (words[t] for t in c["term_id"].tolist()), (int(m) >> 2 for m in c["meta"].tolist()), c["tf"].tolist()))
db.execute("CREATE INDEX idx_tokens_doc_id ON tokens(doc_id)")        # 0001_initial_schema.py:73
db.execute("CREATE INDEX idx_tokens_article_url ON tokens(article_url)")  # 0007_unify_news_tokens.py:26
db.commit()
db.close()
print(f"database: {n_docs} docs, {len(c['term_id'])} token rows", flush=True)

eng = SearchEngine(SqlStore(lambda: sqlite3.connect(path)))
t0 = time.perf_counter()
eng.rebuild()
print(f"rebuild (export + build + write-back): {time.perf_counter() - t0:.2f} s", flush=True)
S.configure(eng)
qs = synth.queries(V, n_req, 3, 50)
texts = [" ".join(words[int(t)] for t in q) for q in qs]
from search_engine_amd.tokenizer import tokenize  # noqa: E402
assert list(tokenize(texts[0]).keys()) == texts[0].split(), (texts[0], tokenize(texts[0]))

async def wave(batch):
    return await asyncio.gather(*(S.perform_web_search(q=t, limit=10) for t in batch))

async def main():
    await wave(texts[:conc])                                    # warm-up
    t0 = time.perf_counter()
    n_hits = 0
    for i in range(0, n_req, conc):
        res = await wave(texts[i:i + conc])
        n_hits += sum(len(r.results) for r in res)
    dt = time.perf_counter() - t0
    print(f"perform_web_search: {n_req} requests, {conc} at a time: {n_req / dt:.0f} requests/s ({dt / (n_req / conc) * 1e3:.1f} ms per wave), "
          f"{n_hits / n_req:.1f} results per request", flush=True)
    # the engine call alone on the same queries (term ids, k = the candidate limit of a first page)
    ex = eng.export
    ids = [[ex.term_index[w] for w in t.split()] for t in texts]
    t0 = time.perf_counter()
    for i in range(0, n_req, conc):
        eng.index.search(ids[i:i + conc], 210)
    dt2 = time.perf_counter() - t0
    print(f"sebm25_search alone, same batches at k = 210: {n_req / dt2:.0f} queries/s", flush=True)

asyncio.run(main())

# the incremental path through the seam: 100 documents re-crawled (their rows read again and replaced)
rng = np.random.default_rng(3)
for rep in range(3):
    ids = (rng.choice(n_docs, 100, replace=False) + 1).tolist()
    t0 = time.perf_counter()
    eng.update(doc_ids=ids, write_back=False)
    print(f"SearchEngine.update of 100 docs (no write-back): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)

if os.environ.get("SEAM_PROFILE"):
    import cProfile
    import pstats

    cProfile.run("asyncio.run(wave(texts[:conc]))", "/tmp/seam.prof")
    pstats.Stats("/tmp/seam.prof").sort_stats("cumulative").print_stats(28)
