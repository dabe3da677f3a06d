"""Measures SqlStore.export_tokens on a synthetic sqlite database with the reference's table shapes:
rows/s of the chunked column-wise export, and of a plain per-row Python conversion of the same
statement for comparison.  Usage: python tools/export_rate.py [n_docs] [db_path]"""
import os
import sqlite3
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from search_engine_amd.store import TOKEN_COLUMNS, SqlStore  # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/export_rate.db"
if os.path.exists(path):
    os.remove(path)
db = sqlite3.connect(path)
db.executescript("""
CREATE TABLE documents (id INTEGER PRIMARY KEY, url TEXT UNIQUE NOT NULL, title TEXT, description TEXT, content TEXT, word_count INT, status TEXT DEFAULT 'done');
CREATE TABLE news_articles (url TEXT PRIMARY KEY, feed_url TEXT, title TEXT, description TEXT, image TEXT, content TEXT, author TEXT, published_at TEXT);
CREATE TABLE tokens (doc_id INTEGER, term TEXT NOT NULL, field INTEGER NOT NULL, frequency INT NOT NULL, article_url TEXT, source_type INTEGER DEFAULT 1);
CREATE TABLE term_statistics (term TEXT PRIMARY KEY, doc_frequency BIGINT NOT NULL, idf DOUBLE PRECISION NOT NULL, avg_doc_len DOUBLE PRECISION NOT NULL);
""")
t0 = time.perf_counter()
step = 20_000
total = 0
for b in range(0, n_docs, step):
    c = synth.token_rows(min(step, n_docs - b), 500_000, seed=1 + b)
    ids = (c["unit_id"].astype(np.int64) + b + 1).tolist()
    db.executemany("INSERT INTO documents(id, url, word_count) VALUES (?,?,?)",
                   [(b + i + 1, "https://s.test/%012d" % (b + i), int(w)) for i, w in enumerate(c["word_count"].tolist())])
    db.executemany("INSERT INTO tokens VALUES (?,?,?,?,NULL,1)",
                   zip(ids, ("t%06d" % t for t in c["term_id"].tolist()), (int(m) >> 2 for m in c["meta"].tolist()), c["tf"].tolist()))
    total += len(ids)
db.commit()
db.close()
print(f"database: {n_docs} docs, {total} token rows, {time.perf_counter() - t0:.1f} s to fill", flush=True)

store = SqlStore(lambda: sqlite3.connect(path))
t0 = time.perf_counter()
ex = store.export_tokens()
dt = time.perf_counter() - t0
print(f"export_tokens: {len(ex.term_id)} rows in {dt:.2f} s = {len(ex.term_id) / dt:.3e} rows/s, "
      f"{len(ex.terms)} terms, unit-major {ex.unit_major}", flush=True)

# the same statement walked row by row in Python (dictionary look-ups per row)
t0 = time.perf_counter()
db = sqlite3.connect(path)
docs = {int(r[0]): i for i, r in enumerate(db.execute("SELECT id FROM documents ORDER BY id"))}
terms = {}
n = 0
tid = np.empty(total, np.uint32); uid = np.empty(total, np.uint32); tf = np.empty(total, np.uint32); meta = np.empty(total, np.uint8)
for d, t, f, fr, art, st in db.execute(f"SELECT {TOKEN_COLUMNS} FROM tokens"):
    tid[n] = terms.setdefault(t, len(terms)); uid[n] = docs[d]; tf[n] = fr; meta[n] = st | (f << 2)
    n += 1
db.close()
dt2 = time.perf_counter() - t0
print(f"row-by-row loop: {n} rows in {dt2:.2f} s = {n / dt2:.3e} rows/s ({dt2 / dt:.1f}x slower)", flush=True)

# PostgreSQL leg without a server: the same rows as COPY text (what `COPY (SELECT ...) TO STDOUT` streams),
# parsed and assembled by the code path psycopg would feed
import pandas as pd  # noqa: E402

from search_engine_amd import store as S  # noqa: E402

frame = pd.DataFrame({0: ex.doc_ids[ex.unit_id], 1: np.array(ex.terms, dtype=object)[ex.term_id], 2: ex.meta >> 2, 3: ex.tf,
                      4: "", 5: ex.meta & 3})
blocks = [frame.iloc[i:i + (1 << 20)].to_csv(sep="\t", header=False, index=False).encode() for i in range(0, len(frame), 1 << 20)]
del frame, tid, uid, tf, meta, docs, terms          # the legs above leave tens of millions of objects behind: the
import gc                                            # collector would walk them again and again during this leg
gc.collect()
class _Stream:      # what psycopg's cursor.copy() hands over: an iterable of byte pieces
    def __enter__(self):
        return (b[i:i + (1 << 20)] for b in blocks for i in range(0, len(b), 1 << 20))

    def __exit__(self, *a):
        return False


class _Cursor:
    def copy(self, sql):
        return _Stream()


t0 = time.perf_counter()
asm = S._Assembler(ex.doc_ids, {})
for cols in SqlStore(None, "%s")._copy_chunks(_Cursor(), "SELECT ..."):
    asm.add(*cols)
ex2 = asm.finish(ex.word_count, ex.doc_flags, ex.doc_urls, [], lambda n, dt: np.empty(n, dt))
dt3 = time.perf_counter() - t0
assert np.array_equal(ex2.term_id, ex.term_id) and np.array_equal(ex2.unit_id, ex.unit_id) and np.array_equal(ex2.tf, ex.tf)
print(f"COPY text -> arrays: {len(ex2.term_id)} rows ({sum(map(len, blocks)) / 1e6:.0f} MB) in {dt3:.2f} s = {len(ex2.term_id) / dt3:.3e} rows/s "
      "(this thread plus two parser threads; excludes the server and the socket)", flush=True)
