"""The `bm25` batch job with the reference's entry point: `async def run() -> None`
(app/batch/bm25_stats.py:6, awaited from app/batch/runner.py:52 next to two sibling jobs).

Instead of the GROUP BY statement inside PostgreSQL, the token rows are exported, the index and
its statistics are built on the GPU, and `term_statistics` is replaced in one transaction.  The
GPU work runs in a worker thread, so the shared event loop keeps serving the sibling jobs.  An
exception propagates to the runner, which logs it and retries the cycle (runner.py:67-69).
"""
from __future__ import annotations

import asyncio

_engine = None


def configure(engine) -> None:
    """Binds the job to a SearchEngine (the same object the search service uses, so a finished
    job also swaps in the new index for searches)."""
    global _engine
    _engine = engine


def _default_engine():
    from ..engine import SearchEngine
    from ..store import SqlStore

    return SearchEngine(SqlStore.from_postgres_env())


async def run() -> None:
    global _engine
    if _engine is None:
        _engine = _default_engine()      # connection parameters from POSTGRES_* (app/common/db.py:12-19)
    await asyncio.to_thread(_engine.rebuild)


async def run_incremental(doc_ids=None, article_urls=None) -> None:
    """The job for the changes since the last run: the documents (documents.id) and news articles (url) the crawler
    and the news fetcher wrote.  Same effect on term_statistics and on searches as run(), without reading the
    whole `tokens` table again.  Without arguments the changes come from the change-capture triggers
    (SqlStore.install_change_capture); with arguments, from the caller."""
    global _engine
    if _engine is None:
        _engine = _default_engine()
    if doc_ids is None and article_urls is None:
        await asyncio.to_thread(_engine.update_pending)
    else:
        await asyncio.to_thread(_engine.update, list(doc_ids or ()), list(article_urls or ()))


if __name__ == "__main__":
    asyncio.run(run())
