"""Search entry points with the reference's signatures, scoring on the GPU index.

Drop-in for app/api/search_service.py:149-367: `perform_web_search(*, q, limit=20, offset=0)`
and `perform_news_search(...)` return the same pydantic models; the candidate limit, the
intent re-rank and the paging follow the reference line for line in meaning.  The one thing
that changes is where candidate rows come from: instead of `cur.execute(SEARCH_SQL, ...)`
(search_service.py:319-320, :351-352) the rows come from `SearchEngine.score_web/score_news`,
in the same tuple layout and order.

A process configures the engine once (`configure(engine)`); the engine builds its index on first use
and can follow the statistics job (`engine.start_auto_refresh`).  Requests that arrive together are
scored as one batch (`_MicroBatcher`); the GPU call runs in a worker thread, so the event loop that
serves other requests is never blocked (ctypes releases the GIL).
"""
from __future__ import annotations

import asyncio
import functools
import math
import re
from typing import Any

from pydantic import BaseModel

from ..tokenizer import STOPWORDS, TOKEN_RE, tokenize

CANDIDATE_BUFFER = 200     # search_service.py:103
MAX_CANDIDATES = 2000      # search_service.py:104

_NON_ALNUM = re.compile(r"[^a-z0-9]+")

# additive bonuses of the intent score (search_service.py:187-220)
_W_LOG, _W_COVERAGE, _W_ALL_TERMS = 12.0, 25.0, 40.0
_PHRASE_BONUS = (("title", 140.0), ("url", 70.0), ("description", 25.0))
_COMPACT_URL_PHRASE = 90.0
_HIT_WEIGHT = (("title", 22.0, 80.0), ("url", 16.0, 55.0), ("compact_url", 12.0, 45.0))   # per hit, all-words bonus
_NEWS_BONUS = 8.0          # search_service.py:297


class NewsFeed(BaseModel):
    feed_url: str | None = None
    home_url: str | None = None
    name: str | None = None
    link: str | None = None
    image: str | None = None
    discovered_by_url: str | None = None
    last_published: str | None = None
    last_fetched: str | None = None
    next_fetch_at: str | None = None
    publish_rate_per_hour: float | None = None


class WebSearchItem(BaseModel):
    title: str
    description: str
    url: str
    score: float


class NewsSearchItem(BaseModel):
    title: str
    description: str
    url: str
    score: float
    feed: NewsFeed | None = None
    author: str | None = None
    published_at: str | None = None


class WebSearchResponse(BaseModel):
    results: list[WebSearchItem]
    count: int


class NewsSearchResponse(BaseModel):
    results: list[NewsSearchItem]
    count: int


def _iso(v):
    if not v:
        return None
    return v.isoformat() if hasattr(v, "isoformat") else str(v)


class _MicroBatcher:
    """Coalesces the scoring calls of coroutines that run at the same time into one batched GPU search
    (and one metadata statement) per kind.  The reference opens a connection and runs one statement per
    request (search_service.py:316-320); here a request waits at most `max_wait_s` for company."""

    def __init__(self, service, max_wait_s: float = 0.001, max_batch: int = 256):
        self._service = service
        self.max_wait_s = max_wait_s
        self.max_batch = max_batch
        self._loop = None
        self._pending: list = []
        self._task = None
        self.batches = 0          # batched engine calls issued (for tests and metrics)

    async def submit(self, kind: str, terms, limit: int):
        loop = asyncio.get_running_loop()
        if loop is not self._loop:          # a new event loop (e.g. a new asyncio.run): nothing carries over
            self._loop, self._pending, self._task = loop, [], None
        fut = loop.create_future()
        self._pending.append((kind, list(terms), int(limit), fut))
        if self._task is None or self._task.done():
            self._task = loop.create_task(self._drain())
        return await fut

    async def _drain(self):
        await asyncio.sleep(self.max_wait_s)
        while self._pending:
            batch, self._pending = self._pending[:self.max_batch], self._pending[self.max_batch:]
            engine = self._service._require_engine()
            try:
                self.batches += 1
                results = await asyncio.to_thread(engine.score_batch, [(k, t, l) for k, t, l, _ in batch])
            except Exception as e:  # noqa: BLE001 - every waiting request sees the failure, as it would its own
                for *_, fut in batch:
                    if not fut.done():
                        fut.set_exception(e)
            else:
                for (*_, fut), res in zip(batch, results):
                    if fut.done():
                        continue
                    if isinstance(res, Exception):    # this request alone cannot be scored
                        fut.set_exception(res)
                    else:
                        fut.set_result(res)


@functools.lru_cache(maxsize=1 << 18)
def _text_features(title: str, description: str, url: str):
    """What the intent score reads from a result's text, which is the same for every query that returns the
    document: the normalised title / url / description, the url without separators, and the token sets of title
    and url.  Cached per document: the regular expressions behind them were most of a request's time."""
    url_l = (url or "").lower()
    return (_NON_ALNUM.sub(" ", (title or "").lower()).strip(), _NON_ALNUM.sub(" ", url_l).strip(),
            _NON_ALNUM.sub(" ", (description or "").lower()).strip(), _NON_ALNUM.sub("", url_l),
            frozenset(TOKEN_RE.findall((title or "").lower())), frozenset(TOKEN_RE.findall(url_l)))


class SearchService:
    def __init__(self, engine=None):
        self.engine = engine
        self.batcher = _MicroBatcher(self)

    # ---- text helpers (search_service.py:150-172)
    def _normalize_text(self, text: str) -> str:
        return _NON_ALNUM.sub(" ", (text or "").lower()).strip()

    def _extract_query_words(self, text: str) -> list[str]:
        words: dict[str, None] = {}
        for w in TOKEN_RE.findall(text.lower()):
            if w not in STOPWORDS:
                words.setdefault(w)
        return list(words)

    def _count_hits(self, text: str, query_words: list[str]) -> int:
        if not query_words:
            return 0
        present = set(TOKEN_RE.findall((text or "").lower()))
        return sum(w in present for w in query_words)

    def _compact_word_hits(self, compact_text: str, query_words: list[str]) -> int:
        if not query_words or not compact_text:
            return 0
        return sum(w in compact_text for w in query_words)

    # ---- intent score (search_service.py:174-222)
    def _intent_score(self, *, token_score: float, matched_terms: int, total_terms: int, query_phrase: str,
                      query_compact: str, query_words: list[str], title: str, description: str, url: str) -> float:
        score = math.log1p(max(float(token_score), 0.0)) * _W_LOG
        if total_terms:
            score += (matched_terms / total_terms) * _W_COVERAGE
            if matched_terms == total_terms:
                score += _W_ALL_TERMS
        n_title, n_url, n_desc, compact_url, title_tokens, url_tokens = _text_features(title, description, url)
        normalized = {"title": n_title, "url": n_url, "description": n_desc}
        if query_phrase:
            for field, bonus in _PHRASE_BONUS:
                if query_phrase in normalized[field]:
                    score += bonus
        if query_compact and query_compact in compact_url:
            score += _COMPACT_URL_PHRASE
        hits = {"title": sum(w in title_tokens for w in query_words), "url": sum(w in url_tokens for w in query_words),
                "compact_url": self._compact_word_hits(compact_url, query_words)}
        for field, per_hit, _ in _HIT_WEIGHT:
            score += hits[field] * per_hit
        if query_words:
            for field, _, all_bonus in _HIT_WEIGHT:
                if hits[field] == len(query_words):
                    score += all_bonus
        return score

    # ---- query context (search_service.py:224-240)
    def _search_context(self, q: str, limit: int, offset: int) -> dict[str, Any] | None:
        query_terms = list(tokenize(q).keys())
        if not query_terms:
            return None
        query_words = self._extract_query_words(q)
        return {
            "query_terms": query_terms,
            "query_phrase": self._normalize_text(q),
            "query_compact": "".join(query_words),
            "query_words": query_words,
            "total_terms": len(query_terms),
            "candidate_limit": min(MAX_CANDIDATES, max(offset + limit + CANDIDATE_BUFFER, limit * 10)),
        }

    def _score_row(self, context, token_score, matched, title, description, url) -> float:
        return self._intent_score(token_score=float(token_score or 0.0), matched_terms=int(matched or 0),
                                  total_terms=context["total_terms"], query_phrase=context["query_phrase"],
                                  query_compact=context["query_compact"], query_words=context["query_words"],
                                  title=title, description=description, url=url)

    # ---- re-rank of candidate rows (search_service.py:242-309)
    def _rank_web_tuples(self, rows: list[tuple[Any, ...]], *, context: dict[str, Any]) -> list[tuple]:
        """(-score, url, title, description) of every candidate row, best first."""
        ranked = []
        for row in rows:
            title, description, url = row[0] or "", row[1] or "", row[2]
            ranked.append((-self._score_row(context, row[3], row[4], title, description, url), url, title, description))
        ranked.sort(key=lambda r: (r[0], r[1]))
        return ranked

    def _rank_web_rows(self, rows: list[tuple[Any, ...]], *, context: dict[str, Any]) -> list[WebSearchItem]:
        return [WebSearchItem(title=t, description=d, url=u, score=-s) for s, u, t, d in self._rank_web_tuples(rows, context=context)]

    def _news_feed_from_row(self, row: tuple[Any, ...]) -> NewsFeed | None:
        if not row[3]:
            return None
        return NewsFeed(feed_url=row[3], home_url=row[4], name=row[5], link=row[6], image=row[7],
                        discovered_by_url=row[8], last_published=_iso(row[9]), last_fetched=_iso(row[10]),
                        next_fetch_at=_iso(row[11]),
                        publish_rate_per_hour=float(row[12]) if row[12] is not None else None)

    def _rank_news_rows(self, rows: list[tuple[Any, ...]], *, context: dict[str, Any]) -> list[NewsSearchItem]:
        # the reference keeps the statement's order here: no re-sort after the bonus
        items = []
        for row in rows:
            title, description, url = row[0] or "", row[1] or "", row[2]
            score = self._score_row(context, row[15], row[16], title, description, url) + _NEWS_BONUS
            items.append(NewsSearchItem(title=title, description=description, url=url, score=score,
                                        feed=self._news_feed_from_row(row), author=row[13], published_at=_iso(row[14])))
        return items

    def _rank_score_only_rows(self, rows, *, context) -> list[WebSearchItem]:
        """Reduced ranking of (token_score, matched_terms) rows: what the reference does with its
        fallback statement when row text cannot be decoded (search_service.py:327-336)."""
        items = []
        for token_score, matched in rows:
            score = math.log1p(max(float(token_score or 0.0), 0.0)) * _W_LOG
            matched = int(matched or 0)
            if context["total_terms"]:
                score += (matched / context["total_terms"]) * _W_COVERAGE
                if matched == context["total_terms"]:
                    score += _W_ALL_TERMS
            items.append(WebSearchItem(title="", description="", url="", score=score))
        items.sort(key=lambda it: it.score, reverse=True)
        return items

    def _require_engine(self):
        if self.engine is None:
            raise RuntimeError("search service is not configured: call configure(engine) first")
        return self.engine

    # ---- entry points (search_service.py:311-356)
    async def perform_web_search(self, *, q: str, limit: int = 20, offset: int = 0) -> WebSearchResponse:
        context = self._search_context(q, limit, offset)
        if not context:
            return WebSearchResponse(results=[], count=0)
        self._require_engine()
        kind, rows = await self.batcher.submit("web", context["query_terms"], context["candidate_limit"])
        if kind == "rows":
            # every candidate is scored and ranked as in the reference; response models are made for the page only
            ranked = self._rank_web_tuples(rows, context=context)
            page = [WebSearchItem(title=t, description=d, url=u, score=-s) for s, u, t, d in ranked[offset: offset + limit]]
        else:   # the row text could not be decoded: score-only rows, reduced ranking (search_service.py:321-338)
            ranked = self._rank_score_only_rows(rows, context=context)
            page = ranked[offset: offset + limit]
        return WebSearchResponse(results=page, count=max(len(ranked), offset + len(page)))

    async def perform_news_search(self, *, q: str, limit: int = 20, offset: int = 0) -> NewsSearchResponse:
        context = self._search_context(q, limit, offset)
        if not context:
            return NewsSearchResponse(results=[], count=0)
        self._require_engine()
        _, rows = await self.batcher.submit("news", context["query_terms"], context["candidate_limit"])
        ranked = self._rank_news_rows(rows, context=context)
        page = ranked[offset: offset + limit]
        return NewsSearchResponse(results=page, count=max(len(ranked), offset + len(page)))


search_service = SearchService()


def configure(engine) -> None:
    """Binds the module-level service to a SearchEngine (once per process)."""
    search_service.engine = engine


async def perform_web_search(*, q: str, limit: int = 20, offset: int = 0) -> WebSearchResponse:
    return await search_service.perform_web_search(q=q, limit=limit, offset=offset)


async def perform_news_search(*, q: str, limit: int = 20, offset: int = 0) -> NewsSearchResponse:
    return await search_service.perform_news_search(q=q, limit=limit, offset=offset)


SearchResponse = WebSearchResponse | NewsSearchResponse
