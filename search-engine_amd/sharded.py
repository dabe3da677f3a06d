"""Document-sharded index over the GPUs of one node: one process (rank) per GPU.

Every rank holds the postings of a contiguous range of documents (and of news articles) with
the full vocabulary.  The statistics job needs one exchange — per-term document frequencies
and the three document totals are sums over shards, merged with an all-reduce (RCCL over xGMI
when the process group's backend is "nccl") — after which every rank computes the same idf.
A search scores the query batch on every shard and merges the per-shard top-k lists with the
same ordering rule, so the result is identical to the one an unsharded index returns.

The reference runs this job on one coordinator against the shared database
(app/batch/runner.py:43-52); its only sharding idiom is `mod(id, nodes)`
(app/batch/duplicate_detection.py:56).

Streams.  Every call into the library blocks until its work is done, so what the library wrote may
be read by any stream.  The other direction needs care: a collective (and any torch copy made from
its result) is only ordered on torch's current stream, while the library reads its inputs on the
stream it was given with `set_stream`.  When that is torch's current stream (the default: the
library's NULL stream is torch's default stream) stream order is enough; when the caller gave the
library a stream of its own, pass it here as `lib_stream` and every hand-over from torch to the
library goes through an event that stream waits for (no host synchronisation).  gloo is a host
backend: with device buffers (rehearsals and tests of the multi-rank path on one GPU) collectives
run on host copies of the buffers; torch's gloo backend lists no all-gather / all-to-all on device
tensors, and running them there was seen to hang.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .native import SRC_NEWS, SRC_WEB, DocTotals, Sebm25Error

def _tail_words(world: int) -> int:
    """int32 words behind df in the all-reduced buffer: four totals as 16-bit limbs, the count of ranks that failed,
    and per rank its (doc base, docs, article base, articles) as 16-bit limbs (only the owner's are non-zero)."""
    return 17 + 16 * world


def _limbs(v: int) -> list[int]:
    return [(int(v) >> (16 * j)) & 0xFFFF for j in range(4)]


def _from_limbs(words) -> int:
    return sum(int(words[j]) << (16 * j) for j in range(4))


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous range [begin, end) of `n` units owned by `rank`."""
    return n * rank // world, n * (rank + 1) // world


class ShardError(RuntimeError):
    """Raised on EVERY rank when the local step of a collective operation failed on some rank (so that no rank is
    left waiting in the collective that would have followed)."""

    def __init__(self, n_failed: int, local: Exception | None):
        msg = f"{n_failed} rank(s) failed their local step"
        if local is not None:
            msg += f"; this rank: {local}"
        super().__init__(msg)
        self.n_failed = n_failed
        self.local = local
        self.code = getattr(local, "code", None)


class ShardedIndex:
    """`engine` is this rank's `native.Index` (anything with the same build_local / local_stats /
    finalize / search / merge_topk methods works); `device` is where collective buffers live.
    `partition_min_entries`: batches with at least this many result entries (queries x k) per rank are merged
    query-partitioned (each rank merges its share of the queries), smaller ones by every rank from one all-gather."""

    def __init__(self, engine, device, group=None, lib_stream=None, partition_min_entries=16384):
        self.engine = engine
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lib_stream = lib_stream
        self.partition_min_entries = partition_min_entries
        self._buf = None   # int32[V + _tail_words(world)]: df and what travels with it (one all-reduce for all)
        self._df = None
        self._via_host = (dist.is_initialized() and self.world > 1 and self.device.type == "cuda"
                          and dist.get_backend(group) == "gloo")
        self._ties = {SRC_WEB: False, SRC_NEWS: False}   # url ranks set (through set_url_rank below)
        self.doc_base = self.article_base = 0
        self.n_docs = self.n_articles = 0
        self.last_exchange_bytes = 0   # bytes this rank received in the collectives of the last search

    # ---- stream order between torch and the library
    def _hand_to_library(self):
        """What torch's current stream has queued so far (collectives included) happens before the library's next
        call.  Nothing to do when both use the same stream or the buffers are host memory."""
        if self.device.type != "cuda" or self.lib_stream is None:
            return
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.lib_stream.cuda_stream:
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        self.lib_stream.wait_event(ev)

    # ---- collectives (on host copies when gloo meets device buffers)
    def _all_reduce(self, t):
        if self._via_host:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)    # SUM over shards (RCCL over xGMI with backend nccl)

    def _all_gather(self, out, inp):
        if self._via_host:
            torch.cuda.synchronize(self.device)
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, inp.cpu(), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)
        self.last_exchange_bytes += out.numel() * out.element_size()

    def _all_to_all(self, out, inp):
        if self._via_host:
            torch.cuda.synchronize(self.device)
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h, inp.cpu(), group=self.group)
            out.copy_(h)
        else:
            dist.all_to_all_single(out, inp, group=self.group)
        self.last_exchange_bytes += out.numel() * out.element_size()

    # ---- statistics job
    @staticmethod
    def _overlap(table):
        """table: per rank (doc base, docs, article base, articles) -> a message when two ranks' global id ranges overlap."""
        for what, b, n in (("document", 0, 1), ("article", 2, 3)):
            spans = sorted((x[b], x[b] + x[n]) for x in table if x[n])
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                if a1 > b0:
                    return (f"sharded index: {what} id ranges [{a0}, {a1}) and [{b0}, {b1}) of two ranks overlap "
                            f"(a shard that grows needs headroom before the next shard's base)")
        return None

    def _check_ranges(self, n_docs, n_articles, doc_base, article_base):
        """Global ids are base + local id: the ranges of the shards must not overlap.  Checked with the sizes every rank
        is ABOUT to have, before anything changes; the verdict is the same on every rank."""
        if self.world == 1:
            return
        mine = torch.tensor([doc_base, n_docs, article_base, n_articles], dtype=torch.int64)
        every = torch.empty(self.world * 4, dtype=torch.int64)
        if dist.get_backend(self.group) == "gloo":
            dist.all_gather_into_tensor(every, mine, group=self.group)
        else:   # a few integers through device memory
            d_every = every.to(self.device)
            dist.all_gather_into_tensor(d_every, mine.to(self.device), group=self.group)
            every.copy_(d_every)
        msg = self._overlap(every.view(self.world, 4).tolist())
        if msg:
            raise ValueError(msg)

    def build(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None,
              doc_base=0, article_base=0, unit_major=True):
        """Rows of this rank's documents (unit ids local to the shard); doc_base / article_base
        are the global ids of its first document / article.  The ranks' id ranges travel with the df all-reduce and
        are checked after it (ValueError on every rank when two overlap)."""
        e = self.engine
        err = None
        try:
            e.build_local(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags,
                          unit_major=unit_major)
            e.set_unit_base(doc_base, article_base)
        except Sebm25Error as ex:   # e.g. more rows than one shard holds: every rank must learn of it
            err = ex
        self.doc_base, self.article_base, self.n_docs, self.n_articles = doc_base, article_base, n_docs, n_articles
        return self._reduce_and_finalize(n_terms, err)

    def update(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None,
               changed_docs=None, changed_articles=None, unit_major=False, doc_base=None, article_base=None):
        """Incremental update of this rank's shard (unit ids local to the shard, as in build): the listed units
        are replaced by the rows given, then df and the totals are summed over the shards again and every rank
        recomputes the same idf.  Every rank must call it (with empty arrays when its shard did not change).
        A shard may grow (n_docs / n_articles larger than before) only into ids no other shard uses: build the
        shards with headroom between their bases, or pass new bases here; an overlap raises ValueError on every
        rank before anything is changed."""
        doc_base = self.doc_base if doc_base is None else doc_base
        article_base = self.article_base if article_base is None else article_base
        self._check_ranges(n_docs, n_articles, doc_base, article_base)
        err = None
        try:
            self.engine.update(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags,
                               changed_docs=changed_docs, changed_articles=changed_articles, unit_major=unit_major, local=True)
            self.engine.set_unit_base(doc_base, article_base)
        except Sebm25Error as ex:
            err = ex
        else:
            self.doc_base, self.article_base, self.n_docs, self.n_articles = doc_base, article_base, n_docs, n_articles
        return self._reduce_and_finalize(n_terms, err)

    def _reduce_and_finalize(self, n_terms, err=None):
        e = self.engine
        tail = _tail_words(self.world)
        if self._buf is None or self._buf.numel() != n_terms + tail:
            self._buf = torch.zeros(n_terms + tail, dtype=torch.int32, device=self.device)
            self._df = self._buf[:n_terms]
        if err is None:
            _, t = e.local_stats(self._df)
        else:
            self._df.zero_()
            t = DocTotals(0, 0, 0, 0)
        if self.world > 1:
            # 64-bit values travel as four 16-bit limbs each, so that an int32 SUM over the ranks cannot overflow (up to
            # 32768 ranks) and df, totals, the count of failed ranks and the ranks' id ranges share one collective
            words = [w for v in (t.n_done, t.n_wc, t.sum_wc, t.n_postings) for w in _limbs(v)]
            words.append(1 if err is not None else 0)
            for r in range(self.world):
                mine = r == self.rank
                for v in (self.doc_base, self.n_docs, self.article_base, self.n_articles):
                    words += _limbs(v) if mine else [0, 0, 0, 0]
            self._buf[n_terms:].copy_(torch.tensor(words, dtype=torch.int32))
            self._all_reduce(self._buf)
            got = self._buf[n_terms:].tolist()
            n_failed = int(got[16])
            if n_failed:
                raise ShardError(n_failed, err)
            table = [[_from_limbs(got[17 + 16 * r + 4 * i:]) for i in range(4)] for r in range(self.world)]
            msg = self._overlap(table)
            if msg:
                raise ValueError(msg)
            t = DocTotals(*(_from_limbs(got[4 * i:]) for i in range(4)))
            self._hand_to_library()   # the summed df is read by finalize on the library's stream
        elif err is not None:
            raise err
        e.finalize(self._df, t)
        self.totals = t
        return t

    # ---- scoring
    def set_url_rank(self, source: int, rank):
        """Tie-break keys of this shard's units (global url ranks), or None for the default (the global unit id).
        Set them through this method: the exchange of a search carries tie-break keys only when ranks are set."""
        self.engine.set_url_rank(source, rank)
        self._ties[source] = rank is not None

    def _bufs(self, nq_pad, k, ties):
        """Result arrays of this rank's search over its shard, rows padded to a multiple of the world size."""
        key = (nq_pad, k, ties)
        if getattr(self, "_bufs_key", None) != key:
            dev = self.device
            b = dict(score=torch.zeros((nq_pad, k), dtype=torch.float64, device=dev),
                     unit=torch.full((nq_pad, k), -1, dtype=torch.int32, device=dev),
                     matched=torch.zeros((nq_pad, k), dtype=torch.int32, device=dev),
                     tie=torch.full((nq_pad, k), -1, dtype=torch.int32, device=dev),
                     count=torch.zeros(nq_pad, dtype=torch.int32, device=dev))
            self._loc, self._bufs_key = b, key
        return self._loc

    def search(self, queries, k, source=SRC_WEB, out_device=False):
        """-> dict(unit, score, matched, count) of the global top-k (global unit ids), the same on every
        rank: numpy arrays, or tensors on `device` with out_device=True."""
        e = self.engine
        nq, G = len(queries), self.world
        ties = self._ties[source]
        P = (nq + G - 1) // G            # queries per rank in the partitioned merge
        if nq == 0:
            z = dict(unit=torch.zeros((0, k), dtype=torch.int32), score=torch.zeros((0, k), dtype=torch.float64),
                     matched=torch.zeros((0, k), dtype=torch.int32), count=torch.zeros(0, dtype=torch.int32))
            return z if out_device else {key: _to_numpy(v) for key, v in z.items()}
        loc = self._bufs(P * G, k, ties)
        if nq < P * G:   # the padding rows read as empty lists
            loc["unit"][nq:] = -1
            loc["score"][nq:] = 0
            loc["matched"][nq:] = 0
        e.search(queries, k, source=source, out={key: v[:nq] for key, v in loc.items() if ties or key != "tie"})
        self.last_exchange_bytes = 0
        fields = ("score", "unit", "matched") + (("tie",) if ties else ())
        if G == 1:
            res = {key: v[:nq] for key, v in loc.items()}
        elif nq * k >= self.partition_min_entries:
            # Query-partitioned merge: rank r receives every shard's lists of queries [r P, (r + 1) P) (one all-to-all per
            # field; rows of one query are contiguous, so no buffer is repacked), merges them, and the merged rows of all
            # partitions are all-gathered.  Per rank and batch this moves 2 x 16 B x nq x k (20 B with url ranks) whatever
            # the number of ranks; list lengths are recomputed from the padding (unit 0xFFFFFFFF), not exchanged.
            recv = {f: torch.empty((G * P, k), dtype=loc[f].dtype, device=self.device) for f in fields}
            for f in fields:
                self._all_to_all(recv[f], loc[f])
            r_unit = recv["unit"].view(G, P, k)
            r_count = (r_unit != -1).sum(dim=2, dtype=torch.int32)
            part = dict(unit=torch.empty((P, k), dtype=torch.int32, device=self.device),
                        score=torch.empty((P, k), dtype=torch.float64, device=self.device),
                        matched=torch.empty((P, k), dtype=torch.int32, device=self.device),
                        tie=torch.empty((P, k), dtype=torch.int32, device=self.device),
                        count=torch.empty(P, dtype=torch.int32, device=self.device))
            self._hand_to_library()
            # without url ranks the tie-break key of a result is its global unit id
            e.merge_topk(G, P, k, r_unit, recv["score"].view(G, P, k), recv["matched"].view(G, P, k),
                         recv["tie"].view(G, P, k) if ties else r_unit, r_count, out=part)
            res = {}
            for f in ("score", "unit", "matched"):
                full = torch.empty((G * P, k), dtype=part[f].dtype, device=self.device)
                self._all_gather(full, part[f])
                res[f] = full[:nq]
            res["count"] = (res["unit"] != -1).sum(dim=1, dtype=torch.int32)
        else:
            # Small batches: one all-gather of one packed buffer per rank, every rank merges every query.
            n = P * G * k
            mine = torch.cat([loc[f].view(torch.uint8).reshape(-1) for f in fields])
            every = torch.empty(G * mine.numel(), dtype=torch.uint8, device=self.device)
            self._all_gather(every, mine)
            g, o = {}, 0
            for f in fields:
                nb = n * loc[f].element_size()
                g[f] = every.view(G, -1)[:, o:o + nb].contiguous().view(loc[f].dtype).view(G, P * G, k)
                o += nb
            g_count = (g["unit"] != -1).sum(dim=2, dtype=torch.int32)
            res = dict(unit=torch.empty_like(loc["unit"]), score=torch.empty_like(loc["score"]),
                       matched=torch.empty_like(loc["matched"]), tie=torch.empty_like(loc["tie"]),
                       count=torch.empty_like(loc["count"]))
            self._hand_to_library()
            e.merge_topk(G, P * G, k, g["unit"], g["score"], g["matched"], g["tie"] if ties else g["unit"], g_count, out=res)
            res = {key: v[:nq] for key, v in res.items()}
        if out_device:
            return {key: v for key, v in res.items() if key != "tie"}
        return {key: _to_numpy(v) for key, v in res.items() if key != "tie"}


def _to_numpy(t: torch.Tensor) -> np.ndarray:
    a = t.detach().cpu().numpy()
    return a.view(np.uint32) if a.dtype == np.int32 else a
