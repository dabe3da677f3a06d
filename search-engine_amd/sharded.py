"""Document-sharded index over the GPUs of one node: one process (rank) per GPU.

Every rank holds the postings of a contiguous range of documents (and of news articles) with
the full vocabulary.  The statistics job needs one exchange — per-term document frequencies
and the three document totals are sums over shards, merged with an all-reduce (RCCL over xGMI
when the process group's backend is "nccl") — after which every rank computes the same idf.
A search scores the query batch on every shard, all-gathers the per-shard top-k lists and
merges them on each rank with the same ordering rule, so the result is identical to the one
an unsharded index returns.

The reference runs this job on one coordinator against the shared database
(app/batch/runner.py:43-52); its only sharding idiom is `mod(id, nodes)`
(app/batch/duplicate_detection.py:56).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .native import SRC_NEWS, SRC_WEB, DocTotals


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous range [begin, end) of `n` units owned by `rank`."""
    return n * rank // world, n * (rank + 1) // world


class ShardedIndex:
    """`engine` is this rank's `native.Index` (anything with the same build_local / local_stats /
    finalize / search / merge_topk methods works); `device` is where collective buffers live."""

    def __init__(self, engine, device, group=None):
        self.engine = engine
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._buf = None   # int32[V + 16]: df, then the four totals as 16-bit limbs (one all-reduce for both)
        self._df = None
        # gloo is a host backend: with device buffers (rehearsals and tests of the multi-rank path on one GPU) the
        # collectives run on host copies.  With nccl (RCCL over xGMI) they run on the device buffers in place.
        self._via_host = (dist.is_initialized() and self.world > 1 and self.device.type == "cuda"
                          and dist.get_backend(group) == "gloo")

    # ---- statistics job
    def build(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None,
              doc_base=0, article_base=0, unit_major=True):
        """Rows of this rank's documents (unit ids local to the shard); doc_base / article_base
        are the global ids of its first document / article."""
        e = self.engine
        e.build_local(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags,
                      unit_major=unit_major)
        e.set_unit_base(doc_base, article_base)
        return self._reduce_and_finalize(n_terms)

    def update(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None,
               changed_docs=None, changed_articles=None, unit_major=False):
        """Incremental update of this rank's shard (unit ids local to the shard, as in build): the listed units
        are replaced by the rows given, then df and the totals are summed over the shards again and every rank
        recomputes the same idf.  Every rank must call it (with empty arrays when its shard did not change)."""
        self.engine.update(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags,
                           changed_docs=changed_docs, changed_articles=changed_articles, unit_major=unit_major, local=True)
        return self._reduce_and_finalize(n_terms)

    def _reduce_and_finalize(self, n_terms):
        e = self.engine
        if self._buf is None or self._buf.numel() != n_terms + 16:
            self._buf = torch.zeros(n_terms + 16, dtype=torch.int32, device=self.device)
            self._df = self._buf[:n_terms]
        _, t = e.local_stats(self._df)
        if self.world > 1:
            # 64-bit totals travel as four 16-bit limbs each, so that an int32 SUM over the ranks cannot
            # overflow (up to 32768 ranks) and df and totals share one collective
            vals = (t.n_done, t.n_wc, t.sum_wc, t.n_postings)
            limbs = [(int(v) >> (16 * j)) & 0xFFFF for v in vals for j in range(4)]
            self._buf[n_terms:].copy_(torch.tensor(limbs, dtype=torch.int32))
            if self._via_host:
                tmp = self._buf.cpu()
                dist.all_reduce(tmp, group=self.group)
                self._buf.copy_(tmp)
            else:
                dist.all_reduce(self._buf, group=self.group)    # SUM over shards (RCCL over xGMI with backend nccl)
            got = self._buf[n_terms:].tolist()
            t = DocTotals(*(sum(int(got[4 * i + j]) << (16 * j) for j in range(4)) for i in range(4)))
        e.finalize(self._df, t)
        self.totals = t
        return t

    # ---- scoring
    def _pack(self, nq, k):
        """One byte buffer per rank holding score f64[nq, k] | unit | matched | tie i32[nq, k] | count i32[nq], and
        the buffer that receives every rank's copy: the whole exchange of a search is ONE all-gather."""
        key = (nq, k)
        if getattr(self, "_pack_key", None) != key:
            n = nq * k
            self._off = dict(score=0, unit=8 * n, matched=12 * n, tie=16 * n, count=20 * n)
            nbytes = (20 * n + 4 * nq + 7) // 8 * 8
            self._mine = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._all = torch.empty(self.world * nbytes, dtype=torch.uint8, device=self.device)
            self._pack_key = key
        return self._mine, self._all

    def _views(self, buf, nq, k, lists=None):
        """Typed views of one packed buffer (lists=None) or of the gathered buffers, list-major copies."""
        n = nq * k
        out = {}
        for name, dt, cnt, shape in (("score", torch.float64, n, (nq, k)), ("unit", torch.int32, n, (nq, k)),
                                     ("matched", torch.int32, n, (nq, k)), ("tie", torch.int32, n, (nq, k)),
                                     ("count", torch.int32, nq, (nq,))):
            o, nb = self._off[name], cnt * (8 if dt == torch.float64 else 4)
            if lists is None:
                out[name] = buf[o:o + nb].view(dt).view(shape)
            else:
                out[name] = buf.view(lists, -1)[:, o:o + nb].contiguous().view(dt).view((lists,) + shape)
        return out

    def search(self, queries, k, source=SRC_WEB, out_device=False):
        """-> dict(unit, score, matched, count) of the global top-k (global unit ids), the same on every
        rank: numpy arrays, or tensors on `device` with out_device=True."""
        e = self.engine
        nq = len(queries)
        mine, gathered = self._pack(nq, k)
        loc = self._views(mine, nq, k)
        e.search(queries, k, source=source, out=loc)
        if self.world == 1:
            res = loc
        else:
            if self._via_host:
                torch.cuda.synchronize(self.device)
                g_host = torch.empty(gathered.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(g_host, mine.cpu(), group=self.group)
                gathered.copy_(g_host)
            else:
                dist.all_gather_into_tensor(gathered, mine, group=self.group)    # per-shard top-k lists, one collective
            g = self._views(gathered, nq, k, lists=self.world)
            # the collective and the list-major copies made above are only ordered against torch's current stream; the
            # merge below runs on the library's stream, which may be another one: wait here until the lists are in place
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()
            res = dict(unit=torch.empty_like(loc["unit"]), score=torch.empty_like(loc["score"]),
                       matched=torch.empty_like(loc["matched"]), tie=torch.empty_like(loc["tie"]),
                       count=torch.empty_like(loc["count"]))
            e.merge_topk(self.world, nq, k, g["unit"], g["score"], g["matched"], g["tie"], g["count"], out=res)
        if out_device:
            return {key: v for key, v in res.items() if key != "tie"}
        return {key: _to_numpy(v) for key, v in res.items() if key != "tie"}


def _to_numpy(t: torch.Tensor) -> np.ndarray:
    a = t.detach().cpu().numpy()
    return a.view(np.uint32) if a.dtype == np.int32 else a
