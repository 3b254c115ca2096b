"""Synthetic workloads at BASELINE.json scale: annotation + simulated pairs -> SoA fragment batches.

Fragments are produced by the product's own host path (synthetic BAM records -> irf_reader -> SoA), chunked and
multi-threaded (the C calls release the GIL), so generating 50 M pairs takes seconds, not minutes.
"""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .. import engine
from .._abi import Batch


def _one_chunk(sim, header, first, n, params):
    r = engine.Reader(params)
    r.feed(header)
    r.feed(sim.records(first, n))
    r.finish()
    b = r.take_batch()
    st = r.stats()
    r.close()
    return b, st


def generate_batches(sim, first, n_pairs, chunk=500_000, threads=8, params=None):
    """Returns (list of Batch, summed reader stats) for pairs [first, first+n_pairs)."""
    header = sim.header()
    jobs = [(first + o, min(chunk, n_pairs - o)) for o in range(0, n_pairs, chunk)]
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        res = list(ex.map(lambda j: _one_chunk(sim, header, j[0], j[1], params), jobs))
    stats = {}
    for _, st in res:
        for k, v in st.items():
            stats[k] = stats.get(k, 0) + v
    return [b for b, _ in res], stats


def concat_batches(batches):
    """One Batch out of many (block offsets re-based)."""
    if not batches:
        return Batch.empty()
    offs, base = [np.zeros(1, np.uint32)], 0
    for b in batches:
        offs.append((b.frag_blk_off[1:].astype(np.uint64) + base).astype(np.uint32))
        base += b.n_blk
    cat = lambda n: np.concatenate([getattr(b, n) for b in batches])
    return Batch(frag_chr=cat("frag_chr"), frag_flags=cat("frag_flags"), frag_start=cat("frag_start"),
                 frag_end=cat("frag_end"), frag_blk_off=np.concatenate(offs), blk_start=cat("blk_start"),
                 blk_len=cat("blk_len"), blk_read=cat("blk_read"))


def sort_by_coordinate(batch):
    """Coordinate order (chr, start) — what a coordinate-sorted BAM hands over (north_star)."""
    key = (batch.frag_chr.astype(np.uint64) << np.uint64(32)) | batch.frag_start.astype(np.uint64)
    return batch.select(np.argsort(key, kind="stable"))


def restrict_refs(ref_texts, keep_chroms):
    """The four reference files with only the lines of the given chromosomes (CPU-baseline sample)."""
    keep = set(keep_chroms)
    out = []
    for t in ref_texts:
        out.append("".join(l for l in t.splitlines(True) if l.split("\t", 1)[0] in keep))
    return tuple(out)
