"""BAI-driven sharding (SURVEY §8 e): per-reference work estimates and byte ranges from the index, every rank reading
only its references' ranges of a coordinate-sorted BGZF file (mates joined by name).  CPU only: host plumbing."""
import struct

import numpy as np
import pytest

from irfinder_amd import bai, engine
from irfinder_amd import distributed as irfdist
from irfinder_amd.synth.bam import bgzf_compress, build_bai, records_to_bam
from test_host_reader import _fragment_multiset, _split_bam
from util import small_world


def sorted_world(seed=23, n_pairs=6000, n_chr=4):
    ann, sim = small_world(n_pairs=0, seed=seed, n_chr=n_chr, single_frac=0.03)
    head, recs = _split_bam(sim.bam(0, n_pairs))
    extra = _split_bam(records_to_bam(ann.chr_names, ann.chr_lens, [
        dict(ref_id=0, pos=900, flag=0x41, cigar="50M", name="split"), dict(ref_id=2, pos=900, flag=0x81, cigar="50M", name="split"),
        dict(ref_id=-1, pos=-1, flag=0x4, cigar="", name="nocoor")]))[1]
    recs = recs + extra

    def key(r):
        ref_id, pos = struct.unpack_from("<ii", r, 4)
        return (ref_id if ref_id >= 0 else 1 << 30, pos)
    recs.sort(key=key)
    return ann, head, recs, bgzf_compress(head + b"".join(recs), block=0x4000)


def test_bai_roundtrip_counts_ranges_and_weights():
    ann, head, recs, z = sorted_world()
    idx = bai.parse_bai(build_bai(z))
    assert idx.n_ref == len(ann.chr_names) and idx.n_no_coor == 1
    by_ref = {}
    for r in recs:
        by_ref.setdefault(struct.unpack_from("<i", r, 4)[0], []).append(r)
    for c in range(idx.n_ref):
        assert idx.refs[c].n_mapped == len(by_ref.get(c, [])) and idx.refs[c].n_unmapped == 0
        rng = idx.ref_range(c)
        if c in by_ref:
            assert bai.read_virtual_range(z, *rng) == b"".join(by_ref[c])       # exactly that reference's records
            assert len(idx.refs[c].linear) >= (max(struct.unpack_from("<i", r, 8)[0] for r in by_ref[c]) >> 14) + 1
        else:
            assert rng is None
    w = idx.weights()
    assert w.tolist() == [len(by_ref.get(c, [])) for c in range(idx.n_ref)]
    masks = irfdist.shard_chromosomes(w, 2)
    assert np.array_equal(masks[0] + masks[1], np.ones(idx.n_ref, np.uint8))
    # without the pseudo-bin the weights fall back to the compressed span of the linear index
    for r in idx.refs:
        r.n_mapped = None
    w2 = idx.weights()
    assert (w2 > 0).sum() == sum(1 for c in by_ref if c >= 0)


def test_sharded_readers_cover_the_whole_file():
    ann, head, recs, z = sorted_world(seed=29)
    idx = bai.parse_bai(build_bai(z))
    whole = engine.Reader(engine.default_params(mate_join=1))
    whole.feed(z, bgzf=True)
    whole.finish()
    want, want_stats = _fragment_multiset(whole.take_batch()), whole.stats()
    assert want_stats["n_error"] == 2 and want_stats["n_skipped"] == 1
    world = 3
    masks = irfdist.shard_chromosomes(idx.weights(), world)
    got, stats = [], {}
    for rank in range(world):
        rd = engine.Reader(engine.default_params(mate_join=1))
        rd.feed(head)                                            # the header, then only this rank's byte ranges
        for c in np.nonzero(masks[rank])[0]:
            rng = idx.ref_range(int(c))
            if rng:
                rd.feed(bai.read_virtual_range(z, *rng))
        rd.finish()
        b = rd.take_batch()
        assert set(np.unique(b.frag_chr).tolist()) <= set(np.nonzero(masks[rank])[0].tolist())
        got += _fragment_multiset(b)
        for k, v in rd.stats().items():
            stats[k] = stats.get(k, 0) + v
    assert sorted(got) == want
    # the record without coordinates belongs to no reference range; everything else adds up (the pair split over two
    # chromosomes is one orphan on each side)
    assert stats["n_records"] == want_stats["n_records"] - 1 and stats["n_skipped"] == 0
    for k in ("n_pairs", "n_singles", "n_pairs_long", "n_pairs_short", "n_pairs_intersect", "n_error"):
        assert stats[k] == want_stats[k], k


def test_bai_malformed():
    ann, head, recs, z = sorted_world(n_pairs=300)
    good = build_bai(z)
    with pytest.raises(ValueError):
        bai.parse_bai(b"BAM\1" + good[4:])
    with pytest.raises(ValueError):
        bai.parse_bai(good[:len(good) // 2])
    with pytest.raises(ValueError):
        bai.read_virtual_range(z, 0, (len(z) + 100) << 16)
