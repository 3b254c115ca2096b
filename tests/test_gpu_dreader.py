"""GPU parity of the device reader (SURVEY §8 f N1): BGZF inflate + CRC, record location, mate pairing and fragment
assembly on the device vs zlib and vs the host reader / the oracle's reader, column by column, through the C ABI.
"""
import struct
import zlib

import numpy as np
import pytest

import orc
from irfinder_amd import engine
from irfinder_amd._abi import Batch
from irfinder_amd.synth.bam import BGZF_EOF, bgzf_compress, records_to_bam
from util import assert_results_equal, small_world

pytestmark = pytest.mark.gpu


def member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    comp = co.compress(chunk) + co.flush()
    assert len(comp) + 26 <= 65536
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
            struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))


def concat(batches):
    bs = [b for b in batches if b.n_frag]
    if not bs:
        return Batch.empty()
    off, base = [np.zeros(1, np.uint32)], 0
    for b in bs:
        off.append(b.frag_blk_off[1:] + np.uint32(base))
        base += b.n_blk
    cols = {n: np.concatenate([getattr(b, n) for b in bs]) for n, _ in Batch.FIELDS if n != "frag_blk_off"}
    return Batch(frag_blk_off=np.concatenate(off), **cols)


def host_read(z, **over):
    r = engine.Reader(engine.default_params(**over))
    r.feed(z, bgzf=True)
    r.finish()
    return r.header(), r.take_batch(), r.stats()


def device_read(z, pieces=None, pipeline=False, **over):
    """pieces: iterable of chunk sizes (cycled); None = one feed.  pipeline: irf_dreader_set_pipeline (the batch after a
    feed is the one of the feed before it, the last one appears after finish)."""
    d = engine.DeviceReader(engine.default_params(**over), pipeline=pipeline)
    out, pos, k = [], 0, 0
    while pos < len(z):
        n = len(z) if pieces is None else int(pieces[k % len(pieces)])
        d.feed(z[pos:pos + n])
        out.append(d.batch_to_host())
        pos += n
        k += 1
    if pipeline:
        assert out[0].n_frag == 0
    d.finish()
    if pipeline:
        out.append(d.batch_to_host())
    return d.header(), concat(out), d.stats(), d


def test_inflate_matches_zlib_on_every_block_type():
    rng = np.random.default_rng(1)
    text = (b"ACGTTGCA" * 50 + bytes(rng.integers(65, 90, 3000, dtype=np.uint8).tolist())) * 12
    noise = bytes(rng.integers(0, 256, 60000, dtype=np.uint8).tolist())
    skew = bytes(np.minimum(rng.geometric(0.08, 65000), 255).astype(np.uint8).tolist())
    chunks = []
    for data in (text[:65000], noise, skew, b"\0" * 65536, b"a", b"ab" * 300, bytes(range(256)) * 200, text[:17], text[:4099]):
        for level in (0, 1, 6, 9):
            chunks.append((data, level, zlib.Z_DEFAULT_STRATEGY))
        chunks.append((data, 6, zlib.Z_FIXED))
        chunks.append((data, 6, zlib.Z_HUFFMAN_ONLY))
        chunks.append((data, 6, zlib.Z_RLE))
    parts, want = [], []
    for data, level, strat in chunks:
        if level == 0 and len(data) > 65000:
            data = data[:65000]              # stored blocks: the member must stay below 64 KiB
        parts.append(member(data, level, strat))
        want.append(data)
        if len(parts) % 7 == 0:
            parts.append(BGZF_EOF)           # empty members in the middle
    # several deflate blocks inside one member (Z_FULL_FLUSH closes a block; the second one reuses the window)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    two = co.compress(text[:20000]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(text[:20000]) + co.flush(zlib.Z_SYNC_FLUSH) + \
        co.compress(noise[:100]) + co.flush()
    parts.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(two) + 25) + two +
                 struct.pack("<II", zlib.crc32(text[:20000] * 2 + noise[:100]) & 0xffffffff, 40100))
    want.append(text[:20000] * 2 + noise[:100])
    z = b"".join(parts)
    got = engine.gpu_bgzf_inflate(z)
    exp = np.frombuffer(b"".join(want), np.uint8)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), int(np.argmax(got != exp))
    assert engine.gpu_bgzf_inflate(b"").size == 0 and engine.gpu_bgzf_inflate(BGZF_EOF).size == 0


def test_inflate_short_overlapping_matches_of_every_period():
    """Matches that overlap their own output (distance < length): runs of every period 1 .. 9 and every length up to 40,
    between random bytes — the deferred copy stores periods below 8 as a register pattern (all four 8-byte words, every
    phase), 8 and above go through the in-place copy."""
    rng = np.random.default_rng(11)
    parts, want = [], []
    for rep in range(4):
        buf = bytearray()
        for period in range(1, 10):
            for total in range(period + 3, 41):
                unit = bytes(rng.integers(0, 256, period, dtype=np.uint8).tolist())
                buf += (unit * (total // period + 1))[:total]
                buf += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tolist())
        data = bytes(buf[:65000])
        for level, strat in ((6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED)):
            parts.append(member(data, level, strat))
            want.append(data)
    got = engine.gpu_bgzf_inflate(b"".join(parts))
    exp = np.frombuffer(b"".join(want), np.uint8)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), int(np.argmax(got != exp))


def test_inflate_refuses_corrupt_members():
    rng = np.random.default_rng(2)
    data = bytes(rng.integers(65, 70, 30000, dtype=np.uint8).tolist())
    good = member(data) + member(data[::-1]) + BGZF_EOF
    assert engine.gpu_bgzf_inflate(good).size == 60000
    bad_crc = bytearray(good)
    bad_crc[len(member(data)) - 8] ^= 1                       # CRC field of the first member
    with pytest.raises(engine.IrfError):
        engine.gpu_bgzf_inflate(bytes(bad_crc))
    bad_isize = bytearray(good)
    struct.pack_into("<I", bad_isize, len(member(data)) - 4, 29999)
    with pytest.raises(engine.IrfError):
        engine.gpu_bgzf_inflate(bytes(bad_isize))
    n_bad = 0
    for at in range(18, len(member(data)) - 8, 97):            # payload corruption: never a wrong answer, never a fault
        b = bytearray(good)
        b[at] ^= 0x40
        try:
            out = engine.gpu_bgzf_inflate(bytes(b))
            assert out.size == 60000                           # (a flip zlib would accept too must give zlib's bytes)
            assert bytes(out) == zlib.decompress(bytes(b[18:len(member(data)) - 8]), -15) + data[::-1]
        except engine.IrfError:
            n_bad += 1
    assert n_bad > 50
    with pytest.raises(engine.IrfError):
        engine.gpu_bgzf_inflate(good[:-5])                     # not whole members


@pytest.mark.parametrize("stranded", [False, True])
def test_device_reader_equals_host_reader_and_oracle(stranded):
    ann, sim = small_world(n_pairs=0, stranded=stranded, seed=11, with_seq=2)
    bam = sim.bam(0, 30000)
    z = bgzf_compress(bam, block=0xff00, level=6)
    (hn, hl), hb, hs = host_read(z)
    (dn, dl), db, ds, d = device_read(z)
    assert (dn, dl) == (hn, hl) == (ann.chr_names, ann.chr_lens)
    assert db.equals(hb) and ds == hs
    ob, ost, rc = orc.bam_to_batch(bam)
    assert rc == 0 and db.equals(ob) and ds == ost
    assert ds["n_pairs_intersect"] > 0 and ds["n_pairs_long"] > 0 and ds["n_singles"] > 0 and ds["n_pairs_short"] > 0
    assert d.member_slots() >= 64 * 64 and d.member_slots() % 64 == 0
    t = d.timing()
    assert t["bytes_in"] == len(z) and t["bytes_inflated"] == len(bam) and t["n_members"] == (len(bam) + 0xfeff) // 0xff00 + 1


def test_device_reader_ragged_feeds_and_member_sizes():
    ann, sim = small_world(seed=5, with_seq=2, single_frac=0.2)
    bam = sim.bam(100, 9000)
    ob, ost, _ = orc.bam_to_batch(bam)
    for block, pieces in ((20000, [len(bam)]), (0xff00, [70001, 13, 250000]), (3000, [4096]), (700, [1, 7, 50000, 333]),
                          (0xff00, [100])):
        z = bgzf_compress(bam, block=block, level=1)
        for pipeline in (False, True):
            _, db, ds, _ = device_read(z, pieces, pipeline)
            assert db.equals(ob) and ds == ost, (block, pieces, pipeline)


def test_device_reader_edge_records_and_switches():
    chrs, lens = ["c1", "c2"], [10000, 5000]
    recs = [dict(ref_id=0, pos=1, flag=0x41, cigar="10M"), dict(ref_id=1, pos=1, flag=0x81, cigar="10M"),     # different refs
            dict(ref_id=0, pos=1, flag=0x45, cigar="10M"), dict(ref_id=0, pos=1, flag=0x81, cigar="10M"),     # unmapped mate
            dict(ref_id=0, pos=5, flag=0x0, cigar="4S"),                                                    # no aligned block
            dict(ref_id=-1, pos=-1, flag=0x0, cigar="10M"),                                                  # no reference
            dict(ref_id=0, pos=100, flag=0x61, cigar="20M300N20M"), dict(ref_id=0, pos=90, flag=0x91, cigar="15M"),
            dict(ref_id=0, pos=200, flag=0x63, cigar="30M100N30M2D10M"), dict(ref_id=0, pos=250, flag=0x93, cigar="5S40M50N20M"),
            dict(ref_id=0, pos=1, flag=0x41, cigar="10M")]                                                   # dangling mate
    bam = records_to_bam(chrs, lens, recs)
    ob, ost, rc = orc.bam_to_batch(bam)
    for pieces, pipeline in ((None, False), ([1], False), ([29], False), (None, True), ([29], True), ([400, 3], True)):
        _, db, ds, _ = device_read(bgzf_compress(bam, block=64), pieces, pipeline)
        assert db.equals(ob) and ds == ost, (pieces, pipeline)
    assert ost["n_error"] == 2 + 1 + 1 + 1 and ost["n_skipped"] == 2 and ob.n_frag == 2
    # header only; header spread over many members
    (dn, dl), db, ds, _ = device_read(bgzf_compress(records_to_bam(chrs, lens, []), block=5))
    assert (dn, dl) == (chrs, lens) and db.n_frag == 0 and ds["n_records"] == 0
    # the reader's switches (OQ-1/2/3)
    recs = [dict(ref_id=0, pos=100, flag=0x63, cigar="50M"), dict(ref_id=0, pos=150, flag=0x93, cigar="50M"),
            dict(ref_id=0, pos=100, flag=0x63 | 0x100, cigar="50M"), dict(ref_id=0, pos=400, flag=0x93, cigar="50M"),
            dict(ref_id=0, pos=1000, flag=0x63, cigar="30M500N30M"), dict(ref_id=0, pos=1200, flag=0x93, cigar="400M")]    # read 2 starts in read 1's N gap
    bam = records_to_bam(chrs, lens, recs)
    for over in ({}, {"pair_long_inclusive": 1}, {"skip_flag_mask": 0x104}, {"pair_short_inclusive": 0}, {"gap_pair_is_intersect": 1}):
        ob, ost, _ = orc.bam_to_batch(bam, orc.default_params(**over))
        _, db, ds, _ = device_read(bgzf_compress(bam), **over)
        assert db.equals(ob) and ds == ost, over


def test_device_reader_record_payload_that_looks_like_records():
    """The record-start speculation may be fooled; the verification may not: read names and sequence bytes that look
    like plausible record headers (block_size, refID, pos, l_read_name ...) change nothing in the result."""
    chrs, lens = ["c1"], [1 << 20]
    # the smallest well-formed record: no cigar, no sequence, an empty name
    fake = struct.pack("<iiiBBHHHiiii", 33, 0, 77, 1, 0, 0, 0, 0, 0, 0, 99, 0) + b"\0"
    assert len(fake) == 37
    recs = []
    for i in range(3000):
        recs.append(dict(ref_id=0, pos=1000 + 7 * i, flag=0x0 if i % 3 else 0x10, cigar="80M200N70M", name="n%d" % i, l_seq=150))
    bam = bytearray(records_to_bam(chrs, lens, recs))
    # three chained fakes inside the sequence + quality bytes of every record, at varying offsets
    o = len(records_to_bam(chrs, lens, []))
    k = 0
    while o < len(bam):
        bs, = struct.unpack_from("<i", bam, o)
        l_name = bam[o + 12]
        pay = o + 36 + l_name + 4 * 3
        room = o + 4 + bs - pay
        assert room == 225
        at = pay + (k % 16)
        bam[at:at + 111] = fake * 3
        o += 4 + bs
        k += 1
    bam = bytes(bam)
    ob, ost, rc = orc.bam_to_batch(bam)
    assert rc == 0 and ob.n_frag == 3000
    for pieces, pipeline in ((None, False), ([5000], False), ([5000], True)):
        _, db, ds, d = device_read(bgzf_compress(bam, block=0x4000), pieces, pipeline)
        assert db.equals(ob) and ds == ost
        assert 1 <= d.timing()["chain_rounds"] < 200            # reported: rounds beyond 2-3 mean the speculation was fooled


def test_four_samples_streamed_through_one_context_with_overlapped_reading():
    """BASELINE configs[4] shape at test size: four different BAMs (same reference) through ONE context, irf_reset
    between them, the device reader of sample k+1 running on a second host thread under the count / finalize of sample
    k; every sample's tables equal the oracle's on that sample's bytes."""
    from concurrent.futures import ThreadPoolExecutor
    from util import assert_results_equal
    ann, sim = small_world(seed=31, stranded=True)
    bams = [sim.bam(20000 * k, 9000 + 1000 * k) for k in range(4)]          # different reads, different sizes
    zs = [bgzf_compress(b, block=0x3000) for b in bams]
    eng = engine.Engine(engine.RefTables.from_texts(ann.ref_texts()))

    def read_sample(k):
        d = engine.DeviceReader()
        d.feed(zs[k])
        d.finish()
        return d, d.batch_struct(), d.header()

    with ThreadPoolExecutor(1) as ex:
        fut = ex.submit(read_sample, 0)
        for k in range(4):
            d, b, (names, lens) = fut.result()
            if k + 1 < 4:
                fut = ex.submit(read_sample, k + 1)
            if k == 0:
                eng.chr_map_update(names, lens)
            else:
                eng.reset()
            eng.submit(b)
            eng.finalize()
            got = eng.results()
            d.close()
            o = orc.Oracle(ann.ref_texts())
            o.process_bam(bams[k])
            o.finalize()
            assert_results_equal(got, o.results(), rel=0.0, what="sample %d" % k)
            assert got["n_fragments"] > 8000


def _split_bam(bam):
    l_text, = struct.unpack_from("<i", bam, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", bam, o)
    o += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", bam, o)
        o += 4 + l_name + 4
    head, recs = bam[:o], []
    while o < len(bam):
        bs, = struct.unpack_from("<i", bam, o)
        recs.append(bam[o:o + 4 + bs])
        o += 4 + bs
    return head, recs


def test_device_reader_joins_mates_by_name_on_coordinate_sorted_input():
    """params.mate_join = 1 on the device (north_star's input order): the coordinate-sorted file through
    irf_dreader_* gives the host reader's batch — column by column, in the same order — and the same counters, for any
    chunking and with the pipeline; irregular records included: an orphan, a pair split over two chromosomes (its mates
    are feeds apart), an unmapped mate, a displaced first mate (a second record of the SAME mate under one name),
    unpaired reads in between."""
    ann, sim = small_world(seed=37, single_frac=0.05)
    bam = sim.bam(0, 9000)
    head, recs = _split_bam(bam)
    extra = records_to_bam(ann.chr_names, ann.chr_lens, [
        dict(ref_id=0, pos=5000, flag=0x41, cigar="50M", name="orphan"),
        dict(ref_id=0, pos=6000, flag=0x41, cigar="50M", name="split"), dict(ref_id=2, pos=6000, flag=0x81, cigar="50M", name="split"),
        dict(ref_id=0, pos=7000, flag=0x41, cigar="50M", name="half"), dict(ref_id=0, pos=7000, flag=0x81 | 0x4, cigar="50M", name="half"),
        dict(ref_id=1, pos=100, flag=0x41, cigar="50M", name="twice"), dict(ref_id=1, pos=200, flag=0x41 | 0x100, cigar="40M", name="twice"),
        dict(ref_id=1, pos=90000, flag=0x81, cigar="50M10N20M", name="twice"),
        dict(ref_id=1, pos=300000, flag=0x41, cigar="30M", name="late-orphan")])
    recs_x = _split_bam(extra)[1]

    def key(rec):
        ref_id, pos = struct.unpack_from("<ii", rec, 4)
        return (ref_id if ref_id >= 0 else 1 << 30, pos)
    srt = sorted(recs + recs_x, key=key)
    data = head + b"".join(srt)
    P = dict(mate_join=1)
    r = engine.Reader(engine.default_params(**P))
    r.feed(data)
    r.finish()
    want, want_stats = r.take_batch(), r.stats()
    assert want_stats["n_error"] >= 5 and want_stats["n_pairs"] > 8000 and want_stats["n_singles"] > 100
    z = bgzf_compress(data, block=0x3000)
    for pieces, pipeline in ((None, False), ([len(z) // 7 + 1], False), ([70001, 333], False), ([len(z) // 5 + 1], True), ([9999], True)):
        _, db, ds, d = device_read(z, pieces, pipeline, **P)
        assert ds == want_stats, (pieces, pipeline, ds, want_stats)
        assert db.equals(want), (pieces, pipeline)
    # the same file read the upstream way (adjacent records are mates) is mis-paired: that is D-1
    _, db0, ds0, _ = device_read(z)
    assert not db0.equals(want)


def test_device_reader_fault_injection():
    ann, sim = small_world(seed=5)
    bam = sim.bam(0, 2000)
    z = bgzf_compress(bam, block=4000)
    d = engine.DeviceReader()
    d.feed(bgzf_compress(bam[:-5], block=4000))
    with pytest.raises(engine.IrfError, match="truncated record"):
        d.finish()
    d = engine.DeviceReader()
    with pytest.raises(engine.IrfError, match="magic"):
        d.feed(bgzf_compress(b"BAX\1" + bam[4:]))
    bad = bytearray(z)
    bad[len(z) // 2] ^= 0xff
    d = engine.DeviceReader()
    with pytest.raises(engine.IrfError, match="BGZF"):
        d.feed(bytes(bad))
    with pytest.raises(engine.IrfError) as e2:
        d.feed(z)                                              # the error is sticky: same code, same text
    assert e2.value.code == -2 and "BGZF" in str(e2.value)
    d = engine.DeviceReader()
    d.feed(z[:-40])
    with pytest.raises(engine.IrfError, match="truncated block"):
        d.finish()
    # a record whose block_size is impossible / whose fields overrun it: refused like the host reader refuses it
    head = len(records_to_bam(ann.chr_names, ann.chr_lens, []))
    for patch, msg in (((head + 40000, struct.pack("<i", 7)), "block_size"), ((head + 12, b"\xff"), "overrun")):
        raw = bytearray(bam)
        o = head
        while o < patch[0] and patch[0] != head + 12:
            o += 4 + struct.unpack_from("<i", raw, o)[0]
        at = o if patch[0] != head + 12 else head + 12
        raw[at:at + len(patch[1])] = patch[1]
        r = engine.Reader()
        with pytest.raises(engine.IrfError, match=msg):
            r.feed(bytes(raw))
            r.finish()
        d = engine.DeviceReader()
        with pytest.raises(engine.IrfError, match=msg):
            d.feed(bgzf_compress(bytes(raw)))
            d.finish()


def test_device_reader_pipeline_errors_and_long_carry():
    """Pipeline on: errors of a feed surface one call later (next feed or finish); a carry longer than the room kept for
    it in front of the next feed's members (a 1.8 MB record cut by many feeds) takes the path that moves the members."""
    ann, sim = small_world(seed=5)
    bam = sim.bam(0, 2000)
    z = bgzf_compress(bam, block=4000)
    bad = bytearray(z)
    bad[len(z) // 2] ^= 0xff
    d = engine.DeviceReader(pipeline=True)
    d.feed(bytes(bad))                                         # copied and inflated; not looked at yet
    with pytest.raises(engine.IrfError, match="BGZF"):
        d.finish()
    d = engine.DeviceReader(pipeline=True)
    d.feed(bytes(bad[:len(z) // 4]))
    with pytest.raises(engine.IrfError, match="BGZF"):
        d.feed(bytes(bad[len(z) // 4:]))                       # inflates the second piece, parses the first: fine ...
        d.finish()                                             # ... the corrupt member is met here
    d = engine.DeviceReader(pipeline=True)
    d.feed(bgzf_compress(bam[:-5], block=4000))
    with pytest.raises(engine.IrfError, match="truncated record"):
        d.finish()
    d = engine.DeviceReader()
    d.feed(z[:100])
    with pytest.raises(engine.IrfError):
        d._chk(d.L.irf_dreader_set_pipeline(d.h, 1))           # only before the first feed
    # one very long record between ordinary ones
    chrs, lens = ["c1"], [1 << 22]
    recs = [dict(ref_id=0, pos=1000 + 9 * i, flag=0x0, cigar="80M200N70M", name="a%d" % i, l_seq=150) for i in range(50)]
    recs.append(dict(ref_id=0, pos=5000, flag=0x10, cigar="100M1000N100M", name="long", l_seq=1_200_000))
    recs += [dict(ref_id=0, pos=6000 + 9 * i, flag=0x0, cigar="75M", name="b%d" % i, l_seq=75) for i in range(50)]
    raw = bytearray(records_to_bam(chrs, lens, recs))
    rng = np.random.default_rng(8)
    o = len(records_to_bam(chrs, lens, []))
    while o < len(raw):
        bs, = struct.unpack_from("<i", raw, o)
        if bs > 1_000_000:                                     # incompressible sequence + quality bytes
            pay = o + 36 + raw[o + 12] + 4 * 3
            raw[pay:o + 4 + bs] = rng.integers(0, 256, o + 4 + bs - pay, dtype=np.uint8).tobytes()
        o += 4 + bs
    raw = bytes(raw)
    ob, ost, rc = orc.bam_to_batch(raw)
    assert rc == 0 and ob.n_frag == 101
    z = bgzf_compress(raw, block=0xff00, level=1)
    assert len(z) > 1_700_000
    for pieces, pipeline in (([300_000], True), ([300_000], False), ([1_000_000, 50_000], True)):
        _, db, ds, _ = device_read(z, pieces, pipeline)
        assert db.equals(ob) and ds == ost, (pieces, pipeline)


@pytest.mark.parametrize("pipeline", [False, True])
def test_device_reader_feeds_the_engine_without_a_host_copy(pipeline):
    """compressed bytes -> irf_dreader -> DEVICE batch -> irf_submit_batch -> IR table == oracle on the same BAM."""
    ann, sim = small_world(seed=3, with_seq=2)
    bam = sim.bam(0, 40000)
    z = bgzf_compress(bam, block=0xff00, level=6)
    d = engine.DeviceReader(pipeline=pipeline)
    e = None
    step = 1 << 20
    for pos in range(0, len(z), step):
        if e is not None:
            e.sync()                                          # the batch before the last one is overwritten by this feed
        d.feed(z[pos:pos + step])
        if e is None:
            names, lens = d.header()
            e = engine.Engine(engine.RefTables.from_texts(ann.ref_texts()))
            e.chr_map_update(names, lens)
        b = d.batch_struct()
        if b.n_frag:
            e.submit(b)
    e.sync()
    d.finish()
    if pipeline:
        b = d.batch_struct()
        assert b.n_frag
        e.submit(b)
    e.finalize()
    g = e.results()
    ob, _, _ = orc.bam_to_batch(bam)
    o = orc.Oracle(ann.ref_texts(), orc.default_params())
    o.chr_map_update(names, lens)
    o.process_batch(ob)
    o.finalize()
    assert_results_equal(g, o.results(), rel=0.0, what="device reader -> engine")
    assert g["n_fragments"] == ob.n_frag > 30000
