"""Host logic of the product (no GPU): BAM reader / fragment assembly vs the oracle's reader, C-ABI surface."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import orc
from irfinder_amd import engine
from irfinder_amd._abi import IrfParams
from irfinder_amd.synth.bam import bgzf_compress, records_to_bam
from util import small_world

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "irf.h")).read()
    declared = set(re.findall(r"\b(irf_[a-z0-9_]+)\s*\(", hdr))
    out = subprocess.check_output(["nm", "-D", engine.LIB_PATH]).decode()
    exported = set(re.findall(r" T (irf_\w+)", out))
    assert declared and declared <= exported, sorted(declared - exported)
    L = engine.lib()
    for name in declared:
        assert hasattr(L, name)


def test_rccl_collection_recipe_is_plain_c_and_its_program_is_built(tmp_path):
    """host/rccl_collect.h (the one-process-per-GPU collection over RCCL) compiles as C99 against
    <rccl/rccl.h>, and the program around it was built and links librccl (its GPU run: test_gpu_parity.py)."""
    src = tmp_path / "rc.c"
    src.write_text('#include "irfinder_amd/host/rccl_collect.h"\nint main(void) { return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + ROOT, "-fsyntax-only", str(src)])
    exe = os.path.join(ROOT, "irfinder_amd", "irf_rccl_collect")
    assert os.path.exists(exe)
    needed = subprocess.check_output(["readelf", "-d", exe]).decode()
    assert "librccl" in needed and "libirfhip" in needed


def test_params_defaults_match_oracle_and_struct_size():
    a, b = engine.default_params(), orc.default_params()
    assert a.struct_size == C.sizeof(IrfParams) == b.struct_size
    for f, _ in IrfParams._fields_:
        assert getattr(a, f) == getattr(b, f), f


def test_cigar2block_kat1_product():
    for cig, exp in [("50M1000N50M", ([0, 1050], [50, 50], 1100)), ("5S45M2D5M100N3I50M", ([0, 152], [52, 50], 202)),
                     ("10M100N200N10M", ([0, 310], [10, 10], 320)), ("10S5I", ([], [], 0)), ("3=2X5M", ([0], [10], 10))]:
        assert engine.cigar2block(orc.cigar_words(cig)) == exp
        assert orc.cigar2block(cig) == exp


@pytest.mark.parametrize("stranded", [False, True])
def test_reader_matches_oracle_on_synthetic_bam(stranded):
    ann, sim = small_world(n_pairs=0, stranded=stranded, seed=11)
    bam = sim.bam(0, 30000)
    ob, ost, rc = orc.bam_to_batch(bam)
    assert rc == 0
    r = engine.Reader()
    r.feed(bam)
    r.finish()
    assert r.header() == (ann.chr_names, ann.chr_lens)
    pb = r.take_batch()
    assert pb.equals(ob)
    assert r.stats() == ost
    assert ost["n_pairs_intersect"] > 0 and ost["n_pairs_long"] > 0 and ost["n_singles"] > 0


def test_reader_chunked_and_bgzf():
    ann, sim = small_world(seed=5)
    bam = sim.bam(100, 5000)
    ref, _, _ = orc.bam_to_batch(bam)
    rng = np.random.default_rng(0)
    # raw stream in ragged chunks (including 1-byte pieces across the header)
    r = engine.Reader()
    pos = 0
    while pos < len(bam):
        n = int(rng.integers(1, 4000)) if pos > 64 else 1
        r.feed(bam[pos:pos + n])
        pos += n
    r.finish()
    assert r.take_batch().equals(ref)
    # BGZF, fed in ragged chunks
    z = bgzf_compress(bam, block=20000)
    r = engine.Reader()
    pos = 0
    while pos < len(z):
        n = int(rng.integers(1, 30000))
        r.feed(z[pos:pos + n], bgzf=True)
        pos += n
    r.finish()
    assert r.take_batch().equals(ref)


def test_reader_fault_injection():
    ann, sim = small_world(seed=5)
    bam = sim.bam(0, 200)
    r = engine.Reader()
    r.feed(bam[:-5])
    with pytest.raises(engine.IrfError):
        r.finish()                                  # truncated record
    r = engine.Reader()
    with pytest.raises(engine.IrfError):
        r.feed(b"BAX\1" + bam[4:])                  # bad magic
    z = bytearray(bgzf_compress(bam, block=4000))
    z[40] ^= 0xff                                   # corrupt deflate payload
    r = engine.Reader()
    with pytest.raises(engine.IrfError):
        r.feed(bytes(z), bgzf=True)
        r.finish()
    r = engine.Reader()
    r.feed(bgzf_compress(bam)[:-40], bgzf=True)     # truncated BGZF block
    with pytest.raises(engine.IrfError):
        r.finish()


def test_reader_edge_records():
    chrs, lens = ["c1", "c2"], [10000, 5000]
    recs = [dict(ref_id=0, pos=1, flag=0x41, cigar="10M"), dict(ref_id=1, pos=1, flag=0x81, cigar="10M"),     # different refs
            dict(ref_id=0, pos=1, flag=0x45, cigar="10M"), dict(ref_id=0, pos=1, flag=0x81, cigar="10M"),     # unmapped mate
            dict(ref_id=0, pos=5, flag=0x0, cigar="4S"),                                                    # no aligned block
            dict(ref_id=-1, pos=-1, flag=0x0, cigar="10M"),                                                  # no reference
            dict(ref_id=0, pos=100, flag=0x61, cigar="20M300N20M"), dict(ref_id=0, pos=90, flag=0x91, cigar="15M"),
            dict(ref_id=0, pos=1, flag=0x41, cigar="10M")]                                                   # dangling mate
    bam = records_to_bam(chrs, lens, recs)
    ob, ost, rc = orc.bam_to_batch(bam)
    r = engine.Reader()
    r.feed(bam)
    r.finish()
    assert r.take_batch().equals(ob) and r.stats() == ost
    assert ost["n_error"] == 2 + 1 + 1 + 1 and ost["n_skipped"] == 2 and ob.n_frag == 1
    # empty BAM (header only)
    r = engine.Reader()
    r.feed(records_to_bam(chrs, lens, []))
    r.finish()
    assert r.take_batch().n_frag == 0


def test_oq_switches_reach_the_reader():
    chrs, lens = ["c1"], [10000]
    recs = [dict(ref_id=0, pos=100, flag=0x63, cigar="50M"), dict(ref_id=0, pos=150, flag=0x93, cigar="50M"),
            dict(ref_id=0, pos=100, flag=0x63 | 0x100, cigar="50M"), dict(ref_id=0, pos=400, flag=0x93, cigar="50M"),
            dict(ref_id=0, pos=1000, flag=0x63, cigar="30M500N30M"), dict(ref_id=0, pos=1200, flag=0x93, cigar="400M")]    # read 2 starts in read 1's N gap
    bam = records_to_bam(chrs, lens, recs)
    for over in ({}, {"pair_long_inclusive": 1}, {"skip_flag_mask": 0x104}, {"pair_short_inclusive": 0}, {"gap_pair_is_intersect": 1}):
        ob, ost, _ = orc.bam_to_batch(bam, orc.default_params(**over))
        r = engine.Reader(engine.default_params(**over))
        r.feed(bam)
        r.finish()
        assert r.take_batch().equals(ob) and r.stats() == ost, over


def _split_bam(bam):
    """raw BAM bytes -> (header bytes, [record bytes])."""
    import struct
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


def _fragment_multiset(b):
    out = []
    for i in range(b.n_frag):
        a, z = int(b.frag_blk_off[i]), int(b.frag_blk_off[i + 1])
        out.append((int(b.frag_chr[i]), int(b.frag_flags[i]), int(b.frag_start[i]), int(b.frag_end[i]),
                    tuple(zip(b.blk_start[a:z].tolist(), b.blk_len[a:z].tolist(), b.blk_read[a:z].tolist()))))
    return sorted(out)


def test_mate_join_on_coordinate_sorted_input():
    """params.mate_join: a coordinate-sorted file read with the name join gives the same multiset of
    fragments and the same reader counters as its name-collated form read the upstream way (adjacent records)."""
    import struct
    ann, sim = small_world(n_pairs=0, seed=17, single_frac=0.05)
    bam = sim.bam(0, 8000)
    head, recs = _split_bam(bam)
    # a few irregular records: an orphan whose mate is missing, a pair on two chromosomes, a pair with an unmapped mate
    extra = records_to_bam(ann.chr_names, ann.chr_lens, [
        dict(ref_id=0, pos=5000, flag=0x41, cigar="50M", name="orphan"),
        dict(ref_id=0, pos=6000, flag=0x41, cigar="50M", name="split"), dict(ref_id=1, pos=6000, flag=0x81, cigar="50M", name="split"),
        dict(ref_id=0, pos=7000, flag=0x41, cigar="50M", name="half"), dict(ref_id=0, pos=7000, flag=0x81 | 0x4, cigar="50M", name="half")])
    recs_x = _split_bam(extra)[1]
    collated = head + b"".join(recs + recs_x[1:])            # the orphan last: an unpaired trailing record
    r0 = engine.Reader()
    r0.feed(collated + recs_x[0])
    r0.finish()
    want, want_stats = _fragment_multiset(r0.take_batch()), r0.stats()
    assert want_stats["n_error"] == 3 and want_stats["n_skipped"] == 2 and want_stats["n_pairs"] > 7000

    def key(rec):
        ref_id, pos = struct.unpack_from("<ii", rec, 4)
        return (ref_id if ref_id >= 0 else 1 << 30, pos)
    srt = sorted(recs + recs_x, key=key)
    assert srt != recs + recs_x
    r1 = engine.Reader(engine.default_params(mate_join=1))
    data = head + b"".join(srt)
    for i in range(0, len(data), 100003):            # odd chunking
        r1.feed(data[i:i + 100003])
    r1.finish()
    assert _fragment_multiset(r1.take_batch()) == want
    assert r1.stats() == want_stats
    # read the upstream way, the sorted file is mis-paired (that is D-1)
    r2 = engine.Reader()
    r2.feed(data)
    r2.finish()
    assert _fragment_multiset(r2.take_batch()) != want


def test_threaded_bgzf_decode_is_identical_and_reports_the_same_errors():
    """irf_reader_set_threads: BGZF members inflated by a worker pool while the parser runs — same batch, same counters,
    same errors as the single-threaded reader, for large and small feeds."""
    ann, sim = small_world(n_pairs=0, seed=41, read_len=100, with_seq=True)
    bam = sim.bam(0, 12000)
    z = bgzf_compress(bam, block=0x2000)                      # many members per feed
    r1 = engine.Reader()
    r1.feed(z, bgzf=True)
    r1.finish()
    want, want_stats = r1.take_batch(), r1.stats()
    for threads, piece in ((4, len(z)), (3, 300_000), (8, 4097)):
        r = engine.Reader(threads=threads)
        for i in range(0, len(z), piece):
            r.feed(z[i:i + piece], bgzf=True)
        r.finish()
        assert r.take_batch().equals(want) and r.stats() == want_stats, (threads, piece)
    # a corrupted member in the middle: same sticky error either way, nothing after it is parsed
    bad = bytearray(z)
    bad[len(z) // 2] ^= 0x55
    msgs, after = [], []
    for threads, where in ((1, len(z) // 2), (4, len(z) // 2), (1, len(z) * 7 // 8), (4, len(z) * 7 // 8), (2, len(z) * 7 // 8)):
        bad = bytearray(z)
        bad[where] ^= 0x55
        r = engine.Reader(threads=threads)
        with pytest.raises(engine.IrfError) as ei:
            r.feed(bytes(bad), bgzf=True)
            r.finish()
        msgs.append(str(ei.value))
        after.append((where, r.stats(), r.take_batch()))
    assert msgs[0] == msgs[1] and "BGZF" in msgs[0]
    # what the reader holds after the error (the records in front of the damaged member) does not depend on the thread count
    for a in after:
        for b in after:
            if a[0] == b[0]:
                assert a[1] == b[1] and a[2].equals(b[2])
    assert after[2][1]["n_records"] > after[0][1]["n_records"] > 0
    with pytest.raises(engine.IrfError):
        engine.Reader(threads=0)
