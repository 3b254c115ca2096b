"""Known-answer tests of SURVEY.md Appendix C on the CPU oracle.

These are hand-derived from the recalled spec (App. B); they pin the oracle to
the spec, not to upstream IRFinder (parity unpinned — SURVEY §0.3).
"""
import numpy as np
import pytest

import orc
from irfinder_amd.synth.bam import records_to_bam

CHR = ["chrA"]
LEN = [100000]


def pair(pos1, cig1, pos2, cig2, flag1=0x1 | 0x40 | 0x20, flag2=0x1 | 0x80 | 0x10, name="p"):
    """read 1 forward at pos1, read 2 reverse at pos2 -> fragment direction + (read 1 forward)."""
    return [dict(ref_id=0, pos=pos1, flag=flag1, cigar=cig1, name=name),
            dict(ref_id=0, pos=pos2, flag=flag2, cigar=cig2, name=name)]


def frags(records, **over):
    bam = records_to_bam(CHR, LEN, records)
    batch, stats, st = orc.bam_to_batch(bam, orc.default_params(**over))
    assert st == 0
    return batch, stats


def blocks_of(batch, i):
    a, b = int(batch.frag_blk_off[i]), int(batch.frag_blk_off[i + 1])
    return [(int(batch.blk_start[k]), int(batch.blk_start[k] + batch.blk_len[k]), int(batch.blk_read[k])) for k in range(a, b)]


# ---- KAT-1 cigar2block -------------------------------------------------
def test_kat1_cigar2block():
    assert orc.cigar2block("50M1000N50M") == ([0, 1050], [50, 50], 1100)
    assert orc.cigar2block("5S45M2D5M100N3I50M") == ([0, 152], [52, 50], 202)
    assert orc.cigar2block("10M100N200N10M") == ([0, 310], [10, 10], 320)
    assert orc.cigar2block("10S5I") == ([], [], 0)
    assert orc.cigar2block("3=2X5M") == ([0], [10], 10)


# ---- KAT-2 pair classes --------------------------------------------------
def test_kat2_pair_classes():
    b, st = frags(pair(100, "50M", 130, "50M"))
    assert b.n_frag == 1 and blocks_of(b, 0) == [(100, 180, 0)] and b.frag_flags[0] == 1 and b.frag_end[0] == 180
    assert st["n_pairs_intersect"] == 1
    b, st = frags(pair(100, "50M", 150, "50M"))          # abutting: 150 < 150 false => Intersect
    assert blocks_of(b, 0) == [(100, 200, 0)] and st["n_pairs_intersect"] == 1
    b, st = frags(pair(100, "50M", 151, "50M"))          # Long
    assert blocks_of(b, 0) == [(100, 150, 0), (151, 201, 1)] and b.frag_flags[0] == 3 and st["n_pairs_long"] == 1
    assert b.frag_end[0] == 201
    b, st = frags(pair(100, "100M", 120, "50M"))         # Short: read 1 blocks only
    assert blocks_of(b, 0) == [(100, 200, 0)] and st["n_pairs_short"] == 1
    b, st = frags(pair(100, "30M500N20M", 640, "40M"))   # spliced merge
    assert blocks_of(b, 0) == [(100, 130, 0), (630, 680, 0)] and b.frag_end[0] == 680
    # OQ-1 switches
    b, st = frags(pair(100, "50M", 150, "50M"), pair_long_inclusive=1)
    assert st["n_pairs_long"] == 1 and b.frag_flags[0] & 2
    # pos2 inside an N gap of read 1 -> two reads, originals (OQ-3)
    b, st = frags(pair(100, "30M500N20M", 300, "400M"))
    assert blocks_of(b, 0) == [(100, 130, 0), (630, 650, 0), (300, 700, 1)] and st["n_pairs_long"] == 1


def test_pair_order_and_direction():
    # records arrive right-mate first; left record is read 2, so direction = !(flag & 0x20) of the left record
    recs = [dict(ref_id=0, pos=500, flag=0x1 | 0x40 | 0x10, cigar="50M", name="q"),     # read 1, reverse, right
            dict(ref_id=0, pos=100, flag=0x1 | 0x80 | 0x20, cigar="50M", name="q")]     # read 2, forward, left
    b, st = frags(recs)
    assert blocks_of(b, 0) == [(100, 150, 0), (500, 550, 1)]
    assert (b.frag_flags[0] & 1) == 0            # read 1 is reverse => fragment from - strand
    # single-end
    b, st = frags([dict(ref_id=0, pos=10, flag=0x10, cigar="20M"), dict(ref_id=0, pos=10, flag=0x0, cigar="20M"),
                   dict(ref_id=0, pos=10, flag=0x4, cigar="20M")])
    assert b.n_frag == 2 and list(b.frag_flags) == [0, 1] and st["n_singles"] == 2 and st["n_skipped"] == 1


def test_reader_error_paths():
    recs = pair(100, "50M", 300, "50M")
    recs[1]["ref_id"] = 0
    two = CHR + ["chrB"], LEN + [5000]
    bam = records_to_bam(two[0], two[1], [dict(ref_id=0, pos=1, flag=0x41, cigar="10M"), dict(ref_id=1, pos=1, flag=0x81, cigar="10M"),
                                         dict(ref_id=0, pos=1, flag=0x41 | 0x4, cigar="10M"), dict(ref_id=0, pos=1, flag=0x81, cigar="10M"),
                                         dict(ref_id=0, pos=1, flag=0x41, cigar="10M")])
    b, st, rc = orc.bam_to_batch(bam)
    assert rc == 0 and b.n_frag == 0
    assert st["n_error"] == 3 and st["n_skipped"] == 2 and st["n_records"] == 5
    # truncated stream
    b, st, rc = orc.bam_to_batch(bam[:-3])
    assert rc != 0


# ---- engine-level KATs ------------------------------------------------------
COVER = "chrA\t90\t210\tG1/ENSG1/+/1/90/210/clean\t0\t+\t90\t210\t255,0,0\t1\t120\t0\n"
SJ = "chrA\t130\t630\t+\n"
SP = "chrA\t105\t+\nchrA\t106\t+\nchrA\t175\t+\nchrA\t176\t+\n"
ROI = "chrA\t0\t1000\tROI0\nchrA\t2000\t3000\tROI1\n"


def run(records, refs=(COVER, SJ, SP, ROI), **over):
    o = orc.Oracle(refs, orc.default_params(**over))
    o.process_bam(records_to_bam(CHR, LEN, records))
    o.finalize()
    return o, o.results()


def test_kat3_junction():
    o, r = run(pair(100, "30M500N20M", 640, "40M"))
    assert r["junc_left"].tolist() == [130] and r["junc_right"].tolist() == [630]
    assert r["junc_cnt"][:, 0].tolist() == [0, 1] and r["junc_refdir"].tolist() == [2]
    # first block 4M: no count, but the known junction row stays (zero)
    o, r = run(pair(126, "4M500N20M", 640, "40M"))
    assert r["junc_cnt"].sum() == 0 and len(r["junc_left"]) == 1
    # Long pair, both mates carry the junction -> twice; Intersect-merged -> once
    o, r = run(pair(100, "30M500N10M", 641, "40M") + pair(110, "20M500N20M", 120, "10M500N30M"))
    i = r["junc_left"].tolist().index(130)
    assert r["junc_cnt"][1, i] == 2
    o, r = run(pair(100, "30M500N10M", 110, "20M500N30M") * 1 + pair(100, "30M500N20M", 1000, "10M"))
    i = r["junc_left"].tolist().index(130)
    assert r["junc_cnt"][1, i] == 2          # one from each fragment


def kat4_records():
    return pair(100, "50M", 130, "50M") + pair(100, "50M", 150, "50M") + pair(100, "100M", 120, "50M")


def test_kat4_depth_and_kat5_stats():
    o, r = run(kat4_records(), keep_depth=1)
    d = o.depth(0, 90, 210, 1)
    exp = np.array([0] * 10 + [3] * 80 + [2] * 20 + [0] * 10, np.uint32)
    assert np.array_equal(d, exp)
    assert np.array_equal(o.depth(0, 90, 210, 2), exp) and o.depth(0, 90, 210, 0).sum() == 0
    t = r["nondir"]
    assert t["coverage"][0] == pytest.approx(100 / 120, rel=1e-15)
    assert t["intron_depth"][0] == pytest.approx(140 / 48, rel=1e-15)       # 2.916666...
    assert t["p50"][0] == 3.0 and t["p25"][0] == 2.0 and t["p75"][0] == 3.0
    s = orc.stats_from_values(exp)
    assert s["mean"] == pytest.approx(280 / 120) and s["trimmed"] == 140 / 48


def test_kat5_percentile_interpolation():
    # N=4 values 0,10,20,30: p25 rank 1.25 -> 0+0.25*10 ; p50 rank 2.5 -> 15 ; p75 rank 3.75 -> 27.5
    s = orc.stats_from_values([30, 0, 20, 10])
    assert (s["p25"], s["p50"], s["p75"]) == (2.5, 15.0, 27.5)
    s = orc.stats_from_values([30, 0, 20, 10], orc.default_params(pct_interpolate=0))
    assert (s["p25"], s["p50"], s["p75"]) == (0.0, 10.0, 20.0)
    s = orc.stats_from_values([7])
    assert (s["p25"], s["p50"], s["p75"], s["trimmed"], s["coverage"]) == (7.0, 7.0, 7.0, 7.0, 1.0)
    s = orc.stats_from_values([])
    assert s["trimmed"] == 0.0 and s["p50"] == 0.0 and s["coverage"] == 0.0


def test_kat6_spanspoint():
    o, r = run([dict(ref_id=0, pos=100, flag=0, cigar="80M")])
    assert r["sp_pos"].tolist() == [105, 106, 175, 176]
    assert r["sp_cnt"][1].tolist() == [0, 1, 1, 0] and r["sp_cnt"][0].sum() == 0
    o, r = run([dict(ref_id=0, pos=100, flag=0, cigar="80M")], sp_inclusive=1)
    assert r["sp_cnt"][1].tolist() == [1, 1, 1, 1]
    o, r = run([dict(ref_id=0, pos=100, flag=0x10, cigar="9M"), dict(ref_id=0, pos=170, flag=0x10, cigar="9M")])
    assert r["sp_cnt"].sum() == 0


def test_kat7_roi():
    recs = pair(100, "50M", 400, "50M") + pair(900, "100M", 1000, "100M") + [dict(ref_id=0, pos=2000, flag=0x10, cigar="1000M")]
    o, r = run(recs)
    assert r["roi_cnt"].tolist() == [[0, 1], [1, 0]]
    assert r["chr_cnt"].tolist() == [[1, 2]]


def test_kat8_irratio_and_warning():
    # KAT-4 depth (IntronDepth 2.91666) + splice left 7 / right 5 / exact 5 at intron (90,210)
    recs = kat4_records()
    for k in range(5):
        recs += [dict(ref_id=0, pos=40, flag=0, cigar="50M120N30M", name="e%d" % k)]        # exact 90..210
    for k in range(2):
        recs += [dict(ref_id=0, pos=40, flag=0, cigar="50M400N30M", name="l%d" % k)]        # left only (90..490)
    o, r = run(recs)
    t = r["nondir"]
    assert (t["splice_left"][0], t["splice_right"][0], t["splice_exact"][0]) == (7, 5, 5)
    d = 140 / 48
    assert t["intron_depth"][0] == d
    assert t["irratio"][0] == d / (d + 7.0)
    assert t["warning"][0] == 1          # 5 + 2.9 < 10 -> LowCover


OQ12_REFS = ("chrA\t1000\t3000\tG2/ENSG2/+/1/1000/3000/clean\t0\t+\t1000\t3000\t255,0,0\t1\t2000\t0\n", "chrA\t1000\t3000\t+\n",
             "chrA\t1000\t+\nchrA\t3000\t+\n", "chrA\t50000\t60000\tROI0\n")


def oq12_records():
    """IntronDepth 10, largest flank measure 16 (first-50 median; ExonToIntronReadsLeft 6), SpliceExact 5."""
    recs = [dict(ref_id=0, pos=1000, flag=0, cigar="2000M", name="d%d" % k) for k in range(10)]      # depth 10 over the intron
    recs += [dict(ref_id=0, pos=950, flag=0, cigar="110M", name="x%d" % k) for k in range(6)]        # +6 on [1000,1060), span point 1000
    recs += [dict(ref_id=0, pos=900, flag=0, cigar="100M2000N100M", name="s%d" % k) for k in range(5)]
    return recs


def test_oq12_nonuniform_single_tolerance_vs_two_conditions():
    """The KAT row that tells the two recalled forms of the NonUniformIntronCover test apart: depth 10, max flank 16.
    warn_nonuniform_and = 0 (default): 16 > 10*1.5 + 4 = 19 is false -> '-';  = 1: 16 > 10 + 4 and 16 > 10*1.5 -> warning."""
    o, r = run(oq12_records(), refs=OQ12_REFS)
    t = r["nondir"]
    assert t["intron_depth"][0] == 10.0 and t["first50"][0] == 16.0 and t["last50"][0] == 10.0
    assert (t["sp_left"][0], t["sp_right"][0], t["splice_exact"][0]) == (6, 0, 5)
    assert t["warning"][0] == 0
    o, r = run(oq12_records(), refs=OQ12_REFS, warn_nonuniform_and=1)
    assert r["nondir"]["warning"][0] == 4
    assert r["nondir"]["intron_depth"][0] == 10.0


def test_oq3_gap_pair_summary_class():
    """An overlapping pair whose read 2 starts in an N gap of read 1: two reads either way; the summary class is a switch."""
    recs = pair(100, "30M500N30M", 300, "40M")          # e1 = 660 > pos2 = 300 (not Long), e1 > e2 = 340?  no: Short needs e1 >= e2
    b, st = frags(recs)
    assert st["n_pairs_short"] == 1                     # (read 2 inside read 1's span: Short, read-1 blocks only)
    recs = pair(100, "30M500N30M", 300, "400M")         # pos2 in the gap, e2 = 700 > e1 = 660: neither Long nor Short
    b, st = frags(recs)
    assert (st["n_pairs_long"], st["n_pairs_intersect"]) == (1, 0) and len(blocks_of(b, 0)) == 3 and int(b.frag_flags[0]) & 2
    b2, st2 = frags(recs, gap_pair_is_intersect=1)
    assert (st2["n_pairs_long"], st2["n_pairs_intersect"]) == (0, 1) and b2.equals(b)


def test_outputs_written(tmp_path):
    o, r = run(kat4_records() + [dict(ref_id=0, pos=40, flag=0, cigar="50M120N30M")])
    o.write_outputs(str(tmp_path))
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["IRFinder-ChrCoverage.txt", "IRFinder-IR-nondir.txt", "IRFinder-JuncCount.txt", "IRFinder-ROI.txt",
                     "IRFinder-SpansPoint.txt"]
    ir = (tmp_path / "IRFinder-IR-nondir.txt").read_text().splitlines()
    assert ir[0].split("\t")[:4] == ["Chr", "Start", "End", "Name"] and len(ir[0].split("\t")) == 21
    row = ir[1].split("\t")
    assert row[:3] == ["chrA", "90", "210"] and row[6] == "0" and row[7] == "0.833333" and row[8] == "2.91667"
    assert (tmp_path / "IRFinder-ChrCoverage.txt").read_text() == "chrA\t4\t0\t4\n"
    jc = (tmp_path / "IRFinder-JuncCount.txt").read_text().splitlines()
    assert jc == ["chrA\t90\t210\t.\t1\t1\t0", "chrA\t130\t630\t+\t0\t0\t0"]


def test_directionality_vote():
    recs = []
    for k in range(12):
        recs += [dict(ref_id=0, pos=100, flag=0, cigar="30M500N20M", name="d%d" % k)]   # + strand on a '+' junction
    o, r = run(recs)
    assert r["directionality"] == 1 and r["has_dir"]
    recs = [dict(ref_id=0, pos=100, flag=0x10, cigar="30M500N20M", name="d%d" % k) for k in range(12)]
    o, r = run(recs)
    assert r["directionality"] == -1
    recs += [dict(ref_id=0, pos=100, flag=0, cigar="30M500N20M", name="e%d" % k) for k in range(6)]
    o, r = run(recs)
    assert r["directionality"] == 0 and not r["has_dir"]
