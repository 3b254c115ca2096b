"""Committed fixtures (tests/golden/, written by tests/golden/make_golden.py).

kat_vectors.json holds the hand-derived known answers of SURVEY.md App. C; world_*/ hold seeded synthetic samples with
the output files frozen from the CPU oracle.  None of it comes from upstream IRFinder (no source, no fixtures in
/root/reference): parity unpinned.  CPU tests pin the oracle to the fixtures; the GPU tests pin the HIP path (through
the C ABI) to the same fixtures without consulting the oracle at all.
"""
import gzip
import json
import os

import numpy as np
import pytest

from irfinder_amd.synth.bam import records_to_bam

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_NAMES = ("ref-cover.bed", "ref-sj.ref", "ref-read-continues.ref", "ref-ROI.bed")
KAT = json.load(open(os.path.join(GOLD, "kat_vectors.json")))
WORLDS = sorted(d[len("world_"):] for d in os.listdir(GOLD) if d.startswith("world_"))
WARN = {"-": 0, "LowCover": 1, "LowSplicing": 2, "MinorIsoform": 3, "NonUniformIntronCover": 4}


def world(tag):
    d = os.path.join(GOLD, "world_" + tag)
    refs = tuple(open(os.path.join(d, n)).read() for n in REF_NAMES)
    bam = gzip.open(os.path.join(d, "sample.bam.raw.gz")).read()
    exp = {f: open(os.path.join(d, "expected", f), "rb").read() for f in sorted(os.listdir(os.path.join(d, "expected")))}
    return refs, bam, exp


def kat_bam(records):
    e = KAT["engine"]
    recs = [dict(ref_id=0, pos=p, flag=f, cigar=c, name="r%d" % (i // 2 if f & 1 else 1000 + i)) for i, (p, f, c) in enumerate(records)]
    return records_to_bam(e["chr"], e["len"], recs)


def check_engine_case(case, r, depth):
    """r: results dict of either backend; depth(chr, a, b, plane) -> per-base depth."""
    if "junc" in case:
        got = [[int(l), int(rr), int(n), int(p)] for l, rr, n, p in zip(r["junc_left"], r["junc_right"], r["junc_cnt"][0], r["junc_cnt"][1])]
        assert got == case["junc"], case["name"]
    if "sp_pos" in case:
        assert r["sp_pos"].tolist() == case["sp_pos"] and r["sp_cnt"][1].tolist() == case["sp_pos_cnt"] and r["sp_cnt"][0].sum() == 0
    if "roi_cnt" in case:
        assert r["roi_cnt"].tolist() == case["roi_cnt"] and r["chr_cnt"].tolist() == case["chr_cnt"]
    if "splice" in case:
        t = r["nondir"]
        assert [int(t["splice_left"][0]), int(t["splice_right"][0]), int(t["splice_exact"][0])] == case["splice"]
        num, den = case["intron_depth"]
        d = num / den
        assert t["intron_depth"][0] == d and t["irratio"][0] == d / (d + case["irratio_den_add"])
        assert int(t["warning"][0]) == WARN[case["warning"]]
        assert depth(0, 90, 210, 1).tolist() == case["depth_pos_90_210"]
        assert depth(0, 90, 210, 0).sum() == 0


# ---------------------------------------------------------------- CPU: the oracle against the fixtures
def test_oracle_cigar_and_pair_vectors():
    import orc
    for v in KAT["cigar2block"]:
        assert orc.cigar2block(v["cigar"]) == (v["starts"], v["lens"], v["genome_len"])
    e = KAT["engine"]
    for v in KAT["pairs"]:
        bam = records_to_bam(e["chr"], e["len"], [dict(ref_id=0, pos=v["pos1"], flag=0x61, cigar=v["cigar1"], name="p"),
                                                  dict(ref_id=0, pos=v["pos2"], flag=0x91, cigar=v["cigar2"], name="p")])
        b, st, rc = orc.bam_to_batch(bam)
        assert rc == 0 and b.n_frag == 1 and st["n_pairs_" + v["class"]] == 1
        got = [[int(b.blk_start[k]), int(b.blk_start[k] + b.blk_len[k]), int(b.blk_read[k])] for k in range(b.n_blk)]
        assert got == v["blocks"], v


def test_oracle_stats_vectors():
    import orc
    for v in KAT["stats"]:
        s = orc.stats_from_values(v["values"])
        assert s["coverage"] == v["coverage"][0] / v["coverage"][1]
        assert s["trimmed"] == v["trimmed"][0] / v["trimmed"][1]
        assert (s["p25"], s["p50"], s["p75"]) == (v["p25"], v["p50"], v["p75"])


@pytest.mark.parametrize("case", KAT["engine"]["cases"], ids=lambda c: c["name"])
def test_oracle_engine_vectors(case):
    import orc
    e = KAT["engine"]
    o = orc.Oracle(tuple(e["refs"][n] for n in REF_NAMES), orc.default_params(keep_depth=1))
    o.process_bam(kat_bam(case["records"]))
    o.finalize()
    check_engine_case(case, o.results(), o.depth)


@pytest.mark.parametrize("tag", WORLDS)
def test_oracle_reproduces_world_outputs(tag, tmp_path):
    import orc
    refs, bam, exp = world(tag)
    o = orc.Oracle(refs)
    o.process_bam(bam)
    o.finalize()
    o.write_outputs(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted(exp)
    for f, want in exp.items():
        assert (tmp_path / f).read_bytes() == want, f


def test_fixture_worlds_are_not_trivial():
    for tag in WORLDS:
        refs, bam, exp = world(tag)
        ir = exp["IRFinder-IR-nondir.txt"].decode().splitlines()
        assert len(ir) > 10 and sum(1 for l in ir[1:] if float(l.split("\t")[8]) > 0) >= 5
        jc = exp["IRFinder-JuncCount.txt"].decode().splitlines()
        assert sum(int(l.split("\t")[4]) for l in jc) > 500
    assert "IRFinder-IR-dir.txt" in world("stranded")[2] and "IRFinder-IR-dir.txt" not in world("unstranded")[2]


# ---------------------------------------------------------------- GPU: the HIP path against the fixtures (no oracle)
def product_run(refs, bam, **over):
    from irfinder_amd import engine
    p = engine.default_params(**over)
    r = engine.Reader(p)
    r.feed(bam)
    r.finish()
    names, lens = r.header()
    e = engine.Engine(engine.RefTables.from_texts(refs), p)
    e.chr_map_update(names, lens)
    e.process(r.take_batch())
    e.finalize()
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("tag", WORLDS)
def test_hip_reproduces_world_outputs(tag, tmp_path):
    refs, bam, exp = world(tag)
    e = product_run(refs, bam)
    e.write_outputs(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted(exp)
    for f, want in exp.items():
        assert (tmp_path / f).read_bytes() == want, f


@pytest.mark.gpu
@pytest.mark.parametrize("case", KAT["engine"]["cases"], ids=lambda c: c["name"])
def test_hip_engine_vectors(case):
    e = KAT["engine"]
    eng = product_run(tuple(e["refs"][n] for n in REF_NAMES), kat_bam(case["records"]), keep_depth=1)
    check_engine_case(case, eng.results(), eng.depth)


def test_product_reader_pair_vectors():            # host code of libirfhip.so: runs without a GPU
    from irfinder_amd import engine
    e = KAT["engine"]
    for v in KAT["pairs"]:
        bam = records_to_bam(e["chr"], e["len"], [dict(ref_id=0, pos=v["pos1"], flag=0x61, cigar=v["cigar1"], name="p"),
                                                  dict(ref_id=0, pos=v["pos2"], flag=0x91, cigar=v["cigar2"], name="p")])
        r = engine.Reader()
        r.feed(bam)
        r.finish()
        b = r.take_batch()
        got = [[int(b.blk_start[k]), int(b.blk_start[k] + b.blk_len[k]), int(b.blk_read[k])] for k in range(b.n_blk)]
        assert got == v["blocks"], v
