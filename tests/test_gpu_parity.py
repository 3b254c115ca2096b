"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.

Bar (north_star): integer counts, junction tables, depths bit-exact; fp64 columns within 1e-9 relative — in fact
asserted bit-exact here (rel=0), because the same integer inputs go through the same IEEE operations.
"Oracle" = this repo's CPU restatement of the recalled spec (parity unpinned vs upstream IRFinder).
"""
import filecmp
import os

import numpy as np
import pytest

import orc
from irfinder_amd import _abi, engine
from irfinder_amd._abi import Batch
from irfinder_amd.synth.bam import records_to_bam
from util import assert_results_equal, small_world

pytestmark = pytest.mark.gpu


def read_batch(bam, **over):
    r = engine.Reader(engine.default_params(**over))
    r.feed(bam)
    r.finish()
    names, lens = r.header()
    return names, lens, r.take_batch()


def run_gpu(ref_texts, names, lens, batches, owned=None, **over):
    ref = engine.RefTables.from_texts(ref_texts)
    e = engine.Engine(ref, engine.default_params(**over))
    e.chr_map_update(names, lens, owned)
    for b in batches:
        e.process(b)
    e.finalize()
    return e, e.results()


def run_orc(ref_texts, names, lens, batches, **over):
    o = orc.Oracle(ref_texts, orc.default_params(**over))
    o.chr_map_update(names, lens)
    for b in batches:
        o.process_batch(b)
    o.finalize()
    return o, o.results()


def split(batch, k):
    n = batch.n_frag
    cuts = np.linspace(0, n, k + 1).astype(int)
    return [batch.select(np.arange(cuts[i], cuts[i + 1])) for i in range(k)]


@pytest.mark.parametrize("stranded", [False, True])
def test_synthetic_world_matches_oracle(stranded, tmp_path):
    ann, sim = small_world(stranded=stranded, seed=3)
    names, lens, batch = read_batch(sim.bam(0, 40000))
    names = names + ["chrUn_extra"]            # a BAM reference no ref file mentions
    lens = lens + [5000]
    e, g = run_gpu(ann.ref_texts(), names, lens, split(batch, 3))
    o, c = run_orc(ann.ref_texts(), names, lens, [batch])
    assert_results_equal(g, c, rel=0.0, what="stranded=%s" % stranded)
    assert g["has_dir"] == stranded
    assert g["junc_cnt"].sum() > 1000 and g["sp_cnt"].sum() > 100 and g["roi_cnt"].sum() > 10
    assert (g["nondir"]["intron_depth"] > 0).sum() > 10
    # the six text files are byte-identical
    gd, cd = tmp_path / "gpu", tmp_path / "cpu"
    gd.mkdir(); cd.mkdir()
    e.write_outputs(str(gd)); o.write_outputs(str(cd))
    files = sorted(os.listdir(cd))
    assert files == sorted(os.listdir(gd)) and len(files) == (6 if stranded else 5)
    for f in files:
        assert filecmp.cmp(gd / f, cd / f, shallow=False), f


def test_order_and_batch_invariance_P1_P2():
    ann, sim = small_world(seed=4)
    names, lens, batch = read_batch(sim.bam(0, 15000))
    e, base = run_gpu(ann.ref_texts(), names, lens, [batch])
    perm = np.random.default_rng(1).permutation(batch.n_frag)
    e2, shuf = run_gpu(ann.ref_texts(), names, lens, split(batch.select(perm), 7))
    assert_results_equal(base, shuf, rel=0.0, what="permuted+split")


@pytest.mark.parametrize("how", ["device", "host"])
def test_input_in_any_order_is_binned_on_the_device(how):
    """irf_set_input_order(IRF_ORDER_ANY): shuffled fragments, batches of uneven size (one of them below the threshold the
    reorder starts at), device-resident and host batches: the oracle's tables (P1: every output is a commutative sum)."""
    import torch
    ann, sim = small_world(seed=6)
    names, lens, batch = read_batch(sim.bam(0, 30000))
    rng = np.random.default_rng(5)
    shuffled = batch.select(rng.permutation(batch.n_frag))
    parts = [shuffled.select(np.arange(a, b)) for a, b in ((0, 700), (700, 19000), (19000, shuffled.n_frag))]
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref)
    e.chr_map_update(names, lens)
    e.set_input_order(_abi.ORDER_ANY)
    keep = []
    for p in parts:
        if how == "device":
            dten = {n: torch.from_numpy(np.ascontiguousarray(getattr(p, n))).to("cuda:0") for n, _ in Batch.FIELDS}
            keep.append(dten)
            e.submit(engine.device_batch(dten, p.n_frag, p.n_blk))
        else:
            e.process(p)
    e.finalize()
    o, c = run_orc(ann.ref_texts(), names, lens, [batch])
    assert_results_equal(e.results(), c, rel=0.0, what="ORDER_ANY " + how)
    with pytest.raises(engine.IrfError):
        e.set_input_order(7)


def test_nondir_is_pos_plus_neg_P4_and_junction_sums_P6():
    ann, sim = small_world(seed=9)
    names, lens, batch = read_batch(sim.bam(0, 15000))
    e, g = run_gpu(ann.ref_texts(), names, lens, [batch], keep_depth=1)
    for c in range(len(names)):
        d0, d1, d2 = (e.depth(c, 0, lens[c], p) for p in (0, 1, 2))
        assert np.array_equal(d0 + d1, d2)
    assert g["chr_cnt"].sum() == batch.n_frag == g["n_fragments"]
    assert g["junc_cnt"].sum() == g["n_junction_events"]


def test_full_and_compact_planes_agree():
    """keep_depth=1 lays the planes out over the whole genome (FragmentsMap), the default keeps only the buckets with
    measured intron bases; every result must be identical, and depth is only served in the first mode."""
    ann, sim = small_world(seed=13, stranded=True)
    names, lens, batch = read_batch(sim.bam(0, 20000))
    e1, full = run_gpu(ann.ref_texts(), names, lens, split(batch, 2), keep_depth=1)
    e0, compact = run_gpu(ann.ref_texts(), names, lens, split(batch, 3))
    assert_results_equal(full, compact, rel=0.0, what="full vs compact planes")
    with pytest.raises(engine.IrfError):
        e0.depth(0, 0, 100, 2)


def test_depth_matches_fragmentsmap():
    ann, sim = small_world(seed=12)
    names, lens, batch = read_batch(sim.bam(0, 10000))
    e, g = run_gpu(ann.ref_texts(), names, lens, [batch], keep_depth=1)
    o, c = run_orc(ann.ref_texts(), names, lens, [batch], keep_depth=1)
    for ch in range(len(names)):
        for p in (0, 1, 2):
            assert np.array_equal(e.depth(ch, 0, lens[ch], p), o.depth(ch, 0, lens[ch], p)), (ch, p)


@pytest.mark.parametrize("over", [dict(flank_use_measured=0), dict(sp_inclusive=1), dict(pct_interpolate=0),
                                  dict(trim_centre_pct=100), dict(trim_centre_pct=0), dict(junc_min_flank=1),
                                  dict(flank_offset=0, flank_len=200), dict(dir_ratio=1), dict(warn_nonuniform_and=1)])
def test_oq_switches(over):
    ann, sim = small_world(seed=21, stranded=True)
    names, lens, batch = read_batch(sim.bam(0, 12000))
    e, g = run_gpu(ann.ref_texts(), names, lens, split(batch, 2), **over)
    o, c = run_orc(ann.ref_texts(), names, lens, [batch], **over)
    assert_results_equal(g, c, rel=0.0, what=str(over))


@pytest.mark.parametrize("order", [0, 1])
def test_device_batch_with_malformed_csr_offsets_is_reported(order):
    """frag_blk_off of a device-resident batch (a third-party producer) is checked by the count kernel itself: the host
    cannot read it.  Not non-decreasing / beyond n_blk / not spanning [0, n_blk] -> irf_finalize fails with IRF_E_ARG."""
    import torch
    from irfinder_amd._abi import Batch as B_
    ann, sim = small_world(seed=2)
    names, lens, batch = read_batch(sim.bam(0, 3000))
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref)
    e.chr_map_update(names, lens)
    e.set_input_order(order)          # (ORDER_ANY: the reorder step checks the offsets before it follows them)

    def run_with(off):
        e.reset()
        t = {n: torch.from_numpy(np.ascontiguousarray(off if n == "frag_blk_off" else getattr(batch, n))).cuda() for n, _ in B_.FIELDS}
        e.submit(engine.device_batch(t, batch.n_frag, batch.n_blk))
        e.sync()
        e.finalize()

    good = batch.frag_blk_off.copy()
    run_with(good)                                            # the unmodified offsets pass
    o, c = run_orc(ann.ref_texts(), names, lens, [batch])
    assert_results_equal(e.results(), c, rel=0.0, what="device batch")
    for name, edit in (("decreasing inside a group", lambda a: a.__setitem__(5, a[4] - 1 if a[4] else a[6] + 1)),
                       ("decreasing across groups", lambda a: a.__setitem__(1024, a[1023] - 1)),
                       ("beyond n_blk", lambda a: a.__setitem__(2000, batch.n_blk + 7)),
                       ("does not start at 0", lambda a: a.__setitem__(0, 1)),
                       ("does not end at n_blk", lambda a: a.__setitem__(batch.n_frag, batch.n_blk - 1)),
                       # every single range inside [0, n_blk] but their sum far beyond it (ORDER_ANY: the reordered blocks
                       # must not be written past the arrays; the bounds were sized from n_blk)
                       ("drops to 0 in the middle", lambda a: a.__setitem__(5, 0)),
                       ("alternating 0 / n_blk", lambda a: a.__setitem__(slice(1, batch.n_frag, 2), batch.n_blk) or a.__setitem__(slice(2, batch.n_frag, 2), 0))):
        bad = good.copy()
        edit(bad)
        with pytest.raises(engine.IrfError) as ei:
            run_with(bad)
        assert ei.value.code == -1, name


def test_oq12_kat_row_on_the_gpu():
    """depth 10, largest flank measure 16: '-' under the single mixed tolerance, NonUniformIntronCover under the
    two-condition form (tests/test_oracle_kat.py: the same vectors on the oracle)."""
    from test_oracle_kat import CHR, LEN, OQ12_REFS, oq12_records
    names, lens, batch = read_batch(records_to_bam(CHR, LEN, oq12_records()))
    for over, want in (({}, 0), (dict(warn_nonuniform_and=1), 4)):
        e, g = run_gpu(OQ12_REFS, names, lens, [batch], **over)
        o, c = run_orc(OQ12_REFS, names, lens, [batch], **over)
        assert_results_equal(g, c, rel=0.0, what=str(over))
        assert g["nondir"]["warning"][0] == want and g["nondir"]["intron_depth"][0] == 10.0 and g["nondir"]["first50"][0] == 16.0


def test_novel_junctions_and_many_events_per_thread():
    """Half of ref-sj.ref removed (those junctions become NOVEL: event buffer + radix sort-reduce), plus reads with
    100 junctions each (more events per thread than the 64-bit novel mask holds), alternating known / novel."""
    ann, sim = small_world(seed=14, stranded=True)
    names, lens, batch = read_batch(sim.bam(0, 30000))
    sj = "".join(l for i, l in enumerate(ann.sj_ref.splitlines(True)) if i % 2 == 0)
    # a comb of 101 blocks on chrT1: junction k spans [1000+16k+6, 1000+16k+16)
    comb_known = "".join("chrT1\t%d\t%d\t+\n" % (1000 + 16 * k + 6, 1000 + 16 * k + 16) for k in range(0, 100, 2))
    recs = [dict(ref_id=0, pos=1000, flag=0x10 if i % 2 else 0, cigar="6M10N" * 100 + "6M") for i in range(700)]
    n2, l2, comb = read_batch(records_to_bam(names, lens, recs))
    refs = (ann.cover_bed, sj + comb_known, ann.read_continues_ref, ann.roi_bed)
    e, g = run_gpu(refs, names, lens, split(batch, 2) + [comb])
    o, c = run_orc(refs, names, lens, [batch, comb])
    assert_results_equal(g, c, rel=0.0, what="novel")
    novel = (g["junc_refdir"] == 0) & (g["junc_cnt"].sum(axis=0) > 0)
    assert novel.sum() > 50 and g["junc_cnt"][:, novel].sum() > 5000


COVER = ("chrA\t1000\t3000\tG/E/+/1/1000/3000/clean\t0\t+\t1000\t3000\t0\t2\t600,900\t0,1100\n"
         "chrA\t5000\t5040\tH/E/-/1/5000/5040/clean\t0\t-\t5000\t5040\t0\t1\t40\t0\n"
         "chrZ\t10\t500\tZ/E/+/1/10/500/clean\t0\t+\t10\t500\t0\t1\t490\t0\n"
         "chrA\t7000\t7001\tI/E/+/1/7000/7001/clean\t0\t+\t7000\t7001\t0\t1\t1\t0\n")
SJ = "chrA\t1000\t3000\t+\nchrA\t5000\t5040\t-\nchrZ\t10\t500\t+\n"
SP = "chrA\t1000\t+\nchrA\t3000\t+\nchrA\t5000\t-\nchrA\t5040\t-\nchrZ\t10\t+\n"
ROI = "chrA\t8000\t9000\tR0\nchrZ\t0\t100\tRZ\nchrA\t6000\t6500\tR1\n"


def test_deep_introns_use_multilevel_select():
    """Depth far beyond the 2048-bin histogram (forces the radix-select levels), reference-only chromosome chrZ,
    1-base and 40-base introns (flank windows clipped), ROI/points on a chromosome absent from the BAM."""
    names, lens = ["chrB", "chrA"], [4000, 20000]
    rng = np.random.default_rng(5)
    recs = []
    for k in range(9000):                       # deep pile over the first intron: depth up to ~9000
        s = int(rng.integers(900, 1400))
        recs.append(dict(ref_id=1, pos=s, flag=0x10 if k % 3 else 0, cigar="%dM" % int(rng.integers(30, 1500))))
    for k in range(5000):                       # 2^22-scale is unreachable here; a second level is what we can force
        recs.append(dict(ref_id=1, pos=2100 + (k % 7), flag=0, cigar="100M"))
    for k in range(300):
        recs.append(dict(ref_id=1, pos=4990 + (k % 5), flag=0x10, cigar="10M40N12M"))
        recs.append(dict(ref_id=1, pos=900, flag=0, cigar="100M2000N50M"))
    recs.append(dict(ref_id=0, pos=5, flag=0, cigar="50M"))
    recs.append(dict(ref_id=1, pos=8100, flag=0, cigar="200M"))
    names_, lens_, batch = read_batch(records_to_bam(names, lens, recs))
    refs = (COVER, SJ, SP, ROI)
    e, g = run_gpu(refs, names, lens, split(batch, 4))
    o, c = run_orc(refs, names, lens, [batch])
    assert_results_equal(g, c, rel=0.0, what="deep")
    assert g["nondir"]["intron_depth"][0] > 2048          # really beyond the single-level histogram
    assert g["chr_names"] == ["chrB", "chrA", "chrZ"]
    assert g["nondir"]["warning"][2] == 1 and g["nondir"]["coverage"][2] == 0.0   # chrZ intron: all-zero row, LowCover


def test_empty_and_reset():
    ann, sim = small_world(seed=2)
    names, lens, batch = read_batch(sim.bam(0, 3000))
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref)
    e.chr_map_update(names, lens)
    e.process(Batch.empty())
    e.finalize()
    z = e.results()
    o, c = run_orc(ann.ref_texts(), names, lens, [])
    assert_results_equal(z, c, rel=0.0, what="empty")
    with pytest.raises(engine.IrfError):
        e.process(batch)                      # finalized: needs reset
    e.reset()
    e.process(batch)
    e.finalize()
    o, c = run_orc(ann.ref_texts(), names, lens, [batch])
    assert_results_equal(e.results(), c, rel=0.0, what="after reset")
    # second sample through the same context
    e.reset()
    b2 = read_batch(sim.bam(5000, 2000))[2]
    e.process(b2)
    e.finalize()
    o, c = run_orc(ann.ref_texts(), names, lens, [b2])
    assert_results_equal(e.results(), c, rel=0.0, what="second sample")


def test_queued_table_exports_overlap_the_next_sample():
    """IRF_COL_NO_SYNC exports of the IR tables run on the library's export stream: the next sample is reset, submitted and
    finalized without an irf_sync in between (include/irf.h).  Three samples through one context, every table into its own
    page-locked buffer, one irf_sync at the end: every table equals the oracle's for ITS sample (a copy that ran late
    would hold the next sample's rows; planes that were not cleared, the previous sample's depth)."""
    import torch
    ann, sim = small_world(seed=4)
    names, lens, _ = read_batch(sim.bam(0, 100))
    samples = [read_batch(sim.bam(a, n))[2] for a, n in ((0, 20000), (20000, 6000), (30000, 12000))]
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref)
    e.chr_map_update(names, lens)
    n_in = ann.cover_bed.count("\n")
    bufs = []
    for b in samples:
        e.reset()
        e.process(b)
        e.finalize()
        f64 = torch.empty((8, n_in), dtype=torch.float64, pin_memory=True)
        u32 = torch.empty((6, n_in), dtype=torch.int32, pin_memory=True)
        f64.fill_(-1.0); u32.fill_(-1)
        e.export_device(_abi.COL_IR_NONDIR_F64 | _abi.COL_NO_SYNC, f64.data_ptr())
        e.export_device(_abi.COL_IR_NONDIR_U32 | _abi.COL_NO_SYNC, u32.data_ptr())
        bufs.append((f64, u32))
    e.sync()
    for k, b in enumerate(samples):
        o, c = run_orc(ann.ref_texts(), names, lens, [b])
        nd = c["nondir"]
        f64, u32 = bufs[k][0].numpy(), bufs[k][1].numpy().view(np.uint32)
        for j, col in enumerate(_abi.IR_F64):
            assert np.array_equal(f64[j], np.asarray(nd[col], np.float64)), (k, col)
        for j, col in enumerate(_abi.IR_U32 + ("warning",)):
            assert np.array_equal(u32[j], np.asarray(nd[col]).astype(np.uint32)), (k, col)
        assert (f64[1] > 0).sum() > 3, k


@pytest.mark.parametrize("keep_depth", [0, 1])
def test_many_samples_through_one_context_leave_nothing_behind(keep_depth):
    """Six samples through one context: sorted and shuffled batches (shuffled ones are counted as given: every difference
    event then takes the far path — line cache, fallback atomics), small and large, one of them empty, then the first one
    again.  Every sample's tables equal the oracle's for that sample alone: the planes a sample inherits (two spare sets,
    cleared by the line marks of the sample that used them: engine.hip PlaneSet, count.hip sparse_clear_kernel) hold
    nothing of its predecessors.  keep_depth = 1: whole-genome planes, no plane map."""
    ann, sim = small_world(seed=8)
    names, lens, _ = read_batch(sim.bam(0, 100))
    rng = np.random.default_rng(11)
    spans = [(0, 25000), (25000, 4000), (30000, 15000), None, (0, 25000), (45000, 9000)]
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref, engine.default_params(keep_depth=keep_depth))
    e.chr_map_update(names, lens)
    for k, sp in enumerate(spans):
        if sp is None:
            batches = []
        else:
            b = read_batch(sim.bam(*sp))[2]
            if k % 2 == 1 or k == 4:
                b = b.select(rng.permutation(b.n_frag))          # (counted in this order: IRF_ORDER_SORTED is only a hint)
            batches = split(b, 2)
        e.reset()
        for b in batches:
            e.process(b)
        e.finalize()
        o, c = run_orc(ann.ref_texts(), names, lens, batches, keep_depth=keep_depth)
        assert_results_equal(e.results(), c, rel=0.0, what="sample %d" % k)


def test_diagnostic_environment_switches_do_not_reach_the_product(monkeypatch):
    """IRF_ABLATE / IRF_ABLATE_STATS / IRF_RESCAN_PLANES / IRF_STATS_* exist only in -DIRF_DIAG builds
    (tools/build_variant.sh): with them set, the shipped library still gives the oracle's tables."""
    for k, v in (("IRF_ABLATE", "127"), ("IRF_ABLATE_STATS", "3"), ("IRF_RESCAN_PLANES", "1"), ("IRF_STATS_SMALL_MAX", "1"),
                 ("IRF_STATS_WGS_BIG", "1"), ("IRF_STATS_WGS_SMALL", "1")):
        monkeypatch.setenv(k, v)
    ann, sim = small_world(seed=5)
    names, lens, batch = read_batch(sim.bam(0, 4000))
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref)
    e.chr_map_update(names, lens)
    e.process(batch)
    e.finalize()
    o, c = run_orc(ann.ref_texts(), names, lens, [batch])
    assert_results_equal(e.results(), c, rel=0.0, what="with diagnostic variables set")


def test_bad_input_is_reported_not_counted():
    ann, sim = small_world(seed=2)
    names, lens, batch = read_batch(sim.bam(0, 100))
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref)
    with pytest.raises(engine.IrfError):
        e.process(batch)                      # before chr_map_update
    e.chr_map_update(names, lens)
    bad = batch.select(np.arange(batch.n_frag))
    bad.frag_chr[3] = 999                     # refID beyond the header
    e.process(bad)
    with pytest.raises(engine.IrfError):
        e.finalize()
    e.reset()
    bad = batch.select(np.arange(batch.n_frag))
    bad.blk_len[0] = lens[int(bad.frag_chr[0])]      # block runs off the chromosome
    e.process(bad)
    with pytest.raises(engine.IrfError):
        e.finalize()
    bad = batch.select(np.arange(batch.n_frag))
    bad.frag_blk_off[2] = bad.frag_blk_off[3] + 1      # inconsistent CSR offsets are refused on the host
    e.reset()
    with pytest.raises(engine.IrfError):
        e.process(bad)


def test_device_resident_batch():
    import torch
    ann, sim = small_world(seed=8)
    names, lens, batch = read_batch(sim.bam(0, 8000))
    dev = torch.device("cuda:0")
    t = {n: torch.from_numpy(getattr(batch, n).copy()).to(dev) for n, _ in Batch.FIELDS}
    torch.cuda.synchronize()
    ref = engine.RefTables.from_texts(ann.ref_texts())
    e = engine.Engine(ref)
    e.chr_map_update(names, lens)
    e.submit(engine.device_batch(t, batch.n_frag, batch.n_blk))
    e.finalize()
    o, c = run_orc(ann.ref_texts(), names, lens, [batch])
    assert_results_equal(e.results(), c, rel=0.0, what="device batch")


def test_cli_dropin_direct_adapter_and_bgzf(tmp_path):
    """The command-line engine (same argv as upstream's `irfinder`) vs the oracle CLI: every output file byte-identical,
    through the direct SoA path, through the ReadBlockProcessor adapter (hip_block_processor.h), raw and BGZF input, and
    with the BGZF stream inflated and parsed on the device (--device-reader)."""
    import subprocess
    from irfinder_amd.synth.bam import bgzf_compress
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ann, sim = small_world(seed=31, stranded=True)
    bam = sim.bam(0, 30000)
    names = ["ref-cover.bed", "ref-sj.ref", "ref-read-continues.ref", "ref-ROI.bed"]
    for n, t in zip(names, ann.ref_texts()):
        (tmp_path / n).write_text(t)
    (tmp_path / "in.bam.raw").write_bytes(bam)
    (tmp_path / "in.bam").write_bytes(bgzf_compress(bam))
    # the same records coordinate-sorted (north_star's input order): read with --sorted (mates joined by name)
    import struct
    from test_host_reader import _split_bam
    head, recs = _split_bam(bam)
    recs.sort(key=lambda r: struct.unpack_from("<ii", r, 4))
    (tmp_path / "sorted.bam").write_bytes(bgzf_compress(head + b"".join(recs)))
    # the collated records with their read groups (mates adjacent) in a random order: what BAM2blocks hands the adapter,
    # whose engine bins every batch by coordinate itself (HipReadBlockProcessor sets IRF_ORDER_ANY)
    _, recs = _split_bam(bam)
    groups = []
    for r in recs:
        nm = r[36:36 + r[12]]
        if groups and groups[-1][0] == nm:
            groups[-1][1].append(r)
        else:
            groups.append((nm, [r]))
    order = np.random.default_rng(5).permutation(len(groups))
    (tmp_path / "shuffled.bam.raw").write_bytes(head + b"".join(b"".join(groups[i][1]) for i in order))
    refs = [str(tmp_path / n) for n in names]
    cpu = tmp_path / "cpu"
    cpu.mkdir()
    with open(tmp_path / "in.bam.raw", "rb") as fh:
        p = subprocess.run([os.path.join(root, "oracle", "irfinder_oracle"), str(cpu)] + refs, stdin=fh, capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr
    cpu_stdout = p.stdout
    exe = os.path.join(root, "irfinder_amd", "irfinder_hip")
    for tag, extra, src in (("direct", [], "in.bam.raw"), ("adapter", ["--via-adapter"], "in.bam.raw"), ("adapter-shuffled", ["--via-adapter"], "shuffled.bam.raw"),
                            ("bgzf", ["--bgzf"], "in.bam"),
                            ("sorted", ["--bgzf", "--sorted", "--threads", "4"], "sorted.bam"),
                            ("device-reader", ["--device-reader"], "in.bam"), ("device-reader-1mb", ["--device-reader", "--feed-mb", "1"], "in.bam"),
                            ("device-reader-pipe", ["--device-reader"], "in.bam"),
                            ("device-reader-sorted", ["--device-reader", "--sorted", "--feed-mb", "1"], "sorted.bam")):
        out = tmp_path / tag
        out.mkdir()
        if tag.endswith("-pipe"):      # stdin is a pipe (read in pieces); the other device-reader runs map the file
            p = subprocess.run([exe, str(out)] + refs + extra, input=(tmp_path / src).read_bytes(), capture_output=True, timeout=120)
        else:
            with open(tmp_path / src, "rb") as fh:
                p = subprocess.run([exe, str(out)] + refs + extra, stdin=fh, capture_output=True, timeout=120)
        assert p.returncode == 0, (tag, p.stderr)
        assert p.stdout == cpu_stdout, tag                       # reader summary
        files = sorted(os.listdir(cpu))
        assert files == sorted(os.listdir(out)) and len(files) == 6, tag
        for f in files:
            assert filecmp.cmp(out / f, cpu / f, shallow=False), (tag, f)


@pytest.mark.parametrize("order", [0, 1])
def test_reference_split_at_an_intergenic_gap(order, tmp_path):
    """irf_chr_map_update_ranges (SURVEY §8 e: a sample in which one reference outweighs 1/N of the reads): the heaviest
    reference is cut at an intergenic gap, the two parts go to two contexts, each is fed the WHOLE sample — plus
    hand-made fragments that lie across the cut: one covering it, one with a spliced read whose junction spans it (novel,
    twice; and once more as a junction of ref-sj.ref) — and irf_gather must give the single-context tables and files."""
    from irfinder_amd import distributed as irfdist
    from irfinder_amd._abi import Batch as B_, IRF_FRAG_DIR, IRF_FRAG_TWO
    from irfinder_amd.synth import world as synth_world
    ann, sim = small_world(seed=29, stranded=True, n_chr=3)
    names, lens, batch = read_batch(sim.bam(0, 30000))
    w = np.asarray(lens, np.float64)
    w[0] *= 4                                                  # chrT1 alone is heavier than a rank's fair share
    texts = list(ann.ref_texts())
    masks, starts, ends = irfdist.shard_with_split(w, lens, 2, texts, names)
    cut = int(ends[0][0])
    assert masks[0][0] == 1 and masks[1][0] == 1 and int(starts[1][0]) == cut and 0 < cut < lens[0]
    # fragments across the cut (chrT1 is reference 0): a pair whose read 1 covers the cut; a single read spliced over it
    # (left block ends 150 before the cut, right block starts 250 behind it), twice on the + strand and once on the -
    span_l, span_r = cut - 150, cut + 250
    extra = B_(frag_chr=[0, 0, 0, 0], frag_flags=[IRF_FRAG_DIR | IRF_FRAG_TWO, IRF_FRAG_DIR, IRF_FRAG_DIR, 0],
               frag_start=[cut - 60, cut - 250, cut - 250, cut - 230], frag_end=[cut + 300, cut + 350, cut + 350, cut + 330],
               frag_blk_off=[0, 2, 4, 6, 8],
               blk_start=[cut - 60, cut + 200, cut - 250, span_r, cut - 250, span_r, cut - 230, span_r],
               blk_len=[100, 100, 100, 100, 100, 100, 80, 80], blk_read=[0, 1, 0, 0, 0, 0, 0, 0])
    whole_batch = synth_world.concat_batches([batch, extra])
    for known in (False, True):
        refs = list(texts)
        if known:      # the junction across the cut is listed in ref-sj.ref: known for the owner of its left end, novel for the other context
            refs[1] = refs[1] + "%s\t%d\t%d\t+\n" % (names[0], span_l, span_r)
        e, whole = run_gpu(refs, names, lens, [whole_batch])
        o, c = run_orc(refs, names, lens, [whole_batch])
        assert_results_equal(whole, c, rel=0.0, what="single context vs oracle")
        engs = []
        for r in range(2):
            ei = engine.Engine(engine.RefTables.from_texts(refs))
            ei.chr_map_update(names, lens, masks[r], starts[r], ends[r])
            ei.set_input_order(order)
            ei.process(whole_batch)
            ei.finalize()
            engs.append(ei)
        # each part alone: the chromosome counts of the two contexts add up, a fragment counted where it starts
        parts = [x.results() for x in engs]
        assert np.array_equal(parts[0]["chr_cnt"] + parts[1]["chr_cnt"], whole["chr_cnt"])
        assert np.array_equal(parts[0]["roi_cnt"] + parts[1]["roi_cnt"], whole["roi_cnt"])
        engine.gather(engs, root=0)
        got = engs[0].results()
        assert_results_equal(got, whole, rel=0.0, what="reference split at %d, junction across the cut %s" % (cut, "known" if known else "novel"))
        j = np.nonzero((got["junc_left"] == span_l) & (got["junc_right"] == span_r))[0]
        assert len(j) == 1 and got["junc_cnt"][:, j[0]].sum() == 3
        out_a, out_b = tmp_path / ("whole%d" % known), tmp_path / ("split%d" % known)
        os.makedirs(out_a); os.makedirs(out_b)
        e.write_outputs(str(out_a)); engs[0].write_outputs(str(out_b))
        for f in sorted(os.listdir(out_a)):
            assert filecmp.cmp(out_a / f, out_b / f, shallow=False), f
    # the packed junction columns of such a context are refused (their rows must be filtered by coordinate)
    from irfinder_amd._abi import COL_JUNC_KEYS
    with pytest.raises(engine.IrfError):
        engs[1].export_size(COL_JUNC_KEYS)


def test_rccl_collect_program_with_a_one_rank_communicator(tmp_path):
    """irfinder_amd/irf_rccl_collect (host/rccl_collect_main.cpp + rccl_collect.h: the C recipe of INTEGRATION.md §5,
    linked against librccl) as a fresh child process with a ONE-rank communicator: ncclCommInitRank, the all-reduce of
    the vote, the eight ncclReduce calls with their dtypes and the stream ordering are executed for real; the output
    files must equal the oracle CLI's.  (More ranks need more GPUs than the test box has: tests/test_gpu_distributed.py
    covers the multi-rank arithmetic through torch.distributed.)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ann, sim = small_world(seed=37, stranded=True)
    bam = sim.bam(0, 20000)
    names = ["ref-cover.bed", "ref-sj.ref", "ref-read-continues.ref", "ref-ROI.bed"]
    sj = "".join(l for i, l in enumerate(ann.sj_ref.splitlines(True)) if i % 3)       # a third of ref-sj withheld: novel junctions
    for n, t in zip(names, (ann.cover_bed, sj, ann.read_continues_ref, ann.roi_bed)):
        (tmp_path / n).write_text(t)
    (tmp_path / "in.bam.raw").write_bytes(bam)
    refs = [str(tmp_path / n) for n in names]
    outs = {}
    for tag, exe in (("cpu", os.path.join(root, "oracle", "irfinder_oracle")), ("rccl", os.path.join(root, "irfinder_amd", "irf_rccl_collect"))):
        out = tmp_path / tag
        out.mkdir()
        env = dict(os.environ, IRF_RANK="0", IRF_WORLD="1", IRF_DEVICE="0")
        with open(tmp_path / "in.bam.raw", "rb") as fh:
            p = subprocess.run([exe, str(out)] + refs, stdin=fh, capture_output=True, timeout=180, env=env)
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        outs[tag] = p.stdout
    assert outs["rccl"].endswith(outs["cpu"]) and outs["cpu"]        # (librccl announces its versions on stdout first)
    files = sorted(os.listdir(tmp_path / "cpu"))
    assert files == sorted(os.listdir(tmp_path / "rccl")) and len(files) == 6
    for f in files:
        assert filecmp.cmp(tmp_path / "cpu" / f, tmp_path / "rccl" / f, shallow=False), f


def test_chromosome_shards_sum_to_the_whole_P3():
    """Two contexts with complementary `owned` masks (what two ranks do), each fed the WHOLE batch (foreign fragments
    are ignored), the global strandedness vote imposed on both: the owner's rows / counters equal the single run's."""
    from irfinder_amd import distributed as irfdist
    ann, sim = small_world(seed=19, stranded=True, n_chr=4)
    names, lens, batch = read_batch(sim.bam(0, 30000))
    e, whole = run_gpu(ann.ref_texts(), names, lens, [batch])
    masks = irfdist.shard_chromosomes(lens, 2)
    parts = []
    votes = np.zeros(2, np.int64)
    for m in masks:
        ref = engine.RefTables.from_texts(ann.ref_texts())
        ei = engine.Engine(ref)
        ei.chr_map_update(names, lens, m)
        ei.process(batch)
        ei.finalize()
        votes += np.array(ei.votes(), np.int64)
        parts.append(ei)
    d = 1 if votes[0] > votes[1] * 10 else (-1 if votes[1] > votes[0] * 10 else 0)
    assert d == whole["directionality"] != 0
    res = []
    for ei in parts:
        ei.finalize_directional(d)
        res.append(ei.results())
    name_to_idx = {n: i for i, n in enumerate(names)}
    in_chr = np.array([name_to_idx[l.split("\t", 1)[0]] for l in ann.cover_bed.splitlines()])
    for tname in ("nondir", "dir"):
        for k in whole[tname]:
            merged = np.where(masks[0][in_chr] == 1, res[0][tname][k], res[1][tname][k])
            assert np.array_equal(merged, whole[tname][k]), (tname, k)
    for k in ("chr_cnt", "roi_cnt", "sp_cnt", "junc_cnt"):
        assert np.array_equal(res[0][k] + res[1][k], whole[k]), k
    assert res[0]["n_fragments"] == res[1]["n_fragments"] == batch.n_frag       # submitted, not counted


def test_irf_gather_collects_shards_into_the_root_context(tmp_path):
    """irf_gather (the C ABI's in-process multi-GPU collection): three contexts with disjoint `owned` masks, each fed
    the whole sample; afterwards the ROOT context's results and its six text files equal the single-context run's —
    novel junctions (half of ref-sj.ref withheld), an intron on a chromosome no BAM reference names (nobody's row: the
    root keeps it) and the global strandedness vote included."""
    from irfinder_amd import distributed as irfdist
    ann, sim = small_world(seed=23, stranded=True, n_chr=5)
    names, lens, batch = read_batch(sim.bam(0, 30000))
    cover = ann.cover_bed + "chrNowhere\t10\t500\tQ/E/+/1/10/500/clean\t0\t+\t10\t500\t0\t1\t490\t0\n"
    sj = "".join(l for i, l in enumerate(ann.sj_ref.splitlines(True)) if i % 2 == 0)
    refs = (cover, sj, ann.read_continues_ref, ann.roi_bed)
    e, whole = run_gpu(refs, names, lens, [batch])
    os.makedirs(tmp_path / "whole")
    e.write_outputs(str(tmp_path / "whole"))
    for root in (0, 2):
        masks = irfdist.shard_chromosomes(lens, 3)
        engs = []
        for m in masks:
            ei = engine.Engine(engine.RefTables.from_texts(refs))
            ei.chr_map_update(names, lens, m)
            ei.process(batch)
            ei.finalize()
            engs.append(ei)
        engine.gather(engs, root=root)
        got = engs[root].results()
        assert_results_equal(got, whole, rel=0.0, what="gathered into root %d" % root)
        novel = (got["junc_refdir"] == 0) & (got["junc_cnt"].sum(axis=0) > 0)
        assert novel.sum() > 20 and got["has_dir"]
        out = tmp_path / ("g%d" % root)
        os.makedirs(out)
        engs[root].write_outputs(str(out))
        for f in sorted(os.listdir(tmp_path / "whole")):
            assert filecmp.cmp(out / f, tmp_path / "whole" / f, shallow=False), f
        # a second gather of the same sample would add the peers' columns twice: refused, the results stay as they are
        with pytest.raises(engine.IrfError) as ei2:
            engine.gather(engs, root=root)
        assert ei2.value.code == -4
        assert_results_equal(engs[root].results(), whole, rel=0.0, what="after the refused second gather")
    # the pieces of the recipe for one process per GPU: export (masked rows) -> add -> import, junction rows as coordinates
    import torch
    from irfinder_amd._abi import COL_IR_NONDIR_F64, COL_KEEP_ORPHANS
    a, b = engs[0], engs[1]          # (a was a root above: its tables are complete; b still holds only its own rows)
    n = b.export_size(COL_IR_NONDIR_F64)
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    b.export_device(COL_IR_NONDIR_F64, t.data_ptr())
    name_to_idx = {nm: i for i, nm in enumerate(names)}
    in_chr = [name_to_idx.get(l.split("\t", 1)[0], -1) for l in cover.splitlines()]
    mine = np.array([c >= 0 and masks[1][c] == 1 for c in in_chr])
    tt = t.view(8, -1).cpu().numpy()
    assert np.all(tt[:, ~mine] == 0) and np.array_equal(tt[1, mine], whole["nondir"]["intron_depth"][mine])
    rows = b.export_junctions()
    assert len(rows["chr"]) > 0 and set(rows["chr"].tolist()) <= set(np.nonzero(masks[1])[0].tolist())
    with pytest.raises(engine.IrfError):
        b.import_device(COL_IR_NONDIR_F64, t.data_ptr(), n - 1)          # wrong element count


@pytest.mark.parametrize("stranded", [False, True])
def test_chr22_scale_matches_oracle_and_properties(stranded):
    """BASELINE configs[0] shape (2 M simulated 2x100 pairs, chr22-sized reference; unstranded as the config says, and a
    stranded twin so that the one-pass two-table statistics are oracle-checked at this size) — the largest size the
    oracle finishes in seconds — bit-exact, plus the size-independent properties used at the full bench size:
    P4 nondir = pos + neg (via the directional planes of a stranded twin), P5 sum of FragmentsInChr = fragments,
    P6 SpliceLeft/Right are sums of junction rows."""
    from irfinder_amd.synth import world as synth_world
    from irfinder_amd.synth.annotation import GRCH38, make_annotation
    from irfinder_amd.synth.reads import ReadSimulator
    ann = make_annotation([GRCH38[21]], seed=20240807)
    sim = ReadSimulator(ann, read_len=100, seed=20240807, stranded=stranded)
    batches, st = synth_world.generate_batches(sim, 0, 2_000_000, chunk=250_000, threads=8)
    whole = synth_world.sort_by_coordinate(synth_world.concat_batches(batches))
    e, g = run_gpu(ann.ref_texts(), ann.chr_names, ann.chr_lens, [whole])
    o, c = run_orc(ann.ref_texts(), ann.chr_names, ann.chr_lens, batches)          # oracle sees the unsorted chunks
    assert_results_equal(g, c, rel=0.0, what="chr22 2M pairs, stranded=%s" % stranded)
    assert bool(g["has_dir"]) == stranded
    assert g["chr_cnt"].sum() == whole.n_frag == st["n_pairs"] + st["n_singles"]
    # P6 on the device result: SpliceLeft of an intron = sum over junction rows sharing its start
    starts = np.array([int(l.split("\t")[1]) for l in ann.cover_bed.splitlines()])
    left_sum = {}
    tot = g["junc_cnt"].sum(axis=0)
    for l, t in zip(g["junc_left"], tot):
        left_sum[int(l)] = left_sum.get(int(l), 0) + int(t)
    exp = np.array([left_sum.get(int(s), 0) for s in starts])
    assert np.array_equal(exp, g["nondir"]["splice_left"])


def test_grch38_sized_stranded_sample_matches_oracle():
    """The GRCh38-sized reference of configs[1] / [2] (3.1 G positions, ~200 k introns, compact planes) with a STRANDED
    4 M-pair sample: both IR tables (the one-pass two-table statistics kernel), the vote, every counter and the junction
    table against the oracle, bit for bit.  (The 50 M-pair sizes are property-checked below; the oracle would need
    minutes for them.)"""
    from irfinder_amd.synth import world as synth_world
    from irfinder_amd.synth.annotation import GRCH38, make_annotation
    from irfinder_amd.synth.reads import ReadSimulator
    ann = make_annotation(GRCH38, seed=20240808)
    sim = ReadSimulator(ann, read_len=150, seed=20240808 + 2000, stranded=True)
    batches, st = synth_world.generate_batches(sim, 0, 4_000_000, chunk=500_000, threads=16)
    whole = synth_world.sort_by_coordinate(synth_world.concat_batches(batches))
    e, g = run_gpu(ann.ref_texts(), ann.chr_names, ann.chr_lens, [whole])
    o, c = run_orc(ann.ref_texts(), ann.chr_names, ann.chr_lens, batches)
    assert g["has_dir"] and c["has_dir"]
    assert_results_equal(g, c, rel=0.0, what="GRCh38-sized reference, 4M stranded pairs")
    # the same sample in collated order through the binning step
    e2 = engine.Engine(engine.RefTables.from_texts(ann.ref_texts()))
    e2.chr_map_update(ann.chr_names, ann.chr_lens)
    e2.set_input_order(1)
    for b in batches:
        e2.process(b)
    e2.finalize()
    assert_results_equal(e2.results(), c, rel=0.0, what="GRCh38-sized reference, 4M stranded pairs, IRF_ORDER_ANY")


def test_full_size_properties_configs1():
    """BASELINE configs[1] at full size (50 M simulated 2x150 pairs, GRCh38-sized reference): too large to diff against
    the oracle inside the test suite (bench.py does that once per run: `sample_rows_match_oracle`), so the
    size-independent properties carry it —
      P1/P2 the coordinate-sorted batch in one submit == the unsorted generator chunks in 25 submits, every table bit for bit;
      P5    sum of FragmentsInChr == fragments; every junction event is counted once (known in place + novel sorted);
      P6    SpliceLeft / SpliceRight of an intron == sum of the junction rows that share its start / end;
      and an independent numpy recount of FragmentsInChr and of the reads on known junctions."""
    from irfinder_amd.synth import world as synth_world
    from irfinder_amd.synth.annotation import GRCH38, make_annotation
    from irfinder_amd.synth.reads import ReadSimulator
    ann = make_annotation(GRCH38, seed=20240808)
    sim = ReadSimulator(ann, read_len=150, seed=20240808 + 1000)
    batches, st = synth_world.generate_batches(sim, 0, 50_000_000, chunk=2_000_000, threads=16)
    e1, chunks = run_gpu(ann.ref_texts(), ann.chr_names, ann.chr_lens, batches)
    n_frag = sum(b.n_frag for b in batches)
    whole = synth_world.sort_by_coordinate(synth_world.concat_batches(batches))
    del batches
    e2, g = run_gpu(ann.ref_texts(), ann.chr_names, ann.chr_lens, [whole])
    assert_results_equal(g, chunks, rel=0.0, what="50M pairs: sorted single batch vs unsorted chunks")
    assert g["n_fragments"] == n_frag == st["n_pairs"] + st["n_singles"] == int(g["chr_cnt"].sum())
    assert int(g["junc_cnt"].sum()) == g["n_junction_events"] > 10_000_000
    # FragmentsInChr recounted with numpy
    want = np.zeros((len(ann.chr_names), 2), np.int64)
    np.add.at(want, (whole.frag_chr.astype(np.int64), (whole.frag_flags & 1).astype(np.int64)), 1)
    assert np.array_equal(g["chr_cnt"].astype(np.int64), want)
    # P6
    tot = g["junc_cnt"].sum(axis=0).astype(np.int64)
    key_l = g["junc_chr"].astype(np.int64) << 32 | g["junc_left"].astype(np.int64)
    key_r = g["junc_chr"].astype(np.int64) << 32 | g["junc_right"].astype(np.int64)
    name_to_idx = {n: i for i, n in enumerate(g["chr_names"])}
    cols = [l.split("\t", 3)[:3] for l in ann.cover_bed.splitlines()]
    in_l = np.array([name_to_idx[c[0]] << 32 | int(c[1]) for c in cols], np.int64)
    in_r = np.array([name_to_idx[c[0]] << 32 | int(c[2]) for c in cols], np.int64)
    for key, ik, col in ((key_l, in_l, "splice_left"), (key_r, in_r, "splice_right")):
        uk, inv = np.unique(key, return_inverse=True)
        sums = np.bincount(inv, weights=tot, minlength=len(uk)).astype(np.int64)
        pos = np.searchsorted(uk, ik)
        hit = (pos < len(uk)) & (uk[np.minimum(pos, len(uk) - 1)] == ik)
        exp = np.where(hit, sums[np.minimum(pos, len(uk) - 1)], 0)
        assert np.array_equal(exp, g["nondir"][col].astype(np.int64)), col
    assert (g["nondir"]["intron_depth"] > 0).sum() > 50_000


def test_many_measured_blocks_per_intron_and_limits():
    """Introns whose measured part is cut into more blocks than one descriptor chunk of the statistics kernel holds
    (64: one per lane): 150 and 65 blocks, flank windows spread over several blocks and over the chunk boundary, in
    both size classes (a 30 kb intron for the 256-thread class), stranded and not, both flank modes.  And the stated
    limit: an intron with 2^28 or more measured bases is refused at irf_chr_map_update."""
    rng = np.random.default_rng(77)

    def bed12(chrom, start, blocks, name):
        end = blocks[-1][0] + blocks[-1][1]
        return "%s\t%d\t%d\t%s\t0\t+\t%d\t%d\t255,0,0\t%d\t%s\t%s\n" % (
            chrom, start, end, name, start, end, len(blocks), ",".join(str(l) for _, l in blocks),
            ",".join(str(s - start) for s, _ in blocks))

    def cut(start, n, lo, hi, gap_hi):
        out, p = [], start
        for _ in range(n):
            l = int(rng.integers(lo, hi))
            out.append((p, l))
            p += l + int(rng.integers(1, gap_hi))
        return out
    b150 = cut(1000, 150, 3, 30, 6)                  # tiny blocks: the first window alone spans a dozen of them
    b65 = cut(20000, 65, 20, 90, 10)
    b200 = cut(40000, 200, 100, 220, 30)             # ~30 kb measured: 256-thread class
    cover = bed12("chrA", 1000, b150, "G1/E1/+/1/a/b/clean") + bed12("chrA", 20000, b65, "G2/E2/-/1/a/b/clean") + \
        bed12("chrA", 40000, b200, "G3/E3/+/1/a/b/clean")
    ends = [b150[-1][0] + b150[-1][1], b65[-1][0] + b65[-1][1], b200[-1][0] + b200[-1][1]]
    sj = "chrA\t1000\t%d\t+\nchrA\t20000\t%d\t-\n" % (ends[0], ends[1])
    sp = "chrA\t1000\t+\nchrA\t%d\t+\nchrA\t20000\t-\n" % ends[0]
    roi = "chrA\t90000\t95000\tR0\n"
    names, lens = ["chrA"], [100000]
    recs = []
    for k in range(30000):
        s = int(rng.integers(500, 80000))
        recs.append(dict(ref_id=0, pos=s, flag=0x10 if rng.random() < 0.3 else 0, cigar="%dM" % int(rng.integers(20, 400))))
    for k in range(40):
        recs.append(dict(ref_id=0, pos=950, flag=0, cigar="50M%dN30M" % (ends[0] - 1000)))
    names_, lens_, batch = read_batch(records_to_bam(names, lens, recs))
    refs = (cover, sj, sp, roi)
    for over in (dict(), dict(flank_use_measured=0), dict(flank_len=300, flank_offset=2), dict(dir_ratio=0)):
        e, g = run_gpu(refs, names, lens, split(batch, 3), **over)
        o, c = run_orc(refs, names, lens, [batch], **over)
        assert_results_equal(g, c, rel=0.0, what="many blocks %s" % over)
    assert (g["nondir"]["intron_depth"] > 0).all() and g["nondir"]["splice_exact"][0] == 40
    # the limit
    big = "chrB\t10\t%d\tG/E/+/1/a/b/clean\t0\t+\t10\t%d\t0\t1\t%d\t0\n" % (10 + (1 << 28), 10 + (1 << 28), 1 << 28)
    ref = engine.RefTables.from_texts((big, "", "", ""))
    eng = engine.Engine(ref)
    with pytest.raises(engine.IrfError):
        eng.chr_map_update(["chrB"], [300_000_000])
