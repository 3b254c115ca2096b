"""Shared helpers for the tests: synthetic worlds and result comparison."""
import numpy as np

from irfinder_amd._abi import IR_F64, IR_U32
from irfinder_amd.synth.annotation import make_annotation, small_chroms
from irfinder_amd.synth.reads import ReadSimulator


def small_world(n_pairs=20000, seed=7, stranded=False, read_len=100, n_chr=3, length=400_000, single_frac=0.02,
                extra_bam_chr=True, **sim_kw):
    """Annotation on a toy genome + a simulator; the BAM header carries one extra reference the ref files never name."""
    chroms = small_chroms(n_chr, length)
    ann = make_annotation(chroms, seed=seed)
    sim = ReadSimulator(ann, read_len=read_len, stranded=stranded, seed=seed, single_frac=single_frac, **sim_kw)
    return ann, sim


def assert_results_equal(a, b, rel=0.0, what=""):
    """a, b: dicts from results_to_dict (+chr_names).  Integers bit-exact, doubles within rel (0 => bit-exact)."""
    assert a["chr_names"] == b["chr_names"], what
    assert a["directionality"] == b["directionality"], what
    for k in ("chr_cnt", "roi_cnt", "sp_chr", "sp_pos", "sp_cnt", "junc_chr", "junc_left", "junc_right", "junc_cnt",
              "junc_refdir"):
        assert a[k].shape == b[k].shape, (what, k, a[k].shape, b[k].shape)
        if not np.array_equal(a[k], b[k]):
            bad = np.argwhere(a[k] != b[k])[:5]
            raise AssertionError("%s: %s differs at %s" % (what, k, bad.tolist()))
    assert a["n_intron"] == b["n_intron"] and a["has_dir"] == b["has_dir"], what
    for tname in ("nondir", "dir"):
        ta, tb = a[tname], b[tname]
        for k in IR_U32 + ("warning",):
            if not np.array_equal(ta[k], tb[k]):
                bad = np.argwhere(ta[k] != tb[k])[:5].ravel()
                raise AssertionError("%s: %s.%s differs at rows %s: %s vs %s" % (what, tname, k, bad.tolist(),
                                                                                ta[k][bad].tolist(), tb[k][bad].tolist()))
        for k in IR_F64:
            if rel == 0.0:
                ok = np.array_equal(ta[k], tb[k])
            else:
                ok = np.allclose(ta[k], tb[k], rtol=rel, atol=0.0)
            if not ok:
                bad = np.argwhere(ta[k] != tb[k])[:5].ravel()
                raise AssertionError("%s: %s.%s differs at rows %s: %s vs %s" % (what, tname, k, bad.tolist(),
                                                                                ta[k][bad].tolist(), tb[k][bad].tolist()))
    for k in ("n_fragments", "n_blocks", "n_junction_events"):
        assert a[k] == b[k], (what, k, a[k], b[k])
