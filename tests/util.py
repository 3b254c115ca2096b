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


def split_case(ann, names, lens, batch, world=2):
    """A sample in which reference 0 outweighs a rank's fair share: shard_with_split cuts it at an intergenic gap.
    Returns (masks, own_start, own_end, batch + four hand-made fragments that lie across the cut: a pair whose read 1
    covers it and three single reads spliced over it)."""
    import numpy as np
    from irfinder_amd import distributed as irfdist
    from irfinder_amd._abi import Batch, IRF_FRAG_DIR, IRF_FRAG_TWO
    from irfinder_amd.synth import world as synth_world
    w = np.asarray(lens, np.float64)
    w[0] *= 2 * len(lens)          # (heavier than a fair share whatever the number of references)
    masks, starts, ends = irfdist.shard_with_split(w, lens, world, list(ann.ref_texts()), names)
    cut = int(ends[[r for r in range(world) if masks[r][0] and int(starts[r][0]) == 0][0]][0])
    assert 300 < cut < lens[0] - 400
    span_r = cut + 250
    extra = Batch(frag_chr=[0, 0, 0, 0], frag_flags=[IRF_FRAG_DIR | IRF_FRAG_TWO, IRF_FRAG_DIR, IRF_FRAG_DIR, 0],
                  frag_start=[cut - 60, cut - 250, cut - 250, cut - 230], frag_end=[cut + 300, cut + 350, cut + 350, cut + 330],
                  frag_blk_off=[0, 2, 4, 6, 8],
                  blk_start=[cut - 60, cut + 200, cut - 250, span_r, cut - 250, span_r, cut - 230, span_r],
                  blk_len=[100, 100, 100, 100, 100, 100, 80, 80], blk_read=[0, 1, 0, 0, 0, 0, 0, 0])
    return masks, starts, ends, synth_world.concat_batches([batch, extra]), cut
