"""Seeded synthetic GENCODE-like annotation and the four IRFinder reference files built from it.

No annotation, genome or aligner exists in the build or GPU containers, so every input of the
counting path is synthesised (SURVEY §8 d).  The file formats are the recalled ones of SURVEY App. B.2:

  ref-cover.bed            BED12, one record per intron, blocks = measurable sub-intervals
  ref-sj.ref               chr start end strand
  ref-read-continues.ref   chr pos strand
  ref-ROI.bed              chr start end name   (non-overlapping)
"""
from dataclasses import dataclass, field

import numpy as np

GRCH38 = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259),
    ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422),
    ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189),
    ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415), ("chrM", 16569),
]


@dataclass
class Annotation:
    chr_names: list
    chr_lens: list
    # transcripts (flat arrays)
    tx_chr: np.ndarray          # int32
    tx_strand: np.ndarray       # int8: 1 '+', 0 '-'
    tx_ex_off: np.ndarray       # int64, n_tx+1
    ex_start: np.ndarray        # uint32
    ex_end: np.ndarray          # uint32
    tx_weight: np.ndarray       # float64 (abundance)
    tx_ir_prob: np.ndarray      # float64
    # reference files
    cover_bed: str = ""
    sj_ref: str = ""
    read_continues_ref: str = ""
    roi_bed: str = ""
    n_intron: int = 0
    n_junction: int = 0
    n_point: int = 0
    n_roi: int = 0
    measured_bases: int = 0
    extra: dict = field(default_factory=dict)

    def ref_texts(self, hide_sj_frac=0.0, hide_seed=0):
        """The four reference files.  hide_sj_frac > 0: that fraction of the ref-sj.ref rows (seeded choice) is withheld
        from the engine, so the reads simulated over those junctions are NOVEL junctions for it — real RNA-seq always has
        some (the radix sort-reduce path); the simulator still draws from every transcript."""
        sj = self.sj_ref
        if hide_sj_frac > 0.0:
            lines = sj.splitlines(True)
            keep = np.random.default_rng(hide_seed).random(len(lines)) >= hide_sj_frac
            sj = "".join(l for l, k in zip(lines, keep) if k)
        return (self.cover_bed, sj, self.read_continues_ref, self.roi_bed)


def _subtract(s, e, cuts):
    """[s,e) minus a list of [a,b) intervals -> list of [x,y) pieces."""
    out = []
    cur = s
    for a, b in sorted(cuts):
        if b <= cur:
            continue
        if a >= e:
            break
        if a > cur:
            out.append((cur, min(a, e)))
        cur = max(cur, b)
        if cur >= e:
            break
    if cur < e:
        out.append((cur, e))
    return out


def make_annotation(chroms=None, seed=20240807, genes_per_mb=8.0, max_genes=None, min_gap=2000):
    """chroms: list of (name, length).  Deterministic in (chroms, seed, genes_per_mb)."""
    chroms = chroms or GRCH38
    rng = np.random.default_rng(seed)
    tx_chr, tx_strand, tx_off, ex_s, ex_e, tx_w, tx_ir = [], [], [0], [], [], [], []
    cover_lines, sj_lines, sp_lines, roi_lines = [], [], [], []
    n_intron = n_junc = n_point = n_roi = 0
    measured_total = 0
    gene_no = 0
    for ci, (cname, clen) in enumerate(chroms):
        pos = int(rng.integers(1000, 20000)) if clen > 50000 else 100
        prev_gene_end = 0
        mean_gap = 1e6 / genes_per_mb
        chr_junc = {}
        chr_points = {}
        while True:
            gap = int(rng.exponential(mean_gap)) + min_gap
            gstart = pos + gap
            n_ex = 1 + int(rng.geometric(1.0 / 9.0))
            n_ex = min(n_ex, 60)
            ex_len = np.clip(np.exp(rng.normal(np.log(130.0), 0.6, n_ex)), 20, 5000).astype(np.int64)
            in_len = np.clip(np.exp(rng.normal(np.log(1600.0), 1.3, max(n_ex - 1, 0))), 70, 1_000_000).astype(np.int64)
            span = int(ex_len.sum() + in_len.sum())
            if gstart + span + 1000 >= clen:
                break
            if max_genes is not None and gene_no >= max_genes:
                break
            # master exon list
            starts = np.zeros(n_ex, np.int64)
            cur = gstart
            for k in range(n_ex):
                starts[k] = cur
                cur += int(ex_len[k])
                if k < n_ex - 1:
                    cur += int(in_len[k])
            ends = starts + ex_len
            strand = int(rng.integers(0, 2))
            ssym = "+" if strand else "-"
            gname = "G%05d" % gene_no
            n_tx = int(rng.integers(1, 5)) if n_ex > 2 else 1
            expressed = rng.random() < 0.7
            gene_ir = float(rng.beta(0.5, 8.0))
            gene_introns = {}
            tx_exon_sets = []
            for t in range(n_tx):
                if t == 0 or n_ex <= 2:
                    keep = np.ones(n_ex, bool)
                else:
                    keep = rng.random(n_ex) >= 0.15
                    keep[0] = keep[-1] = True
                idx = np.nonzero(keep)[0]
                tx_exon_sets.append(idx)
                tx_chr.append(ci)
                tx_strand.append(strand)
                for k in idx:
                    ex_s.append(int(starts[k]))
                    ex_e.append(int(ends[k]))
                tx_off.append(len(ex_s))
                tx_w.append(float(np.exp(rng.normal(0.0, 1.5))) if expressed else 0.0)
                tx_ir.append(gene_ir if len(idx) > 1 else 0.0)
                for a, b in zip(idx[:-1], idx[1:]):
                    gene_introns[(int(ends[a]), int(starts[b]))] = (int(a), int(b))
            # ROI: the intergenic gap before this gene
            if gstart - prev_gene_end > 12000 and prev_gene_end > 0:
                roi_lines.append("%s\t%d\t%d\tIntergenic/%s:%d\n" % (cname, prev_gene_end + 1000, gstart - 1000, cname, n_roi))
                n_roi += 1
            # introns -> ref files
            for (is_, ie_), (a, b) in sorted(gene_introns.items()):
                chr_junc[(is_, ie_)] = ssym
                # exclusions: exons of the gene that lie inside this intron (skipped exons), low-mappability stretches
                cuts = [(int(starts[k]), int(ends[k])) for k in range(a + 1, b)]
                kind = "known-exon" if cuts else "clean"
                ilen = ie_ - is_
                n_low = int(rng.integers(0, 4))
                for _ in range(n_low):
                    ll = int(rng.integers(5, max(6, ilen // 10 + 6)))
                    ls = int(rng.integers(is_, max(is_ + 1, ie_ - ll)))
                    cuts.append((ls, min(ie_, ls + ll)))
                pieces = [(x, y) for (x, y) in _subtract(is_, ie_, cuts) if y > x]
                meas = sum(y - x for x, y in pieces)
                if meas < 10:
                    continue
                cover_lines.append("%s\t%d\t%d\t%s/ENSG%08d/%s/%d/%d/%d/%s\t0\t%s\t%d\t%d\t255,0,0\t%d\t%s\t%s\n" % (
                    cname, is_, ie_, gname, gene_no, ssym, a + 1, is_, ie_, kind, ssym, is_, ie_, len(pieces),
                    ",".join(str(y - x) for x, y in pieces), ",".join(str(x - is_) for x, y in pieces)))
                n_intron += 1
                measured_total += meas
                chr_points[is_] = ssym
                chr_points[ie_] = ssym
            prev_gene_end = int(ends[-1])
            pos = prev_gene_end
            gene_no += 1
        if clen - prev_gene_end > 12000 and prev_gene_end > 0:
            roi_lines.append("%s\t%d\t%d\tIntergenic/%s:%d\n" % (cname, prev_gene_end + 1000, clen - 1000, cname, n_roi))
            n_roi += 1
        for (s, e), sym in sorted(chr_junc.items()):
            sj_lines.append("%s\t%d\t%d\t%s\n" % (cname, s, e, sym))
        n_junc += len(chr_junc)
        for p, sym in sorted(chr_points.items()):
            sp_lines.append("%s\t%d\t%s\n" % (cname, p, sym))
        n_point += len(chr_points)
    ann = Annotation(
        chr_names=[c for c, _ in chroms], chr_lens=[int(l) for _, l in chroms],
        tx_chr=np.asarray(tx_chr, np.int32), tx_strand=np.asarray(tx_strand, np.int8),
        tx_ex_off=np.asarray(tx_off, np.int64), ex_start=np.asarray(ex_s, np.uint32), ex_end=np.asarray(ex_e, np.uint32),
        tx_weight=np.asarray(tx_w, np.float64), tx_ir_prob=np.asarray(tx_ir, np.float64),
        cover_bed="".join(cover_lines), sj_ref="".join(sj_lines), read_continues_ref="".join(sp_lines),
        roi_bed="".join(roi_lines), n_intron=n_intron, n_junction=n_junc, n_point=n_point, n_roi=n_roi,
        measured_bases=measured_total)
    if ann.tx_weight.sum() <= 0 and len(ann.tx_weight):
        ann.tx_weight[:] = 1.0
    return ann


def small_chroms(n=3, length=400_000):
    """A toy genome for tests: a few short chromosomes (plus one the reference files never mention)."""
    names = ["chrT%d" % (i + 1) for i in range(n)]
    return [(nm, length + 37 * i) for i, nm in enumerate(names)]
