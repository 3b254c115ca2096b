#!/usr/bin/env python3
"""Regenerates tests/golden/*: small committed input/output vectors for the counting path.

WHAT THESE ARE.  /root/reference holds no code and no fixtures, so no vector here comes from
upstream IRFinder: parity is UNPINNED.  Two kinds of fixture are written:

  kat_*.json      hand-derived known answers of SURVEY.md App. C (the expected values were worked out on paper from
                  the recalled spec, then typed in below — the oracle is CHECKED against them, it does not produce them)
  world_<tag>/    a seeded synthetic sample: the four reference-table texts, a raw BAM byte stream (gzip'ed), and the
                  output files the CPU oracle (oracle/irf_oracle.cpp) writes for it.  They freeze today's behaviour
                  of the restated spec so that neither the oracle nor the HIP path can drift silently; the GPU tests
                  require the HIP path to reproduce the same bytes.

Run from the repo root:  python tests/golden/make_golden.py
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF_NAMES = ("ref-cover.bed", "ref-sj.ref", "ref-read-continues.ref", "ref-ROI.bed")

# ---- hand-derived known answers -----------------------------------------------------------------
KAT_CIGAR = [   # KAT-1: cigar -> (block starts relative to pos, block lengths, reference length)
    {"cigar": "50M1000N50M", "starts": [0, 1050], "lens": [50, 50], "genome_len": 1100},
    {"cigar": "5S45M2D5M100N3I50M", "starts": [0, 152], "lens": [52, 50], "genome_len": 202},
    {"cigar": "10M100N200N10M", "starts": [0, 310], "lens": [10, 10], "genome_len": 320},
]
KAT_PAIRS = [   # KAT-2: read 1 forward at pos1, read 2 reverse at pos2 -> blocks (start, end, read) of the fragment
    {"pos1": 100, "cigar1": "50M", "pos2": 130, "cigar2": "50M", "class": "intersect", "blocks": [[100, 180, 0]]},
    {"pos1": 100, "cigar1": "50M", "pos2": 150, "cigar2": "50M", "class": "intersect", "blocks": [[100, 200, 0]]},
    {"pos1": 100, "cigar1": "50M", "pos2": 151, "cigar2": "50M", "class": "long", "blocks": [[100, 150, 0], [151, 201, 1]]},
    {"pos1": 100, "cigar1": "100M", "pos2": 120, "cigar2": "50M", "class": "short", "blocks": [[100, 200, 0]]},
    {"pos1": 100, "cigar1": "30M500N20M", "pos2": 640, "cigar2": "40M", "class": "intersect",
     "blocks": [[100, 130, 0], [630, 680, 0]]},
]
KAT_STATS = [   # KAT-5 and the percentile rule (OQ-7/8 defaults): depth multiset -> statistics
    {"values": [0] * 20 + [2] * 20 + [3] * 80, "coverage": [100, 120], "trimmed": [140, 48], "p25": 2.0, "p50": 3.0, "p75": 3.0},
    {"values": [30, 0, 20, 10], "coverage": [3, 4], "trimmed": [30, 2], "p25": 2.5, "p50": 15.0, "p75": 27.5},
    {"values": [7], "coverage": [1, 1], "trimmed": [7, 1], "p25": 7.0, "p50": 7.0, "p75": 7.0},
]
KAT_ENGINE = {  # KAT-3/4/6/7/8 on one tiny reference: records -> counters
    "chr": ["chrA"], "len": [100000],
    "refs": {
        "ref-cover.bed": "chrA\t90\t210\tG1/ENSG1/+/1/90/210/clean\t0\t+\t90\t210\t255,0,0\t1\t120\t0\n",
        "ref-sj.ref": "chrA\t130\t630\t+\n",
        "ref-read-continues.ref": "chrA\t105\t+\nchrA\t106\t+\nchrA\t175\t+\nchrA\t176\t+\n",
        "ref-ROI.bed": "chrA\t0\t1000\tROI0\nchrA\t2000\t3000\tROI1\n",
    },
    "cases": [
        {"name": "kat3_spliced_merge", "records": [[100, 0x61, "30M500N20M"], [640, 0x91, "40M"]],
         "junc": [[130, 630, 0, 1]]},                                  # left, right, neg, pos
        {"name": "kat3_short_flank", "records": [[126, 0x61, "4M500N20M"], [640, 0x91, "40M"]],
         "junc": [[130, 630, 0, 0]]},
        {"name": "kat6_spanspoint", "records": [[100, 0x0, "80M"]], "sp_pos": [105, 106, 175, 176], "sp_pos_cnt": [0, 1, 1, 0]},
        {"name": "kat7_roi", "records": [[100, 0x61, "50M"], [400, 0x91, "50M"], [900, 0x61, "100M"], [1000, 0x91, "100M"],
                                         [2000, 0x10, "1000M"]],
         "roi_cnt": [[0, 1], [1, 0]], "chr_cnt": [[1, 2]]},             # [neg][roi] ; [chr][neg,pos]
        {"name": "kat4_kat8", "records": [[100, 0x61, "50M"], [130, 0x91, "50M"], [100, 0x61, "50M"], [150, 0x91, "50M"],
                                          [100, 0x61, "100M"], [120, 0x91, "50M"]]
         + [[40, 0x0, "50M120N30M"]] * 5 + [[40, 0x0, "50M400N30M"]] * 2,
         "depth_pos_90_210": [0] * 10 + [3] * 80 + [2] * 20 + [0] * 10,
         "splice": [7, 5, 5], "intron_depth": [140, 48], "irratio_den_add": 7, "warning": "LowCover"},
    ],
}

WORLDS = {      # tag -> (seed, stranded, pairs)
    "unstranded": (101, False, 6000),
    "stranded": (102, True, 6000),
}


def main():
    import orc
    from util import small_world
    with open(os.path.join(HERE, "kat_vectors.json"), "w") as f:
        json.dump({"source": "hand-derived from SURVEY.md App. C (recalled spec); not produced by any program",
                   "cigar2block": KAT_CIGAR, "pairs": KAT_PAIRS, "stats": KAT_STATS, "engine": KAT_ENGINE}, f, indent=1)
        f.write("\n")
    for tag, (seed, stranded, pairs) in WORLDS.items():
        d = os.path.join(HERE, "world_" + tag)
        os.makedirs(os.path.join(d, "expected"), exist_ok=True)
        ann, sim = small_world(seed=seed, stranded=stranded, n_chr=2, length=400_000)
        bam = sim.bam(0, pairs)
        with open(os.path.join(d, "sample.bam.raw.gz"), "wb") as raw:
            with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, compresslevel=9) as z:   # mtime=0: byte-stable
                z.write(bam)
        for name, text in zip(REF_NAMES, ann.ref_texts()):
            with open(os.path.join(d, name), "w") as f:
                f.write(text)
        o = orc.Oracle(ann.ref_texts())
        o.process_bam(bam)
        o.finalize()
        for old in os.listdir(os.path.join(d, "expected")):
            os.remove(os.path.join(d, "expected", old))
        o.write_outputs(os.path.join(d, "expected"))
        o.close()
        print(tag, len(bam), "BAM bytes ->", sorted(os.listdir(os.path.join(d, "expected"))))


if __name__ == "__main__":
    main()
