"""Appendix-E helper: writes BASELINE configs[0] (2 M simulated 2x100 bp pairs, chr22-sized reference) as the four
reference files and a BGZF BAM into a directory.  usage: tools/appendix_e_data.py OUTDIR [pairs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irfinder_amd.synth.annotation import GRCH38, make_annotation
from irfinder_amd.synth.bam import bgzf_compress_mt
from irfinder_amd.synth.reads import ReadSimulator

out = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
os.makedirs(out, exist_ok=True)
ann = make_annotation([GRCH38[21]], seed=20240807)
sim = ReadSimulator(ann, read_len=100, seed=20240807, with_seq=2)
for n, t in zip(["ref-cover.bed", "ref-sj.ref", "ref-read-continues.ref", "ref-ROI.bed"], ann.ref_texts()):
    open(os.path.join(out, n), "w").write(t)
open(os.path.join(out, "in.bam"), "wb").write(bgzf_compress_mt(sim.bam(0, pairs), 6, 8))
print("wrote %s: 4 reference files + in.bam (%d pairs, name-collated as upstream expects)" % (out, pairs))
