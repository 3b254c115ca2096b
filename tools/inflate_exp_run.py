"""Diagnostic: one feed of a synthetic BGZF BAM through the device reader of the library named by IRF_LIB_PATH (an
IRF_INFLATE_EXP build prints its inflate time to stderr; its output is wrong by design, so errors are ignored).
python tools/inflate_exp_run.py [pairs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cache = "/tmp/inflate_exp_%d.bgzf" % pairs
if not os.path.exists(cache):
    from irfinder_amd.synth.annotation import make_annotation, small_chroms
    from irfinder_amd.synth.bam import bgzf_compress_mt
    from irfinder_amd.synth.reads import ReadSimulator
    sim = ReadSimulator(make_annotation(small_chroms(4, 20_000_000), seed=9), read_len=150, seed=9, with_seq=2)
    open(cache, "wb").write(bgzf_compress_mt(sim.bam(0, pairs), 6, 16))
z = np.fromfile(cache, np.uint8)
from irfinder_amd import engine
for rep in range(2):
    d = engine.DeviceReader()
    try:
        d.feed(z)
    except Exception as e:
        print("[inflate_exp_run] feed: %s" % str(e)[:80], file=sys.stderr)
    d.close()
