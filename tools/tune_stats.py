"""Diagnostic: time the statistics kernel under different size-class thresholds / ablations.
usage: tune_stats.py <pairs> "<SMALL_MAX>:<WGS_BIG>:<WGS_SMALL>:<ABLATE>" ...   (one fresh engine per configuration)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from irfinder_amd import engine
from irfinder_amd._abi import Batch
from irfinder_amd.synth import world
from irfinder_amd.synth.annotation import GRCH38, make_annotation
from irfinder_amd.synth.reads import ReadSimulator

pairs = int(sys.argv[1])
ann = make_annotation(GRCH38, seed=20240808)
sim = ReadSimulator(ann, read_len=150, seed=20240808)
bs, st = world.generate_batches(sim, 0, pairs, threads=16)
host = world.sort_by_coordinate(world.concat_batches(bs))
dev = torch.device("cuda:0")
dten = {n: torch.from_numpy(getattr(host, n)).to(dev) for n, _ in Batch.FIELDS}
db = engine.device_batch(dten, host.n_frag, host.n_blk)
ref = engine.RefTables.from_texts(ann.ref_texts())
for cfg in sys.argv[2:]:
    sm, wb, ws, ab = cfg.split(":")
    os.environ["IRF_STATS_WGS_BIG"] = wb
    os.environ["IRF_STATS_WGS_SMALL"] = ws
    os.environ["IRF_STATS_SMALL_MAX"] = sm
    os.environ["IRF_ABLATE_STATS"] = ab
    e = engine.Engine(ref)
    e.chr_map_update(ann.chr_names, ann.chr_lens)
    t = []
    for _ in range(3):
        e.reset(); e.submit(db); e.finalize()
        t.append(e.timing()["stats_ms"])
    print("tune_stats small_max=%s wgs_big=%s wgs_small=%s ablate=%s stats_ms=%.3f (min of %s)" % (sm, wb, ws, ab, min(t), ["%.3f" % x for x in t]), flush=True)
    del e
