"""Diagnostic: time the count kernel with parts switched off (IRF_ABLATE bits).  Results are wrong by design."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from irfinder_amd import engine
from irfinder_amd._abi import Batch
from irfinder_amd.synth import world
from irfinder_amd.synth.annotation import GRCH38, make_annotation
from irfinder_amd.synth.reads import ReadSimulator

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
order = sys.argv[2] if len(sys.argv) > 2 else "sorted"
ann = make_annotation(GRCH38, seed=20240808)
sim = ReadSimulator(ann, read_len=150, seed=20240808)
bs, st = world.generate_batches(sim, 0, pairs, threads=16)
host = world.concat_batches(bs)
if order == "sorted":
    host = world.sort_by_coordinate(host)
dev = torch.device("cuda:0")
dten = {n: torch.from_numpy(getattr(host, n)).to(dev) for n, _ in Batch.FIELDS}
db = engine.device_batch(dten, host.n_frag, host.n_blk)
ref = engine.RefTables.from_texts(ann.ref_texts(hide_sj_frac=0.1, hide_seed=20240808))     # the bench workload: 10 % novel junctions
e = engine.Engine(ref)
e.chr_map_update(ann.chr_names, ann.chr_lens)
masks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 0, 1, 1 | 8, 1 | 2, 1 | 4, 1 | 16, 1 | 2 | 8, 1 | 2 | 4 | 8, 31, 30, 2 | 8]
for mask in masks:
    os.environ["IRF_ABLATE"] = str(mask)
    e.reset(); e.submit(db)
    try:
        e.finalize()
    except Exception as ex:
        pass
    print("ablate=%3d off=[%s] count_ms=%.3f" % (mask, ",".join(n for b, n in ((1, "depth"), (2, "sp"), (4, "roi"), (8, "junc"), (16, "chr"), (32, "blockloop"), (64, "folding"), (256, "flush-atomics")) if mask & b), e.timing()["count_ms"]), flush=True)
