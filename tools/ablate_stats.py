"""Diagnostic: time the per-intron statistics kernel with parts switched off (IRF_ABLATE_STATS bits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from irfinder_amd import engine
from irfinder_amd._abi import Batch
from irfinder_amd.synth import world
from irfinder_amd.synth.annotation import GRCH38, make_annotation
from irfinder_amd.synth.reads import ReadSimulator

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ann = make_annotation(GRCH38, seed=20240808)
sim = ReadSimulator(ann, read_len=150, seed=20240808)
bs, st = world.generate_batches(sim, 0, pairs, threads=16)
host = world.sort_by_coordinate(world.concat_batches(bs))
dev = torch.device("cuda:0")
dten = {n: torch.from_numpy(getattr(host, n)).to(dev) for n, _ in Batch.FIELDS}
db = engine.device_batch(dten, host.n_frag, host.n_blk)
ref = engine.RefTables.from_texts(ann.ref_texts())
e = engine.Engine(ref)
e.chr_map_update(ann.chr_names, ann.chr_lens)
for mask in [0, 0, 1, 2, 3]:
    os.environ["IRF_ABLATE_STATS"] = str(mask)
    e.reset(); e.submit(db); e.finalize()
    print("ablate_stats=%d stats_ms=%.3f" % (mask, e.timing()["stats_ms"]), flush=True)
