"""Diagnostic: the count kernel alone on the bench workload (configs[1], coordinate-sorted, 10 % novel junctions).
usage: [IRF_LIB_PATH=tools/variants/libirfhip_<tag>.so] python tools/count_bench.py [pairs] [reps]
The workload is generated once per box and kept in /tmp (a later call in the same gpurun command loads it)."""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ann = make_annotation(GRCH38, seed=20240808)
cache = "/tmp/irf_world_%d.npz" % pairs
names = [n for n, _ in Batch.FIELDS]
if os.path.exists(cache):
    z = np.load(cache)
    cols = {n: z[n] for n in names}
else:
    sim = ReadSimulator(ann, read_len=150, seed=20240808)
    bs, st = world.generate_batches(sim, 0, pairs, threads=16)
    host = world.sort_by_coordinate(world.concat_batches(bs))
    cols = {n: getattr(host, n) for n in names}
    np.savez(cache, **cols)
shuffle = bool(os.environ.get("IRF_BENCH_SHUFFLE"))         # the same fragments in a random order, binned on the device (IRF_ORDER_ANY)
if shuffle:
    cols = {n: getattr(Batch(**cols).select(np.random.default_rng(7).permutation(len(cols["frag_chr"]))), n) for n in names}
n_frag, n_blk = len(cols["frag_chr"]), len(cols["blk_start"])
dev = torch.device("cuda:0")
n_batches = int(os.environ.get("IRF_BENCH_BATCHES", "1"))         # the sample submitted as this many batches (consecutive fragments)
dbs, keep = [], []
if n_batches == 1:
    parts = [cols]
else:
    whole = Batch(**cols)
    edges = np.linspace(0, n_frag, n_batches + 1).astype(np.int64)
    parts = [{n: getattr(whole.select(np.arange(edges[i], edges[i + 1])), n) for n in names} for i in range(n_batches)]
for part in parts:
    dten = {n: torch.from_numpy(part[n]).to(dev) for n in names}
    keep.append(dten)
    dbs.append(engine.device_batch(dten, len(part["frag_chr"]), len(part["blk_start"])))
ref = engine.RefTables.from_texts(ann.ref_texts(hide_sj_frac=0.1, hide_seed=20240808))
e = engine.Engine(ref)
e.chr_map_update(ann.chr_names, ann.chr_lens)
if shuffle:
    e.set_input_order(1)
ms, wall = [], []
for r in range(reps):
    e.sync(); t0 = time.perf_counter()
    e.reset()
    for db in dbs:
        e.submit(db)
    e.finalize(); e.sync()
    wall.append((time.perf_counter() - t0) * 1e3)
    t = e.timing()
    ms.append(t["count_ms"])
    if os.environ.get("IRF_CHECK_EVERY_REP"):
        res = e.results()
        d = res["nondir"]["intron_depth"]
        print("  rep %d: depth sum %.6f max %.1f coverage sum %.6f" % (r, float(np.nansum(d)), float(np.nanmax(d)), float(np.nansum(res["nondir"]["coverage"]))), flush=True)
res = e.results()
chk = (int(res["chr_cnt"].sum()), int(res["sp_cnt"].astype(np.uint64).sum()), int(res["junc_cnt"].astype(np.uint64).sum()), int(res["roi_cnt"].sum()),
       float(np.nansum(res["nondir"]["intron_depth"])))
print("lib=%s%s%s count_ms min %.3f all %s stats_ms %.3f step wall ms min %.3f check %s" % (os.path.basename(engine.LIB_PATH), " shuffled+ORDER_ANY" if shuffle else "", " %d batches" % n_batches if n_batches > 1 else "", min(ms),
      ["%.3f" % m for m in ms], t["stats_ms"], min(wall), chk), flush=True)
