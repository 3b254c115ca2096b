"""Worker for test_distributed_cpu.py: one rank of a world_size-N gloo run.

Each rank counts only the fragments on its chromosome shard (with the CPU oracle standing in for the GPU engine —
this test exercises the sharding + collection code of irfinder_amd/distributed.py, which is backend-agnostic),
then the tables are collected on rank 0 exactly as bench.py does on the GPU node.
"""
import os
import pickle
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import orc  # noqa: E402
from irfinder_amd import distributed as irfdist  # noqa: E402
from irfinder_amd import engine  # noqa: E402
from util import small_world, split_case  # noqa: E402


def main():
    out_path = sys.argv[1]
    stranded = bool(int(sys.argv[2]))
    split = len(sys.argv) > 3 and bool(int(sys.argv[3]))       # reference 0 cut at an intergenic gap (shard_with_split)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ann, sim = small_world(seed=17, stranded=stranded, n_chr=5)
    rd = engine.Reader()
    rd.feed(sim.bam(0, 20000))
    rd.finish()
    names, lens = rd.header()
    batch = rd.take_batch()
    if split:
        masks, starts, ends, batch, _ = split_case(ann, names, lens, batch, world)
    else:
        masks = irfdist.shard_chromosomes(lens, world)
        starts = [np.zeros(len(lens), np.uint32) for _ in range(world)]
        ends = [np.full(len(lens), 0xffffffff, np.uint32) for _ in range(world)]
    owned, lo, hi = masks[rank], starts[rank].astype(np.int64), ends[rank].astype(np.int64)

    def in_own(chr_ids, pos):       # the rule of irf_chr_map_update_ranges: a feature belongs to the owner of its start
        chr_ids, pos = np.asarray(chr_ids, np.int64), np.asarray(pos, np.int64)
        return (owned[chr_ids] == 1) & (pos >= lo[chr_ids]) & (pos < hi[chr_ids])

    fc, fs, fe = batch.frag_chr.astype(np.int64), batch.frag_start.astype(np.int64), batch.frag_end.astype(np.int64)
    # counted whole: every fragment that reaches into this rank's part of its reference
    mine = batch.select(np.nonzero((owned[fc] == 1) & (fs < hi[fc]) & (fe > lo[fc]))[0])
    o = orc.Oracle(ann.ref_texts())
    o.chr_map_update(names, lens)
    o.process_batch(mine)
    o.finalize()
    r = o.results()
    # FragmentsInChr / FragmentsInROI: a fragment belongs to the owner of its start
    o2 = orc.Oracle(ann.ref_texts())
    o2.chr_map_update(names, lens)
    o2.process_batch(batch.select(np.nonzero(in_own(fc, fs))[0]))
    o2.finalize()
    r2 = o2.results()
    n_in = r["n_intron"]
    name_to_idx = {n: i for i, n in enumerate(names)}
    cover = [l.split("\t") for l in ann.cover_bed.splitlines()]
    in_chr = np.array([name_to_idx[f[0]] for f in cover], np.int64)
    intron_owned = torch.from_numpy(in_own(in_chr, [int(f[1]) for f in cover]).astype(np.int64))

    def table(t):
        f = np.stack([t[k] for k in ("coverage", "intron_depth", "p25", "p50", "p75", "first50", "last50", "irratio")])
        u = np.stack([t[k].astype(np.int64) for k in ("sp_left", "sp_right", "splice_left", "splice_right", "splice_exact")] +
                     [t["warning"].astype(np.int64)]).astype(np.int32)
        return torch.from_numpy(f.copy()), torch.from_numpy(u.copy())

    f64, u32 = table(r["nondir"])
    irfdist.mask_unowned_rows(f64, u32, intron_owned)
    # observed junctions of this rank, packed with this rank's plane layout
    obs = np.nonzero((r["junc_cnt"].sum(axis=0) > 0) & in_own(r["junc_chr"], r["junc_left"]))[0]        # a junction belongs to the owner of its left end
    sp_mine = in_own(r["sp_chr"], r["sp_pos"]).astype(np.int32)
    base = irfdist.plane_bases(lens, owned)
    keys, cnts = [], []
    for j in obs:
        c, l, rr = int(r["junc_chr"][j]), int(r["junc_left"][j]), int(r["junc_right"][j])
        for d in (0, 1):
            if r["junc_cnt"][d, j]:
                keys.append(((int(base[c]) + l) << irfdist.JUNC_LEFT_SHIFT) | ((rr - l) << 1) | d)
                cnts.append(int(r["junc_cnt"][d, j]))
    order = np.argsort(np.asarray(keys, np.uint64)) if keys else np.zeros(0, np.int64)
    keys = np.asarray(keys, np.uint64)[order]
    cnts = np.asarray(cnts, np.int32)[order]
    tabs = {
        "ir_f64": f64, "ir_u32": u32,
        "sp_cnt": torch.from_numpy((r["sp_cnt"].astype(np.int32) * sp_mine[None, :]).ravel().copy()),
        "roi_cnt": torch.from_numpy(r2["roi_cnt"].astype(np.int32).ravel().copy()),
        "chr_cnt": torch.from_numpy(r2["chr_cnt"].astype(np.int32).ravel().copy()),
        "junc_keys": torch.from_numpy(keys.view(np.int64).copy()),
        "junc_counts": torch.from_numpy(cnts.copy()),
    }
    got = irfdist.gather_tables(tabs, dist)
    if rank == 0:
        junc = []
        for rk, (k, c) in enumerate(zip(got["junc_keys"], got["junc_counts"])):
            ch, l, rr, d = irfdist.decode_junction_keys(k.numpy().view(np.uint64), lens, masks[rk])
            junc += [(int(a), int(b), int(e), int(f), int(g)) for a, b, e, f, g in zip(ch, l, rr, d, c.numpy())]
        with open(out_path, "wb") as fh:
            pickle.dump({"ir_f64": got["ir_f64"].numpy(), "ir_u32": got["ir_u32"].numpy(), "sp_cnt": got["sp_cnt"].numpy(),
                         "roi_cnt": got["roi_cnt"].numpy(), "chr_cnt": got["chr_cnt"].numpy(), "junc": sorted(junc),
                         "masks": [m.tolist() for m in masks]}, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
