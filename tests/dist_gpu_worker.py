"""Worker for test_gpu_distributed.py: one rank of a world_size-N run with REAL engines on the GPU.

Every rank builds an Engine over its chromosome shard (all ranks on cuda:0 — the test box has one GPU — so the
collectives run on gloo over host copies of the exported columns; on the 8-GPU node bench.py runs the same calls on
RCCL with the columns left on the device).  The steps are bench.py's: count, finalize, global strandedness vote,
irf_export_device of every column (rows of foreign introns zeroed by the library, the root keeps nobody's rows),
gather_tables, novel junctions as coordinates.  Rank 0 then IMPORTS what the others sent into its own context
(irf_import_device / irf_import_junctions) and writes the six output files, which the test compares byte for byte with
a single-context run.  Started by torch.distributed.run before anything touches the GPU.
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

from irfinder_amd import distributed as irfdist  # noqa: E402
from irfinder_amd import engine  # noqa: E402
from irfinder_amd._abi import (COL_CHR_CNT, COL_IR_DIR_F64, COL_IR_DIR_U32, COL_IR_NONDIR_F64, COL_IR_NONDIR_U32,  # noqa: E402
                               COL_KEEP_ORPHANS, COL_KNOWN_CNT, COL_ROI_CNT, COL_SP_CNT)


def main():
    world_pkl, out_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w = pickle.load(open(world_pkl, "rb"))
    refs, names, lens, batch = w["refs"], w["names"], w["lens"], w["batch"]
    weights = np.bincount(batch.frag_chr, minlength=len(names))          # what a BAI's mapped-read counts give
    dev = torch.device("cuda", 0)
    eng = engine.Engine(engine.RefTables.from_texts(refs), device=0)
    if "split" in w:          # a reference cut at an intergenic gap (distributed.shard_with_split; irf_chr_map_update_ranges)
        masks, starts, ends = w["split"]
        eng.chr_map_update(names, lens, masks[rank], starts[rank], ends[rank])
    else:
        masks = irfdist.shard_chromosomes(weights, world)
        eng.chr_map_update(names, lens, masks[rank])
    eng.process(batch)                                                   # foreign fragments are ignored by the engine
    eng.finalize()
    v = torch.tensor(eng.votes(), dtype=torch.int64)
    eng.finalize_directional(irfdist.global_directionality(v, dist, eng.params.dir_ratio))
    keep = COL_KEEP_ORPHANS if rank == 0 else 0

    def export(col, dtype):
        n = eng.export_size(col)
        t = torch.empty(n, dtype=dtype, device=dev)
        if n:
            eng.export_device(col | (keep if col in (COL_IR_NONDIR_F64, COL_IR_NONDIR_U32, COL_IR_DIR_F64, COL_IR_DIR_U32) else 0), t.data_ptr())
        return t

    cols = {"ir_f64": (COL_IR_NONDIR_F64, torch.float64), "ir_u32": (COL_IR_NONDIR_U32, torch.int32),
            "ir_dir_f64": (COL_IR_DIR_F64, torch.float64), "ir_dir_u32": (COL_IR_DIR_U32, torch.int32),
            "sp_cnt": (COL_SP_CNT, torch.int32), "roi_cnt": (COL_ROI_CNT, torch.int32), "chr_cnt": (COL_CHR_CNT, torch.int32),
            "known_cnt": (COL_KNOWN_CNT, torch.int32)}
    mine = {k: export(c, dt) for k, (c, dt) in cols.items()}
    # the peers' sum arrives on the root: reduce(SUM) of (own column on the peers, zeros on the root), then one import
    tabs = {k: (torch.zeros_like(t) if rank == 0 else t).cpu() for k, t in mine.items() if t.numel()}
    tabs["junc_keys"] = torch.zeros(0, dtype=torch.int64)
    tabs["junc_counts"] = torch.zeros(0, dtype=torch.int32)
    got = irfdist.gather_tables(tabs, dist)
    rows = eng.export_junctions()
    all_rows = [None] * world
    dist.all_gather_object(all_rows, rows)
    if rank == 0:
        for k, (c, dt) in cols.items():
            if k in got and got[k].numel():
                t = got[k].to(dev)
                eng.import_device(c, t.data_ptr(), t.numel())
        for r in range(1, world):
            eng.import_junctions(all_rows[r])
        eng.write_outputs(out_dir)
        res = eng.results()
        pickle.dump({"results": res, "masks": [m.tolist() for m in masks]}, open(os.path.join(out_dir, "results.pkl"), "wb"))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
