"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the chromosome sharding + table collection (SURVEY §8 e, P3)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

import orc
from irfinder_amd import distributed as irfdist
from irfinder_amd import engine
from util import small_world, split_case

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_lpt_sharding_is_a_partition_and_balanced():
    lens = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57, 1]
    for world in (1, 2, 4, 8):
        masks = irfdist.shard_chromosomes(lens, world)
        tot = np.sum(masks, axis=0)
        assert np.array_equal(tot, np.ones(len(lens)))
        loads = [float(np.dot(m, lens)) for m in masks]
        assert max(loads) <= 1.25 * (sum(lens) / world) + max(lens) * (world > 4)
    assert irfdist.shard_chromosomes(lens, 4)[0].tolist() == irfdist.shard_chromosomes(lens, 4)[0].tolist()


def test_decode_junction_keys_roundtrip():
    lens = [1000, 5000, 70, 3000]
    owned = np.array([1, 0, 1, 1], np.uint8)
    base = irfdist.plane_bases(lens, owned)
    assert base.tolist() == [0, 1024, 1024, 1152, 4160]
    items = [(0, 10, 500, 1), (2, 5, 60, 0), (3, 100, 2999, 1), (3, 100, 2999, 0)]
    keys = np.array([((int(base[c]) + l) << irfdist.JUNC_LEFT_SHIFT) | ((r - l) << 1) | d for c, l, r, d in items], np.uint64)
    ch, l, r, d = irfdist.decode_junction_keys(keys, lens, owned)
    assert list(zip(ch.tolist(), l.tolist(), r.tolist(), d.tolist())) == items


@pytest.mark.parametrize("world,stranded,split", [(2, 0, 0), (3, 1, 0), (2, 1, 1)])
def test_gloo_sharded_run_equals_single_run(world, stranded, split, tmp_path):
    """split = 1: reference 0 outweighs a rank's share and is cut at an intergenic gap (distributed.shard_with_split, the
    ownership rules of irf_chr_map_update_ranges), with fragments lying across the cut."""
    out = tmp_path / "gathered.pkl"
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(HERE, "dist_worker.py"), str(out), str(stranded), str(split)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    got = pickle.load(open(out, "rb"))
    # single-process reference on the same input
    ann, sim = small_world(seed=17, stranded=bool(stranded), n_chr=5)
    rd = engine.Reader()
    rd.feed(sim.bam(0, 20000))
    rd.finish()
    names, lens = rd.header()
    batch = rd.take_batch()
    if split:
        masks, starts, ends, batch, cut = split_case(ann, names, lens, batch, world)
        assert got["masks"][0][0] == 1 and got["masks"][1][0] == 1          # both ranks own a part of reference 0
    o = orc.Oracle(ann.ref_texts())
    o.chr_map_update(names, lens)
    o.process_batch(batch)
    o.finalize()
    r = o.results()
    t = r["nondir"]
    f = np.stack([t[k] for k in ("coverage", "intron_depth", "p25", "p50", "p75", "first50", "last50", "irratio")])
    u = np.stack([t[k].astype(np.int64) for k in ("sp_left", "sp_right", "splice_left", "splice_right", "splice_exact")] +
                 [t["warning"].astype(np.int64)])
    assert np.array_equal(got["ir_f64"], f)                     # bit-exact: x + 0 = x
    assert np.array_equal(got["ir_u32"].astype(np.int64), u)
    assert np.array_equal(got["sp_cnt"].astype(np.int64), r["sp_cnt"].astype(np.int64).ravel())
    assert np.array_equal(got["roi_cnt"].astype(np.int64), r["roi_cnt"].astype(np.int64).ravel())
    assert np.array_equal(got["chr_cnt"].astype(np.int64), r["chr_cnt"].astype(np.int64).ravel())
    exp = []
    for j in range(len(r["junc_left"])):
        for d in (0, 1):
            if r["junc_cnt"][d, j]:
                exp.append((int(r["junc_chr"][j]), int(r["junc_left"][j]), int(r["junc_right"][j]), d, int(r["junc_cnt"][d, j])))
    assert got["junc"] == sorted(exp)
    assert len(got["masks"]) == world


def test_pack_and_unpack_of_the_fixed_tables_are_inverse():
    """gather_tables sends the device-resident tables of one dtype as one flat buffer (one RCCL reduce instead of one per
    table): packing and unpacking must be exact inverses for the shapes it meets ([8,n], [6,n], flat, empty)."""
    import torch
    from irfinder_amd.distributed import pack_tables, unpack_tables
    g = torch.Generator().manual_seed(3)
    tabs = {"ir_u32": torch.randint(0, 1 << 30, (6, 17), generator=g, dtype=torch.int32), "sp_cnt": torch.randint(0, 99, (11,), generator=g, dtype=torch.int32),
            "roi_cnt": torch.zeros(0, dtype=torch.int32), "known_cnt": torch.randint(0, 99, (2, 5), generator=g, dtype=torch.int32)}
    keys = list(tabs)
    flat = pack_tables(tabs, keys)
    assert flat.numel() == sum(t.numel() for t in tabs.values())
    back = unpack_tables(flat + 0, tabs, keys)
    for k in keys:
        assert back[k].shape == tabs[k].shape and torch.equal(back[k], tabs[k]), k
