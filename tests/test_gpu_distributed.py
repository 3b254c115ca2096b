"""The N>1 path with real engines on the GPU: two (and three) child processes on cuda:0, launched with
torch.distributed.run before they touch the GPU, sharded by read-weighted LPT, collected on rank 0 and compared with
a single-context run — tables bit for bit, output files byte for byte (SURVEY §8 e, P3)."""
import filecmp
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from irfinder_amd import engine
from util import assert_results_equal, small_world, split_case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,stranded,split", [(2, True, False), (3, False, False), (2, True, True)])
def test_ranks_with_real_engines_equal_the_single_context_run(world, stranded, split, tmp_path):
    """split: reference 0 outweighs a rank's share and is cut at an intergenic gap between the two ranks (each rank is fed
    the whole sample, fragments across the cut included)."""
    ann, sim = small_world(seed=29, stranded=stranded, n_chr=5)
    r = engine.Reader()
    r.feed(sim.bam(0, 30000))
    r.finish()
    names, lens = r.header()
    batch = r.take_batch()
    extra = {}
    if split:
        masks, starts, ends, batch, cut = split_case(ann, names, lens, batch, world)
        extra["split"] = (masks, starts, ends)
    sj = "".join(l for i, l in enumerate(ann.sj_ref.splitlines(True)) if i % 3)          # a third of the junctions are novel
    cover = ann.cover_bed + "chrNowhere\t10\t500\tQ/E/+/1/10/500/clean\t0\t+\t10\t500\t0\t1\t490\t0\n"
    refs = (cover, sj, ann.read_continues_ref, ann.roi_bed)
    pickle.dump(dict({"refs": refs, "names": names, "lens": lens, "batch": batch}, **extra), open(tmp_path / "world.pkl", "wb"))
    out = tmp_path / "gathered"
    os.makedirs(out)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(HERE, "dist_gpu_worker.py"), str(tmp_path / "world.pkl"), str(out)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    got = pickle.load(open(out / "results.pkl", "rb"))
    # the single-context run, in this process
    e = engine.Engine(engine.RefTables.from_texts(refs))
    e.chr_map_update(names, lens)
    e.process(batch)
    e.finalize()
    whole = e.results()
    assert_results_equal(got["results"], whole, rel=0.0, what="world %d" % world)
    assert whole["has_dir"] == stranded and len(got["masks"]) == world
    assert all(sum(m) > 0 for m in got["masks"])
    os.makedirs(tmp_path / "whole")
    e.write_outputs(str(tmp_path / "whole"))
    files = sorted(os.listdir(tmp_path / "whole"))
    assert len(files) == (6 if stranded else 5)
    for f in files:
        assert filecmp.cmp(out / f, tmp_path / "whole" / f, shallow=False), f
