"""Throughput of the device reader (irf_dreader_*) on a synthetic BGZF BAM with sequence + quality bytes, next to the host
reader on the same bytes.  One JSON line.  python tools/bench_dreader.py [--pairs N] [--level L] [--piece MB]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from irfinder_amd import engine
from irfinder_amd.synth.annotation import make_annotation, small_chroms
from irfinder_amd.synth.bam import bgzf_compress_mt
from irfinder_amd.synth.reads import ReadSimulator

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--piece", type=int, default=0, help="feed size in MB of compressed bytes (0: the whole file at once)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--host-threads", type=int, default=16)
    ap.add_argument("--pinned", type=int, default=0, help="1: the compressed bytes are fed from page-locked host memory")
    ap.add_argument("--pipeline", type=int, default=0, help="1: irf_dreader_set_pipeline (copy of a piece overlaps the inflate of the one before)")
    ap.add_argument("--sorted", type=int, default=0, help="1: the records coordinate-sorted, mates joined by read name (params.mate_join)")
    args = ap.parse_args()
    ann = make_annotation(small_chroms(4, 20_000_000), seed=9)
    sim = ReadSimulator(ann, read_len=args.read_len, seed=9, with_seq=2)
    t0 = time.perf_counter()
    raw = sim.bam(0, args.pairs)
    if args.sorted:
        import struct
        o = len(sim.header())
        head, offs, keys = raw[:o], [], []
        while o < len(raw):
            bs, ref_id, pos = struct.unpack_from("<iii", raw, o)
            offs.append((o, o + 4 + bs)); keys.append((ref_id if ref_id >= 0 else 1 << 30) << 32 | max(pos, 0))
            o += 4 + bs
        order = np.argsort(np.asarray(keys, np.int64), kind="stable")
        raw = head + b"".join(raw[offs[i][0]:offs[i][1]] for i in order)
    params = engine.default_params(mate_join=1) if args.sorted else None
    z = bgzf_compress_mt(raw, args.level, 16)
    print("[bench_dreader] %d pairs, %.1f MB BAM, %.1f MB BGZF (level %d), made in %.1fs" % (
        args.pairs, len(raw) / 1e6, len(z) / 1e6, args.level, time.perf_counter() - t0), file=sys.stderr, flush=True)
    zb = np.frombuffer(z, np.uint8)
    if args.pinned:
        import torch
        zb = torch.from_numpy(zb.copy()).pin_memory().numpy()
    piece = args.piece << 20 if args.piece else len(z)
    out = {"pairs": args.pairs, "bam_MB": round(len(raw) / 1e6, 1), "bgzf_MB": round(len(z) / 1e6, 1), "level": args.level, "pipeline": args.pipeline, "pinned_input": args.pinned, "sorted_name_join": args.sorted,
           "piece_MB": args.piece or round(len(z) / 1e6, 1)}
    best = None
    for rep in range(args.reps):
        d = engine.DeviceReader(params, pipeline=bool(args.pipeline))
        t0 = time.perf_counter()
        n_frag = 0
        for i in range(0, len(z), piece):
            d.feed(zb[i:i + piece])
            n_frag += d.batch_struct().n_frag
        d.finish()
        if args.pipeline:
            n_frag += d.batch_struct().n_frag
        dt = time.perf_counter() - t0
        t = d.timing()
        st = d.stats()
        d.close()
        if best is None or dt < best[0]:
            best = (dt, t, n_frag, st)
    dt, t, n_frag, st = best
    kern = t["inflate_ms"] + t["crc_ms"] + t["chain_ms"] + t["assemble_ms"]
    out["device"] = {"wall_s": round(dt, 4), "pairs_per_s_wall": round(args.pairs / dt, 1), "bgzf_MBps_wall": round(len(z) / dt / 1e6, 1),
                     "bam_GBps_wall": round(len(raw) / dt / 1e9, 2),
                     "device_ms": {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")},
                     "pairs_per_s_device_stages": round(args.pairs / (kern / 1e3), 1),
                     "bam_GBps_device_stages": round(len(raw) / (kern / 1e3) / 1e9, 2),
                     "inflate_GBps_out": round(len(raw) / (t["inflate_ms"] / 1e3) / 1e9, 2), "n_frag": n_frag, "n_members": t["n_members"]}
    r = engine.Reader(params, threads=args.host_threads)
    t0 = time.perf_counter()
    for i in range(0, len(z), 16 << 20):
        r.feed(zb[i:i + (16 << 20)], bgzf=True)
    r.finish()
    dth = time.perf_counter() - t0
    hs = r.stats()
    out["host"] = {"threads": args.host_threads, "wall_s": round(dth, 4), "pairs_per_s": round(args.pairs / dth, 1),
                   "bam_GBps": round(len(raw) / dth / 1e9, 2)}
    out["same_counters"] = hs == st
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
