"""End-to-end wall clock of the command-line engine on a synthetic BGZF BAM file (page cache warm): compressed file ->
the six IRFinder-*.txt files, device reader vs host reader (and the CPU oracle CLI on inflated input for small sizes).
One JSON line.  python tools/bench_cli.py [--pairs N] [--oracle 0/1]"""
import argparse
import filecmp
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irfinder_amd.synth.annotation import make_annotation, small_chroms
from irfinder_amd.synth.bam import bgzf_compress_mt
from irfinder_amd.synth.reads import ReadSimulator


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8_000_000)
    ap.add_argument("--oracle", type=int, default=0)
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ann = make_annotation(small_chroms(4, 20_000_000), seed=9)
    sim = ReadSimulator(ann, read_len=150, seed=9, with_seq=2)
    tmp = tempfile.mkdtemp(prefix="irf_cli_", dir="/tmp")
    names = ["ref-cover.bed", "ref-sj.ref", "ref-read-continues.ref", "ref-ROI.bed"]
    for n, t in zip(names, ann.ref_texts()):
        open(os.path.join(tmp, n), "w").write(t)
    refs = [os.path.join(tmp, n) for n in names]
    raw = sim.bam(0, args.pairs)
    z = bgzf_compress_mt(raw, 6, 16)
    open(os.path.join(tmp, "in.bam"), "wb").write(z)
    out = {"pairs": args.pairs, "bam_MB": round(len(raw) / 1e6, 1), "bgzf_MB": round(len(z) / 1e6, 1)}
    exe = os.path.join(root, "irfinder_amd", "irfinder_hip")
    runs = [("device_reader", [exe, "OUT"] + refs + ["--device-reader"], "in.bam"),
            ("host_reader_16_threads", [exe, "OUT"] + refs + ["--bgzf", "--threads", "16"], "in.bam")]
    if args.oracle:
        open(os.path.join(tmp, "in.raw"), "wb").write(raw)
        runs.append(("cpu_oracle_on_inflated_input", [os.path.join(root, "oracle", "irfinder_oracle"), "OUT"] + refs, "in.raw"))
    del raw, z
    dirs = []
    for tag, cmd, src in runs:
        best = None
        for rep in range(2):
            d = os.path.join(tmp, "%s_%d" % (tag, rep))
            os.mkdir(d)
            c = [d if a == "OUT" else a for a in cmd]
            t0 = time.perf_counter()
            with open(os.path.join(tmp, src), "rb") as fh:
                p = subprocess.run(c, stdin=fh, capture_output=True, timeout=600)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                print(tag, p.stderr.decode()[-400:], file=sys.stderr)
                sys.exit(1)
            best = dt if best is None else min(best, dt)
        dirs.append(d)
        out[tag] = {"wall_s": round(best, 3), "pairs_per_s": round(args.pairs / best, 1)}
    files = sorted(os.listdir(dirs[0]))
    out["same_output_files"] = all(sorted(os.listdir(d)) == files and all(filecmp.cmp(os.path.join(d, f), os.path.join(dirs[0], f), shallow=False) for f in files) for d in dirs[1:])
    out["n_output_files"] = len(files)
    print(json.dumps(out), flush=True)
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
