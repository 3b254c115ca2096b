#!/bin/bash
# HBM traffic per kernel launch of the default bench workload: FETCH_SIZE and WRITE_SIZE in two separate rocprofv3 --pmc
# passes, summarised as profiles/<tag>_pmc_summary.json expects.  usage (GPU box): tools/pmc_traffic.sh <outdir-under-gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-decode-bench --no-random-leg > $out/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $out/$c.log; exit 1; }
done
python3 - $out <<'P'
import csv, glob, json, sys, collections, re, hashlib, os
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        m = re.search(r'(\w+_kernel(?:<[\d, ]+>)?|fillBufferAligned|copyBuffer)', name)
        key = m.group(1) if m else name[:40]
        vals[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-decode-bench --no-random-leg (two separate passes)",
       "unit": "bytes per launch (mean over launches)",
       "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section; factor 2.0 calibrated in r01b on a kernel that read exactly 2*plane_positions*4 B)",
       "sources_sha256": {f: hashlib.sha256(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "irfinder_amd", "csrc", f), "rb").read()).hexdigest()[:16] for f in ("count.hip", "stats.hip", "kernels.h")},
       "kernels": {}}
for k, d in sorted(vals.items()):
    f = d.get('FETCH_SIZE', []); w = d.get('WRITE_SIZE', [])
    if not f or not w: continue
    fb = sum(f) / len(f) * 1024; wb = sum(w) / len(w) * 1024
    out["kernels"][k] = {"launches": len(f), "FETCH_SIZE_bytes_raw": int(fb), "WRITE_SIZE_bytes": int(wb), "hbm_bytes_corrected": int(2 * fb + wb)}
json.dump(out, open(sys.argv[1] + '/pmc_summary.json', 'w'), indent=1)
for k, v in out["kernels"].items(): print(k, v)
P
