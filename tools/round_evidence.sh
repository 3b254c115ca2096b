#!/bin/bash
# GPU box: the round's evidence in one go.  usage: tools/round_evidence.sh <tag, e.g. r04>   (results under gpurun_out/<tag>_evidence/)
# 1. HBM traffic per kernel (two --pmc passes) -> profiles/<tag>_pmc_summary.json on the box, so that the bench lines below carry it
# 2. default bench.py line (all legs)  3. the same workload under rocprofv3 --kernel-trace --stats  4. configs[2] (stranded)
tag=$1; out=gpurun_out/${tag}_evidence; mkdir -p $out
bash tools/pmc_traffic.sh ${tag}_evidence/pmc > $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 1; }
cp $out/pmc/pmc_summary.json profiles/${tag}_pmc_summary.json
timeout -k 10 900 python bench.py > $out/bench_line.json 2> $out/bench_line.err || { tail -5 $out/bench_line.err; exit 1; }
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-decode-bench --no-random-leg --cpu-pairs 0 --samples 0 --steps 10 > $GRAFT_REPO_ROOT/$out/bench_line_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err) || { tail -5 $out/rocprof.err; exit 1; }
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
timeout -k 10 600 python bench.py --stranded 1 --no-decode-bench --samples 0 > $out/bench_line_configs2_stranded.json 2> $out/configs2.err || { tail -5 $out/configs2.err; exit 1; }
python - <<P
import json
for f in ("bench_line", "bench_line_under_rocprof", "bench_line_configs2_stranded"):
    d = json.load(open("$out/%s.json" % f))
    r = d["roofline"]
    print(f, "ms/step", d["ms_per_step"], "value %.3g" % d["value"], {k: (v.get("ms"), v.get("frac"), v.get("wasted")) for k, v in r["per_kernel"].items()}, "frac", r["frac"], "traffic", r["traffic"], "wasted", r.get("wasted"),
          "oracle:", (d.get("oracle_check") or {}).get("differing"), "random:", (d.get("order_random") or {}).get("ms_per_step"))
P
head -8 $out/kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
