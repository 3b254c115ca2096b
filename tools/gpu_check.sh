#!/bin/bash
# GPU box: parity tests, then a short bench line.  usage: tools/gpu_check.sh <tag> [pytest -k expression]
tag=$1; sel=${2:-}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread ${sel:+-k "$sel"} > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -2 gpurun_out/${tag}_tests.log
timeout -k 10 280 python bench.py --no-decode-bench --cpu-pairs 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("ms/step", d["ms_per_step"], {k:v.get("ms") for k,v in d["roofline"]["per_kernel"].items()}, "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
r=d.get("order_random") or {}
print("random:", r.get("ms_per_step"), r.get("count_ms"), r.get("same_rows_as_sorted_run"), "as given:", r.get("counted_as_given"))
P
