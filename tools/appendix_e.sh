#!/bin/bash
# if the upstream engine's source is ever at hand, build it and diff its six output files
# against the CPU restatement (oracle/) and against irfinder_hip on BASELINE configs[0].  Until this has run clean no
# document of this repository may say "bit-exact vs reference": parity is UNPINNED.
#
#   tools/appendix_e.sh <upstream-src-dir> [workdir]        e.g. <checkout>/src/irfinder
#
# Never fetches anything: the source directory must already exist.  The GPU leg runs only where a HIP device is visible.
set -u
src=${1:-}; work=${2:-/tmp/irf_appendix_e}
root=$(cd "$(dirname "$0")/.." && pwd)
if [ -z "$src" ] || ! ls "$src"/*.cpp > /dev/null 2>&1; then
  echo "usage: $0 <upstream-src-dir> [workdir]   (a directory holding the engine's *.cpp; nothing is downloaded)"; exit 2
fi
mkdir -p "$work" || exit 1
echo "== 1. building upstream: g++ -O3 -std=${IRF_AE_STD:-c++11} $src/*.cpp"
g++ -O3 -std=${IRF_AE_STD:-c++11} -I"$src" "$src"/*.cpp -o "$work/irfinder_upstream" -lz 2> "$work/build.log" || { echo "build failed (see $work/build.log): treat upstream as unbuildable here and say so in DESIGN.md"; tail -20 "$work/build.log"; exit 1; }
echo "== 2. configs[0] input (synthetic: no data ships with the reference)"
python3 "$root/tools/appendix_e_data.py" "$work/data" ${IRF_AE_PAIRS:-2000000} || exit 1
refs="$work/data/ref-cover.bed $work/data/ref-sj.ref $work/data/ref-read-continues.ref $work/data/ref-ROI.bed"
run() {   # tag, command... (stdin: the inflated BAM, as upstream's `gzip -cd x.bam |` pipe delivers it)
  tag=$1; shift; rm -rf "$work/$tag"; mkdir -p "$work/$tag"
  gzip -cd < "$work/data/in.bam" | "$@" > "$work/$tag.stdout" 2> "$work/$tag.stderr" || { echo "$tag failed:"; tail -5 "$work/$tag.stderr"; return 1; }
}
echo "== 3. running upstream, the CPU restatement$( [ -e /dev/kfd ] && echo ', irfinder_hip')"
run upstream "$work/irfinder_upstream" "$work/upstream" $refs || exit 1
run oracle "$root/oracle/irfinder_oracle" "$work/oracle" $refs || exit 1
legs="oracle"
if [ -e /dev/kfd ]; then run hip "$root/irfinder_amd/irfinder_hip" "$work/hip" $refs && legs="$legs hip"; fi
echo "== 4. differences (none = this restatement reproduces upstream on this input)"
python3 - "$work" $legs <<'P'
import os, sys
work, legs = sys.argv[1], sys.argv[2:]
# which open question (SURVEY.md §0.35 / irf_params switch) a differing column points at
HINT = {
    "IRFinder-IR-nondir.txt": {"Coverage": "OQ-7/8 (coverage definition)", "IntronDepth": "OQ-8/9 trim_centre_pct, OQ-7 pct_interpolate", "IntronDepth25Percentile": "OQ-7 pct_interpolate",
                               "IntronDepth50Percentile": "OQ-7 pct_interpolate", "IntronDepth75Percentile": "OQ-7 pct_interpolate",
                               "ExonToIntronReadsLeft": "OQ-11 sp_inclusive / sp_overhang_*", "ExonToIntronReadsRight": "OQ-11 sp_inclusive / sp_overhang_*",
                               "IntronDepthFirst50bp": "OQ-13 flank_offset / flank_len / flank_use_measured", "IntronDepthLast50bp": "OQ-13 flank_offset / flank_len / flank_use_measured",
                               "SpliceLeft": "OQ-4 junc_min_flank, OQ-1/2 fragment assembly", "SpliceRight": "OQ-4 junc_min_flank, OQ-1/2 fragment assembly",
                               "SpliceExact": "OQ-4 junc_min_flank, OQ-6 novel junctions", "IRratio": "follows IntronDepth / Splice*", "Warnings": "OQ-12 warn_* thresholds, warn_nonuniform_and"},
    "IRFinder-JuncCount.txt": {"*": "OQ-4 junc_min_flank, OQ-5 dir_ratio (strand column), OQ-1/2/3 mate merge"},
    "IRFinder-ChrCoverage.txt": {"*": "OQ-1/2/16 fragment assembly in BAM2blocks (what counts as one fragment)"},
    "IRFinder-ROI.txt": {"*": "OQ-10 ROI containment rule"},
    "IRFinder-SpansPoint.txt": {"*": "OQ-11 sp_inclusive / sp_overhang_left / sp_overhang_right"},
}
HINT["IRFinder-IR-dir.txt"] = HINT["IRFinder-IR-nondir.txt"]
clean = True
up = os.path.join(work, "upstream")
for leg in legs:
    d = os.path.join(work, leg)
    fa, fb = sorted(os.listdir(up)), sorted(os.listdir(d))
    if fa != fb:
        clean = False
        print("%s: file lists differ: upstream %s, %s %s (IRFinder-IR-dir.txt exists only when the sample is judged stranded: OQ-5 dir_ratio)" % (leg, fa, leg, fb))
    for f in sorted(set(fa) & set(fb)):
        a = open(os.path.join(up, f)).read().splitlines()
        b = open(os.path.join(d, f)).read().splitlines()
        if a == b:
            print("%s: %-28s identical (%d lines)" % (leg, f, len(a)))
            continue
        clean = False
        head = a[0].split("\t") if a else []
        cols = {}
        for x, y in zip(a, b):
            if x == y:
                continue
            cx, cy = x.split("\t"), y.split("\t")
            for k in range(max(len(cx), len(cy))):
                if k >= len(cx) or k >= len(cy) or cx[k] != cy[k]:
                    name = head[k] if k < len(head) and not head[k].replace(".", "").isdigit() else "col%d" % k
                    cols[name] = cols.get(name, 0) + 1
        print("%s: %-28s DIFFERS: %d vs %d lines; differing cells per column: %s" % (leg, f, len(a), len(b), cols))
        h = HINT.get(f, {})
        for name in cols:
            print("      %-28s -> look at %s" % (name, h.get(name, h.get("*", "(no hint: a format difference?)"))))
sa = open(os.path.join(work, "upstream.stdout")).read()
for leg in legs:
    if open(os.path.join(work, leg + ".stdout")).read() != sa:
        clean = False
        print("%s: the reader summary on stdout differs -> OQ-1/2/3/16 (BAM2blocks: pairing, overlap merge, skipped reads)" % leg)
print("RESULT:", "no difference: record the upstream commit in BASELINE.json, then the documents may cite it" if clean else "differences above: resolve the open questions they point at (irf_params switches), run again")
sys.exit(0 if clean else 1)
P
