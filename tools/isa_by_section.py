#!/usr/bin/env python3
"""Diagnostic: static instruction tally of one kernel per source line (compile with -gline-tables-only -S first).
usage: tools/isa_by_section.py /tmp/isa/count.s <kernel name substring> irfinder_amd/csrc/count.hip [min instructions per line]
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --cuda-device-only -o /tmp/isa/count.s count.hip"""
import collections
import re
import sys

asm, kern, src = sys.argv[1], sys.argv[2], open(sys.argv[3]).read().split("\n")
least = int(sys.argv[4]) if len(sys.argv) > 4 else 12
infn, cur = False, None
cnt = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for l in open(asm).read().split("\n"):
    m = re.match(r"^(_Z\S+):", l)
    if m:
        infn = kern in m.group(1)
        continue
    if infn and l.startswith(".Lfunc_end"):
        break
    if not infn:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    s = l.strip()
    if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
        continue
    op = s.split()[0]
    k = "S" if op.startswith("s_") else "V" if op.startswith("v_") else "DS" if op.startswith("ds_") else "MEM"
    if op in ("v_readlane_b32", "v_writelane_b32"):
        k = "LANE"          # SGPR spill traffic and wave-uniform reads of a lane's value
    if op.startswith("scratch_"):
        k = "SCRATCH"
    if op.startswith("s_waitcnt"):
        k = "WAIT"
    if op.startswith("s_cbranch") or op == "s_branch":
        k = "BR"
    cnt[cur] += 1
    kinds[cur][k] += 1
    tot[k] += 1
files = collections.Counter()
for (f, _), c in cnt.items():
    files[f] += c
mainf = files.most_common(1)[0][0]
print("kernel *%s*: %d instructions (static); by kind %s" % (kern, sum(cnt.values()), dict(tot)))
print("lines of the main source file with at least %d instructions (line, instructions, kinds, source):" % least)
for (f, l), c in sorted(cnt.items()):
    if f == mainf and c >= least and 0 < l <= len(src):
        print("%5d %5d %-44s %s" % (l, c, dict(kinds[(f, l)]), src[l - 1].strip()[:100]))
