#!/usr/bin/env python3
"""Diagnostic: print the kernel timeline of the last bench step from a tools/trace_step.sh directory."""
import csv, glob, re, sys
ev = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']; m = re.search(r'(\w+_kernel(?:<[^>]*>)?|fillBuffer\w*|copyBuffer\w*)', n)
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), (m.group(1) if m else n[:50]), r.get('Stream_Id', '')))
ev.sort()
idx = [i for i, e in enumerate(ev) if 'count_kernel' in e[2]]
print('step periods (ms):', [(ev[b][0] - ev[a][0]) / 1e6 for a, b in zip(idx, idx[1:])])
i0 = idx[-2]; t0 = ev[i0][0]
small = 0.0
for e in ev[i0:idx[-1] + 1]:
    d = (e[1] - e[0]) / 1e6
    if d < 0.03 and '-v' not in sys.argv: small += d; continue
    print('%9.3f %9.3f %8.3f  %s  stream %s' % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, d, e[2][:60], e[3]))
print('kernels under 0.03 ms: %.3f ms in total' % small)
