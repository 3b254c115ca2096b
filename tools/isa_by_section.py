#!/usr/bin/env python3
"""Diagnostic: static instruction tally of count_kernel per source section (compile with -gline-tables-only -S first).
usage: tools/isa_by_section.py /tmp/isa/count.s"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
cur = None; infn = False
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
total = 0; ops = collections.Counter()
for l in lines:
    if re.match(r'^_ZN.*count_kernel\w*:', l): infn = True; continue
    if infn and l.startswith('.Lfunc_end'): break
    if not infn: continue
    m = re.match(r'\s*\.loc\s+\d+\s+(\d+)\s+(\d+)', l)
    if m: cur = int(m.group(1)); continue
    s = l.strip()
    if not s or s.startswith('.') or s.startswith(';') or s.endswith(':') or re.match(r'^[.\w$]+:', s): continue
    op = s.split()[0]
    cnt[cur] += 1; total += 1; ops[op] += 1
    k = 'S' if op.startswith('s_') else 'V' if op.startswith('v_') else 'DS' if op.startswith('ds_') else 'MEM'
    if op in ('v_readlane_b32', 'v_writelane_b32'): k = 'SPILL'
    if op == 's_nop': k = 'NOP'
    if op.startswith('s_waitcnt'): k = 'WAIT'
    if op.startswith('s_cbranch') or op == 's_branch': k = 'BR'
    kinds[cur][k] += 1
print('total', total)
ranges = [(205,213,'init'),(241,289,'frag load'),(290,319,'ROI'),(320,339,'chr hist'),(344,374,'blk cols'),(375,398,'owner'),(399,427,'valid+bucket'),(428,442,'junc cand'),(443,460,'SP req'),(461,484,'known search'),(489,494,'junction_add'),(497,513,'events'),(514,556,'SP'),(557,591,'novel'),(597,624,'flush caches'),(625,644,'flush lines'),(645,662,'flush window'),(663,699,'novel out/end'),(141,149,'seg_add/plane_of'),(152,171,'far_event'),(177,186,'depth_event_p'),(191,200,'junction_add fn'),(125,137,'dpp'),(86,87,'unpack')]
agg = collections.Counter(); aggk = collections.defaultdict(collections.Counter)
for ln, c in cnt.items():
    name = 'other'
    for a, b, n in ranges:
        if ln is not None and a <= ln <= b: name = n; break
    agg[name] += c
    for k, v in kinds[ln].items(): aggk[name][k] += v
for n, c in agg.most_common(): print('%-18s %5d  %s' % (n, c, dict(aggk[n])))
print(ops.most_common(40))
