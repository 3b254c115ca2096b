"""Diagnostic (CPU): what the DEFLATE streams of the synthetic BGZF BAM consist of — literals, matches, runs of literals in
front of a match, match lengths, distance classes — for the first members of a small file (device inflate).
A plain bit-by-bit RFC 1951 decoder: slow, and only counts.  python tools/deflate_symbol_stats.py"""
import sys, struct, collections
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from irfinder_amd.synth.annotation import make_annotation, small_chroms
from irfinder_amd.synth.bam import bgzf_compress_mt
from irfinder_amd.synth.reads import ReadSimulator
sim = ReadSimulator(make_annotation(small_chroms(4, 20_000_000), seed=9), read_len=150, seed=9, with_seq=2)
z = bgzf_compress_mt(sim.bam(0, 20000), 6, 4)
# members
members=[]; off=0
while off < len(z):
    bsize = struct.unpack_from('<H', z, off+16)[0]+1
    members.append(z[off+18:off+bsize-8]); off += bsize
class BR:
    def __init__(s,d): s.d=d; s.p=0; s.bb=0; s.bc=0
    def need(s,n):
        while s.bc<n:
            s.bb |= (s.d[s.p] if s.p<len(s.d) else 0) << s.bc; s.p+=1; s.bc+=8
    def take(s,n):
        if n==0: return 0
        s.need(n); v=s.bb&((1<<n)-1); s.bb>>=n; s.bc-=n; return v
def build(lens):
    # returns dict (len,code)->sym
    bl=collections.Counter(l for l in lens if l); code=0; nxt={}
    for L in range(1,16):
        code=(code+bl.get(L-1,0))<<1; nxt[L]=code
    tab={}
    for s,l in enumerate(lens):
        if l: tab[(l,nxt[l])]=s; nxt[l]+=1
    return tab
def dec(br,tab):
    code=0
    for L in range(1,16):
        code=(code<<1)|br.take(1)
        if (L,code) in tab: return tab[(L,code)]
    raise ValueError
LB=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LE=[0]*8+[1]*4+[2]*4+[3]*4+[4]*4+[5]*4+[0]
DB=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DE=[0,0,0,0]+[i//2 for i in range(2,28)]
ORD=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
tot=collections.Counter(); mlen=collections.Counter(); runs=collections.Counter(); dists=collections.Counter()
for m in members[1:6]:
    br=BR(m); out=0; run=0; nsym=0; nlit=0; nmatch=0; iters_old=0; blocks=0
    prev_was=None
    while True:
        last=br.take(1); bt=br.take(2); blocks+=1
        if bt==0: raise SystemExit('stored')
        if bt==1:
            ll=[8]*144+[9]*112+[7]*24+[8]*8; dl=[5]*30
        else:
            hlit=br.take(5)+257; hdist=br.take(5)+1; hclen=br.take(4)+4
            cl=[0]*19
            for i in range(hclen): cl[ORD[i]]=br.take(3)
            ct=build(cl); lens=[]
            while len(lens)<hlit+hdist:
                s=dec(br,ct)
                if s<16: lens.append(s)
                elif s==16: lens+= [lens[-1]]*(3+br.take(2))
                elif s==17: lens+=[0]*(3+br.take(3))
                else: lens+=[0]*(11+br.take(7))
            ll=lens[:hlit]; dl=lens[hlit:]
        lt=build(ll); dt=build(dl)
        while True:
            s=dec(br,lt)
            if s<256:
                nlit+=1; out+=1; run+=1
            elif s==256: break
            else:
                runs[min(run,8)]+=1; run=0
                s-=257; L=LB[s]+br.take(LE[s]); d=dec(br,dt); D=DB[d]+br.take(DE[d])
                nmatch+=1; out+=L; mlen[min(L,40)]+=1
                dists['<8' if D<8 else '<32' if D<32 else '<L' if D<L else 'far']+=1
        if last: break
    print('member out',out,'in',len(m),'blocks',blocks,'literals',nlit,'matches',nmatch)
print('literal run before a match (capped 8):',sorted(runs.items()))
print('match len (capped 40):',sorted(mlen.items()))
print('dist classes',dists)
