"""Minimal BAM / BGZF serialiser for synthetic alignments (SAM spec §4).

Used by tests and by the host-decode leg of bench.py to produce input for the
reader; it is not on the counting path.
"""
import struct
import zlib

import numpy as np

CIGAR_OPS = "MIDNSHP=X"


def cigar_words(cigar):
    out, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | CIGAR_OPS.index(ch))
            num = ""
    return out


def bam_header(chr_names, chr_lens, text="@HD\tVN:1.6\tSO:unsorted\n"):
    t = text.encode()
    out = [b"BAM\1", struct.pack("<i", len(t)), t, struct.pack("<i", len(chr_names))]
    for n, l in zip(chr_names, chr_lens):
        nb = n.encode() + b"\0"
        out += [struct.pack("<i", len(nb)), nb, struct.pack("<i", int(l))]
    return b"".join(out)


def bam_record(ref_id, pos, flag, cigar, name="r", l_seq=None, next_ref=-1, next_pos=-1, tlen=0, mapq=255,
               with_seq=False):
    """One alignment record.  cigar: string or list of words.  l_seq defaults to the query length of the cigar;
    with_seq=False writes l_seq but no seq/qual bytes would be invalid, so then l_seq bytes ARE written as zeros
    only when with_seq is True; otherwise l_seq is stored as 0 (SEQ '*')."""
    words = cigar_words(cigar) if isinstance(cigar, str) else list(cigar)
    qlen = sum(w >> 4 for w in words if (w & 0xf) in (0, 1, 4, 7, 8))
    if l_seq is None:
        l_seq = qlen if with_seq else 0
    nb = name.encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(nb), mapq, 0, len(words), flag, l_seq, next_ref, next_pos, tlen)
    body += nb + struct.pack("<%dI" % len(words), *words)
    if l_seq:
        body += b"\0" * ((l_seq + 1) // 2) + b"\xff" * l_seq
    return struct.pack("<i", len(body)) + body


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_compress(data, block=0xff00, level=1):
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = len(comp) + 25          # total block size - 1
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp +
                   struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    out.append(BGZF_EOF)
    return b"".join(out)


def bgzf_compress_mt(data, level=6, threads=8, block=0xff00):
    """bgzf_compress on a thread pool (zlib releases the GIL): same bytes as bgzf_compress(data, block, level)."""
    from concurrent.futures import ThreadPoolExecutor
    step = block * 64
    mv = memoryview(data)
    with ThreadPoolExecutor(max(1, int(threads))) as ex:
        parts = list(ex.map(lambda i: bgzf_compress(mv[i:i + step], block, level)[:-len(BGZF_EOF)], range(0, len(data), step)))
    return b"".join(parts) + BGZF_EOF


def records_to_bam(chr_names, chr_lens, records):
    """records: iterable of dicts/tuples accepted by bam_record (as kwargs dicts)."""
    parts = [bam_header(chr_names, chr_lens)]
    for r in records:
        parts.append(bam_record(**r))
    return b"".join(parts)


def pairs_to_bam(chr_names, chr_lens, recs):
    """Vectorised serialiser for the synthetic read generator.

    recs: dict of arrays for 2*n records in stream order:
      ref_id, pos, flag (int), cig_off (2n+1), cig (uint32 words).  Names are 'p<i>'.
    """
    n = len(recs["pos"])
    parts = [bam_header(chr_names, chr_lens)]
    ref_id, pos, flag, off, cig = recs["ref_id"], recs["pos"], recs["flag"], recs["cig_off"], recs["cig"]
    for i in range(n):
        words = cig[off[i]:off[i + 1]]
        nb = b"p%d\0" % (i // 2)
        body = struct.pack("<iiBBHHHiiii", int(ref_id[i]), int(pos[i]), len(nb), 255, 0, len(words), int(flag[i]), 0,
                           int(ref_id[i ^ 1]), int(pos[i ^ 1]), 0) + nb + np.asarray(words, "<u4").tobytes()
        parts.append(struct.pack("<i", len(body)) + body)
    return b"".join(parts)


def reg2bin(beg, end):
    """SAM spec §5.3: the smallest bin that contains [beg, end)."""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def build_bai(bgzf):
    """.bai bytes for a coordinate-sorted BGZF BAM (SAM spec §5.2): binning index, 16 kb linear index and the
    pseudo-bin 37450 (reference byte range + mapped / unmapped counts).  Test helper (pure Python, small inputs)."""
    import bisect
    blocks, parts, co, upos = [], [], 0, 0
    while co < len(bgzf):
        xlen, = struct.unpack_from("<H", bgzf, co + 10)
        bsize, = struct.unpack_from("<H", bgzf, co + 16)           # bgzf_compress writes BC first
        total = bsize + 1
        raw = zlib.decompress(bgzf[co + 12 + xlen: co + total - 8], -15)
        blocks.append((upos, co, len(raw)))
        parts.append(raw)
        upos += len(raw)
        co += total
    data = b"".join(parts)
    ustarts = [b[0] for b in blocks if b[2] > 0]
    cos = [b[1] for b in blocks if b[2] > 0]

    def voff(u):
        if u >= len(data):                 # end of data: start of the block after the last data block
            k = len(ustarts) - 1
            last = [b for b in blocks if b[2] > 0][-1]
            nxt = [b for b in blocks if b[1] > last[1]]
            return (nxt[0][1] << 16) if nxt else ((last[1] << 16) | last[2])
        k = bisect.bisect_right(ustarts, u) - 1
        return (cos[k] << 16) | (u - ustarts[k])

    l_text, = struct.unpack_from("<i", data, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, o)
    o += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, o)
        o += 4 + l_name + 4
    refs = [dict(bins={}, lin={}, beg=None, end=None, mapped=0, unmapped=0) for _ in range(n_ref)]
    n_no_coor = 0
    while o < len(data):
        bs, ref_id, pos = struct.unpack_from("<iii", data, o)
        l_read_name = data[o + 12]
        n_cigar, flag = struct.unpack_from("<HH", data, o + 16)
        words = struct.unpack_from("<%dI" % n_cigar, data, o + 36 + l_read_name)
        rlen = sum(w >> 4 for w in words if (w & 0xf) in (0, 2, 3, 7, 8))
        vb, ve = voff(o), voff(o + 4 + bs)
        o += 4 + bs
        if ref_id < 0:
            n_no_coor += 1
            continue
        r = refs[ref_id]
        end = pos + max(rlen, 1)
        b = reg2bin(pos, end)
        ch = r["bins"].setdefault(b, [])
        if ch and ch[-1][1] == vb:
            ch[-1][1] = ve
        else:
            ch.append([vb, ve])
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            r["lin"].setdefault(w, vb)
        r["beg"] = vb if r["beg"] is None else r["beg"]
        r["end"] = ve
        if flag & 0x4:
            r["unmapped"] += 1
        else:
            r["mapped"] += 1
    out = [b"BAI\1", struct.pack("<i", n_ref)]
    for r in refs:
        nb = len(r["bins"]) + (1 if r["beg"] is not None else 0)
        out.append(struct.pack("<i", nb))
        for b in sorted(r["bins"]):
            out.append(struct.pack("<Ii", b, len(r["bins"][b])))
            for vb, ve in r["bins"][b]:
                out.append(struct.pack("<QQ", vb, ve))
        if r["beg"] is not None:
            out.append(struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", r["beg"], r["end"], r["mapped"], r["unmapped"]))
        n_intv = (max(r["lin"]) + 1) if r["lin"] else 0
        out.append(struct.pack("<i", n_intv))
        last = 0
        for w in range(n_intv):
            last = r["lin"].get(w, last)
            out.append(struct.pack("<Q", last))
    out.append(struct.pack("<Q", n_no_coor))
    return b"".join(out)
