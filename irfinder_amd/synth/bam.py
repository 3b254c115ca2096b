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
