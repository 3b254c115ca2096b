"""BAI index reader for BAI-driven chromosome sharding (SURVEY §8 e, App. A; SAM spec §5.2).

north_star shards the work "by chromosome range from the BAI index": a rank needs (i) an estimate of the work per
reference to pack chromosomes onto GPUs (pseudo-bin 37450 carries n_mapped / n_unmapped per reference) and (ii) the
byte range of its references inside the coordinate-sorted BGZF file, so that it inflates and parses only those
(pseudo-bin 37450's first chunk = virtual offsets of the reference's first and past-the-last record).  The reference
has no counterpart (one process reads one stream).

Virtual offset = coffset << 16 | uoffset: coffset = file offset of a BGZF block, uoffset = offset inside its inflated
data.
"""
import struct
import zlib

import numpy as np

PSEUDO_BIN = 37450


class BaiRef:
    __slots__ = ("bins", "linear", "n_mapped", "n_unmapped", "ref_beg", "ref_end")

    def __init__(self):
        self.bins = {}              # bin -> [(beg, end), ...] virtual offsets
        self.linear = np.zeros(0, np.uint64)
        self.n_mapped = None        # from the pseudo-bin; None if the index does not carry it
        self.n_unmapped = None
        self.ref_beg = None
        self.ref_end = None


class Bai:
    def __init__(self, refs, n_no_coor=None):
        self.refs = refs
        self.n_no_coor = n_no_coor

    @property
    def n_ref(self):
        return len(self.refs)

    def weights(self):
        """Expected work per reference for shard_chromosomes: mapped reads where the index has the pseudo-bin, else
        the compressed bytes the reference spans (difference of the first / last linear-index offsets)."""
        w = np.zeros(self.n_ref, np.float64)
        for i, r in enumerate(self.refs):
            if r.n_mapped is not None:
                w[i] = r.n_mapped
            elif len(r.linear):
                nz = r.linear[r.linear > 0]
                w[i] = float((int(nz.max()) >> 16) - (int(nz.min()) >> 16) + 1) if len(nz) else 0.0
        return w

    def ref_range(self, ref_id):
        """(beg, end) virtual offsets of all records of a reference, or None when it has none."""
        r = self.refs[ref_id]
        if r.ref_beg is not None:
            return (r.ref_beg, r.ref_end) if r.ref_end > r.ref_beg else None
        chunks = [c for b, cs in r.bins.items() if b != PSEUDO_BIN for c in cs]
        if not chunks:
            return None
        return min(c[0] for c in chunks), max(c[1] for c in chunks)


def parse_bai(data):
    """bytes of a .bai file -> Bai.  Raises ValueError on a malformed index."""
    if len(data) < 8 or data[:4] != b"BAI\1":
        raise ValueError("BAI: bad magic")
    o = 4
    n_ref, = struct.unpack_from("<i", data, o)
    o += 4
    if n_ref < 0:
        raise ValueError("BAI: negative n_ref")
    refs = []
    try:
        for _ in range(n_ref):
            r = BaiRef()
            n_bin, = struct.unpack_from("<i", data, o)
            o += 4
            for _ in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", data, o)
                o += 8
                chunks = [struct.unpack_from("<QQ", data, o + 16 * k) for k in range(n_chunk)]
                o += 16 * n_chunk
                if b == PSEUDO_BIN and n_chunk >= 2:
                    r.ref_beg, r.ref_end = chunks[0]
                    r.n_mapped, r.n_unmapped = chunks[1]
                r.bins[b] = chunks
            n_intv, = struct.unpack_from("<i", data, o)
            o += 4
            r.linear = np.frombuffer(data, dtype="<u8", count=n_intv, offset=o).copy()
            o += 8 * n_intv
            refs.append(r)
    except struct.error:
        raise ValueError("BAI: truncated index")
    n_no_coor = struct.unpack_from("<Q", data, o)[0] if len(data) >= o + 8 else None
    return Bai(refs, n_no_coor)


def read_virtual_range(bgzf, beg, end):
    """Inflated bytes of a BGZF file (bytes-like) between two virtual offsets: what a rank feeds its reader for one
    reference (after the header, raw).  Host-side plumbing: zlib per block."""
    out = []
    co, cend, uend = beg >> 16, end >> 16, end & 0xffff
    skip = beg & 0xffff
    while co < cend or (co == cend and uend > 0):
        if co + 18 > len(bgzf):
            raise ValueError("BGZF: range beyond the end of the file")
        xlen, = struct.unpack_from("<H", bgzf, co + 10)
        bsize = None
        x = 0
        while x + 4 <= xlen:
            si1, si2, slen = struct.unpack_from("<BBH", bgzf, co + 12 + x)
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize, = struct.unpack_from("<H", bgzf, co + 12 + x + 4)
            x += 4 + slen
        if bsize is None:
            raise ValueError("BGZF: missing BC subfield")
        total = bsize + 1
        raw = zlib.decompress(bytes(bgzf[co + 12 + xlen: co + total - 8]), -15)
        stop = uend if co == cend else len(raw)
        out.append(raw[skip:stop])
        skip = 0
        co += total
    return b"".join(out)
