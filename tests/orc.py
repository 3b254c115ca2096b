"""ctypes wrapper of oracle/liborc.so — the CPU oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from irfinder_amd._abi import (Batch, IrfBatch, IrfParams, IrfReaderStats, IrfResults, results_to_dict)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liborc.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(IrfParams), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_chr_map_update.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32)]
        L.orc_process_batch.argtypes = [C.c_void_p, C.POINTER(IrfBatch)]
        L.orc_process_bam.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_reader_stats.argtypes = [C.c_void_p, C.POINTER(IrfReaderStats)]
        L.orc_finalize.argtypes = [C.c_void_p]
        L.orc_get_results.argtypes = [C.c_void_p, C.POINTER(IrfResults)]
        L.orc_chr_name.restype = C.c_char_p
        L.orc_chr_name.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_n_chr_slots.restype = C.c_uint32
        L.orc_n_chr_slots.argtypes = [C.c_void_p]
        L.orc_write_outputs.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_get_depth.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_cigar2block.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_stats_from_values.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(IrfParams), C.POINTER(C.c_double)]
        L.orc_bam_to_batch.restype = C.c_void_p
        L.orc_bam_to_batch.argtypes = [C.POINTER(IrfParams), C.c_char_p, C.c_size_t, C.POINTER(IrfBatch),
                                       C.POINTER(C.c_int)]
        L.orc_dump_stats.argtypes = [C.c_void_p, C.POINTER(IrfReaderStats)]
        L.orc_dump_free.argtypes = [C.c_void_p]
        L.orc_params_default.argtypes = [C.POINTER(IrfParams)]
        _LIB = L
    return _LIB


def default_params(**over):
    p = IrfParams()
    lib().orc_params_default(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


CIGAR_OPS = "MIDNSHP=X"


def cigar_words(cigar):
    """'50M1000N50M' -> uint32 BAM cigar words."""
    out, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | CIGAR_OPS.index(ch))
            num = ""
    return np.array(out, dtype=np.uint32)


def cigar2block(cigar):
    w = cigar_words(cigar) if isinstance(cigar, str) else np.asarray(cigar, np.uint32)
    starts = np.zeros(64, np.uint32)
    lens = np.zeros(64, np.uint32)
    nb, gl = C.c_uint32(), C.c_uint32()
    rc = lib().orc_cigar2block(w.ctypes.data, len(w), starts.ctypes.data, lens.ctypes.data, 64, C.byref(nb), C.byref(gl))
    assert rc == 0
    return starts[:nb.value].tolist(), lens[:nb.value].tolist(), gl.value


def stats_from_values(values, params=None):
    v = np.ascontiguousarray(values, np.uint32)
    p = params or default_params()
    out = (C.c_double * 6)()
    lib().orc_stats_from_values(v.ctypes.data, len(v), C.byref(p), out)
    return dict(zip(("coverage", "mean", "trimmed", "p25", "p50", "p75"), list(out)))


def bam_to_batch(bam_bytes, params=None):
    """Reader only: raw BAM bytes -> (Batch, reader stats dict, status)."""
    p = params or default_params()
    b = IrfBatch()
    st = C.c_int()
    h = lib().orc_bam_to_batch(C.byref(p), bam_bytes, len(bam_bytes), C.byref(b), C.byref(st))
    try:
        batch = Batch.from_struct(b)
        rs = IrfReaderStats()
        lib().orc_dump_stats(h, C.byref(rs))
        stats = {n: int(getattr(rs, n)) for n, _ in IrfReaderStats._fields_}
    finally:
        lib().orc_dump_free(h)
    return batch, stats, st.value


class Oracle:
    """One sample run through the CPU restatement."""

    def __init__(self, ref_texts, params=None):
        """ref_texts: (cover_bed, sj_ref, read_continues_ref, roi_bed) as bytes/str."""
        self.L = lib()
        self.params = params or default_params()
        bufs = [t.encode() if isinstance(t, str) else bytes(t) for t in ref_texts]
        self._bufs = bufs
        arr = (C.c_char_p * 4)(*bufs)
        lens = (C.c_size_t * 4)(*[len(b) for b in bufs])
        self.h = self.L.orc_create(C.byref(self.params), arr, lens)
        if not self.h:
            raise ValueError("oracle: reference load failed")

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle rc=%d: %s" % (rc, self.L.orc_last_error(self.h).decode()))

    def chr_map_update(self, names, lens):
        n = len(names)
        arr = (C.c_char_p * n)(*[s.encode() for s in names])
        la = (C.c_uint32 * n)(*[int(x) for x in lens])
        self._chk(self.L.orc_chr_map_update(self.h, n, arr, la))

    def process_batch(self, batch):
        b = batch.struct()
        self._chk(self.L.orc_process_batch(self.h, C.byref(b)))

    def process_bam(self, bam_bytes):
        self._chk(self.L.orc_process_bam(self.h, bam_bytes, len(bam_bytes)))

    def reader_stats(self):
        rs = IrfReaderStats()
        self.L.orc_reader_stats(self.h, C.byref(rs))
        return {n: int(getattr(rs, n)) for n, _ in IrfReaderStats._fields_}

    def finalize(self):
        self._chk(self.L.orc_finalize(self.h))

    def results(self):
        r = IrfResults()
        self._chk(self.L.orc_get_results(self.h, C.byref(r)))
        d = results_to_dict(r)
        d["chr_names"] = [self.L.orc_chr_name(self.h, i).decode() for i in range(self.L.orc_n_chr_slots(self.h))]
        return d

    def write_outputs(self, outdir):
        self._chk(self.L.orc_write_outputs(self.h, outdir.encode()))

    def depth(self, chr_id, start, end, plane):
        out = np.zeros(end - start, np.uint32)
        self._chk(self.L.orc_get_depth(self.h, chr_id, start, end, plane, out.ctypes.data))
        return out
