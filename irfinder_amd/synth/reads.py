"""Python side of the synthetic read generator (ctypes over libirfsynth.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

from .bam import bam_header

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SynthModel(C.Structure):
    _fields_ = [
        ("n_chr", C.c_uint32), ("n_tx", C.c_uint32),
        ("chr_len", C.c_void_p), ("chr_cdf", C.c_void_p),
        ("tx_chr", C.c_void_p), ("tx_strand", C.c_void_p), ("tx_ex_off", C.c_void_p),
        ("ex_start", C.c_void_p), ("ex_end", C.c_void_p), ("tx_cdf", C.c_void_p), ("tx_ir_prob", C.c_void_p),
        ("read_len", C.c_uint32), ("stranded", C.c_uint32),
        ("frag_mean", C.c_double), ("frag_sd", C.c_double),
        ("noise_frac", C.c_double), ("indel_frac", C.c_double), ("clip_frac", C.c_double), ("single_frac", C.c_double),
        ("seed", C.c_uint64), ("with_seq", C.c_uint32), ("reserved", C.c_uint32),
    ]


def build_lib(force=False):
    src = os.path.join(_HERE, "reads.cpp")
    so = os.path.join(_HERE, "libirfsynth.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-o", so, src])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libirfsynth.so")
        if not os.path.exists(so):
            build_lib()
        L = C.CDLL(so)
        L.synth_bam_records.restype = C.c_size_t
        L.synth_bam_records.argtypes = [C.POINTER(SynthModel), C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


class ReadSimulator:
    """Deterministic pair generator over an Annotation.  Pair i depends only on (seed, i)."""

    def __init__(self, ann, read_len=150, stranded=False, seed=20240807, frag_mean=300.0, frag_sd=50.0,
                 noise_frac=0.03, indel_frac=0.01, clip_frac=0.02, single_frac=0.0, with_seq=False):
        self.ann = ann
        self._keep = []
        m = SynthModel()
        m.n_chr = len(ann.chr_names)
        m.n_tx = len(ann.tx_chr)

        def hold(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            self._keep.append(a)
            return a.ctypes.data

        lens = np.asarray(ann.chr_lens, np.float64)
        m.chr_len = hold(ann.chr_lens, np.uint32)
        m.chr_cdf = hold(np.cumsum(lens) / lens.sum(), np.float64)
        m.tx_chr = hold(ann.tx_chr, np.int32)
        m.tx_strand = hold(ann.tx_strand, np.int8)
        m.tx_ex_off = hold(ann.tx_ex_off, np.int64)
        m.ex_start = hold(ann.ex_start, np.uint32)
        m.ex_end = hold(ann.ex_end, np.uint32)
        w = np.asarray(ann.tx_weight, np.float64)
        # abundance is per transcript molecule; sampling weight ~ abundance * length keeps long genes realistic
        tlen = np.add.reduceat((ann.ex_end.astype(np.int64) - ann.ex_start.astype(np.int64)), ann.tx_ex_off[:-1]) \
            if m.n_tx else np.zeros(0)
        w = w * np.maximum(tlen, 1)
        cdf = np.cumsum(w)
        if m.n_tx:
            cdf = cdf / cdf[-1]
            cdf[-1] = 1.0
        m.tx_cdf = hold(cdf, np.float64)
        m.tx_ir_prob = hold(ann.tx_ir_prob, np.float64)
        m.read_len, m.stranded = int(read_len), int(bool(stranded))
        m.frag_mean, m.frag_sd = float(frag_mean), float(frag_sd)
        m.noise_frac, m.indel_frac, m.clip_frac, m.single_frac = noise_frac, indel_frac, clip_frac, single_frac
        m.seed, m.with_seq = int(seed), int(with_seq)     # 0: no seq/qual, 1: constant bytes, 2: random bases + binned quals
        self.model = m

    def header(self):
        return bam_header(self.ann.chr_names, self.ann.chr_lens)

    def records(self, first, n):
        """Raw BAM alignment records (no header) of pairs [first, first+n) as a uint8 array."""
        L = lib()
        est = int(n) * (2 * 110 + (2 * (self.model.read_len * 3 // 2 + 8) if self.model.with_seq else 0)) + 4096
        buf = np.empty(est, np.uint8)
        nrec = C.c_uint64()
        need = L.synth_bam_records(C.byref(self.model), int(first), int(n), buf.ctypes.data, est, C.byref(nrec))
        if need > est:
            buf = np.empty(need, np.uint8)
            need2 = L.synth_bam_records(C.byref(self.model), int(first), int(n), buf.ctypes.data, need, C.byref(nrec))
            assert need2 == need
        return buf[:need]

    def bam(self, first, n):
        """A complete uncompressed BAM stream (header + records) as bytes."""
        return self.header() + self.records(first, n).tobytes()
