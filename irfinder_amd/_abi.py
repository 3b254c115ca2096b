"""ctypes mirror of include/irf.h (struct layouts and enums only)."""
import ctypes as C

import numpy as np

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)

IRF_MEM_HOST = 0
IRF_MEM_DEVICE = 1
IRF_FRAG_DIR = 1
IRF_FRAG_TWO = 2

WARNINGS = ["-", "LowCover", "LowSplicing", "MinorIsoform", "NonUniformIntronCover"]

STATUS = {0: "IRF_OK", -1: "IRF_E_ARG", -2: "IRF_E_PARSE", -3: "IRF_E_RANGE", -4: "IRF_E_STATE",
          -5: "IRF_E_DEVICE", -6: "IRF_E_NOMEM", -7: "IRF_E_IO"}

(COL_IR_NONDIR_F64, COL_IR_NONDIR_U32, COL_IR_DIR_F64, COL_IR_DIR_U32, COL_JUNC_KEYS, COL_JUNC_COUNTS,
 COL_SP_CNT, COL_ROI_CNT, COL_CHR_CNT, COL_KNOWN_CNT) = range(10)
ORDER_SORTED, ORDER_ANY = 0, 1      # irf_set_input_order
COL_NO_SYNC = 0x200            # the copy is only queued: Engine.sync() before reading (IRF_COL_NO_SYNC)
COL_KEEP_ORPHANS = 0x100       # OR into an IR column id on the root rank (include/irf.h: IRF_COL_KEEP_ORPHANS)


class IrfParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("skip_flag_mask", C.c_uint32),
        ("pair_long_inclusive", C.c_uint32),
        ("pair_short_inclusive", C.c_uint32),
        ("junc_min_flank", C.c_uint32),
        ("sp_overhang_left", C.c_uint32),
        ("sp_overhang_right", C.c_uint32),
        ("sp_inclusive", C.c_uint32),
        ("trim_centre_pct", C.c_uint32),
        ("pct_interpolate", C.c_uint32),
        ("flank_offset", C.c_uint32),
        ("flank_len", C.c_uint32),
        ("flank_use_measured", C.c_uint32),
        ("dir_ratio", C.c_uint32),
        ("out_precision", C.c_uint32),
        ("warn_lowcover_min", C.c_double),
        ("warn_lowsplice_min", C.c_double),
        ("warn_minor_ratio", C.c_double),
        ("warn_nonuniform_rel", C.c_double),
        ("warn_nonuniform_abs", C.c_double),
        ("junc_event_capacity", C.c_uint64),
        ("keep_depth", C.c_uint32),
        ("mate_join", C.c_uint32),
        ("warn_nonuniform_and", C.c_uint32),
        ("gap_pair_is_intersect", C.c_uint32),
    ]


class IrfBatch(C.Structure):
    _fields_ = [
        ("n_frag", C.c_uint32),
        ("n_blk", C.c_uint32),
        ("frag_chr", C.c_void_p),
        ("frag_flags", C.c_void_p),
        ("frag_start", C.c_void_p),
        ("frag_end", C.c_void_p),
        ("frag_blk_off", C.c_void_p),
        ("blk_start", C.c_void_p),
        ("blk_len", C.c_void_p),
        ("blk_read", C.c_void_p),
        ("mem_kind", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class IrfIrTable(C.Structure):
    _fields_ = [(n, f64p) for n in ("coverage", "intron_depth", "p25", "p50", "p75", "first50", "last50", "irratio")] + \
               [(n, u32p) for n in ("sp_left", "sp_right", "splice_left", "splice_right", "splice_exact")] + \
               [("warning", u8p)]


class IrfResults(C.Structure):
    _fields_ = [
        ("directionality", C.c_int32),
        ("n_chr", C.c_uint32),
        ("chr_cnt", u64p),
        ("n_roi", C.c_uint32),
        ("roi_cnt", u64p),
        ("n_sp", C.c_uint32),
        ("sp_chr", u32p),
        ("sp_pos", u32p),
        ("sp_cnt", u32p),
        ("n_junc", C.c_uint64),
        ("junc_chr", u32p),
        ("junc_left", u32p),
        ("junc_right", u32p),
        ("junc_cnt", u32p),
        ("junc_refdir", u8p),
        ("n_intron", C.c_uint32),
        ("has_dir", C.c_uint32),
        ("nondir", IrfIrTable),
        ("dir", IrfIrTable),
        ("n_fragments", C.c_uint64),
        ("n_blocks", C.c_uint64),
        ("n_junction_events", C.c_uint64),
    ]


class IrfReaderStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_records", "n_nucleotides", "n_singles", "n_pairs", "n_pairs_short",
                                          "n_pairs_intersect", "n_pairs_long", "n_skipped", "n_error")]


class IrfDreaderTiming(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("h2d_ms", "inflate_ms", "crc_ms", "chain_ms", "assemble_ms")] + \
               [("chain_rounds", C.c_uint32)] + [(n, C.c_uint64) for n in ("bytes_in", "bytes_inflated", "n_members")]


class IrfLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("plane_positions", "full_positions", "measured_positions", "junction_capacity")]


class IrfTiming(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("count_ms", "scan_ms", "stats_ms", "junction_ms", "table_ms",
                                         "total_finalize_ms")] + \
               [("n_count_launches", C.c_uint32), ("n_radix_passes", C.c_uint32), ("n_sort_records", C.c_uint64)]


def _arr(ptr, n, dtype):
    """Copy n elements behind a ctypes pointer into a numpy array."""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


IR_F64 = ("coverage", "intron_depth", "p25", "p50", "p75", "first50", "last50", "irratio")
IR_U32 = ("sp_left", "sp_right", "splice_left", "splice_right", "splice_exact")


def table_to_dict(t, n):
    d = {k: _arr(getattr(t, k), n, np.float64) for k in IR_F64}
    d.update({k: _arr(getattr(t, k), n, np.uint32) for k in IR_U32})
    d["warning"] = _arr(t.warning, n, np.uint8)
    return d


def results_to_dict(r):
    """Deep-copy an IrfResults view into plain numpy (safe after the ctx goes away)."""
    out = {
        "directionality": int(r.directionality),
        "chr_cnt": _arr(r.chr_cnt, 2 * r.n_chr, np.uint64).reshape(r.n_chr, 2),
        "roi_cnt": _arr(r.roi_cnt, 2 * r.n_roi, np.uint64).reshape(2, r.n_roi),
        "sp_chr": _arr(r.sp_chr, r.n_sp, np.uint32),
        "sp_pos": _arr(r.sp_pos, r.n_sp, np.uint32),
        "sp_cnt": _arr(r.sp_cnt, 2 * r.n_sp, np.uint32).reshape(2, r.n_sp),
        "junc_chr": _arr(r.junc_chr, r.n_junc, np.uint32),
        "junc_left": _arr(r.junc_left, r.n_junc, np.uint32),
        "junc_right": _arr(r.junc_right, r.n_junc, np.uint32),
        "junc_cnt": _arr(r.junc_cnt, 2 * r.n_junc, np.uint32).reshape(2, r.n_junc),
        "junc_refdir": _arr(r.junc_refdir, r.n_junc, np.uint8),
        "n_intron": int(r.n_intron),
        "has_dir": bool(r.has_dir),
        "nondir": table_to_dict(r.nondir, r.n_intron),
        "dir": table_to_dict(r.dir, r.n_intron if r.has_dir else 0),
        "n_fragments": int(r.n_fragments),
        "n_blocks": int(r.n_blocks),
        "n_junction_events": int(r.n_junction_events),
    }
    return out


class Batch:
    """Host-side SoA batch of fragments (numpy arrays) + its irf_batch view."""

    FIELDS = (("frag_chr", np.uint32), ("frag_flags", np.uint8), ("frag_start", np.uint32), ("frag_end", np.uint32),
              ("frag_blk_off", np.uint32), ("blk_start", np.uint32), ("blk_len", np.uint32), ("blk_read", np.uint8))

    def __init__(self, **arrays):
        for name, dt in self.FIELDS:
            setattr(self, name, np.ascontiguousarray(arrays[name], dtype=dt))
        self.n_frag = int(self.frag_chr.shape[0])
        self.n_blk = int(self.blk_start.shape[0])
        assert self.frag_blk_off.shape[0] == self.n_frag + 1
        assert int(self.frag_blk_off[-1]) == self.n_blk if self.n_frag else True

    @classmethod
    def empty(cls):
        return cls(frag_chr=[], frag_flags=[], frag_start=[], frag_end=[], frag_blk_off=[0], blk_start=[], blk_len=[],
                   blk_read=[])

    @classmethod
    def from_struct(cls, b):
        """Copy a host irf_batch (e.g. from a reader) into numpy."""
        def a(p, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()
        return cls(frag_chr=a(b.frag_chr, b.n_frag, np.uint32), frag_flags=a(b.frag_flags, b.n_frag, np.uint8),
                   frag_start=a(b.frag_start, b.n_frag, np.uint32), frag_end=a(b.frag_end, b.n_frag, np.uint32),
                   frag_blk_off=a(b.frag_blk_off, b.n_frag + 1, np.uint32), blk_start=a(b.blk_start, b.n_blk, np.uint32),
                   blk_len=a(b.blk_len, b.n_blk, np.uint32), blk_read=a(b.blk_read, b.n_blk, np.uint8))

    def struct(self):
        b = IrfBatch()
        b.n_frag, b.n_blk = self.n_frag, self.n_blk
        for name, _ in self.FIELDS:
            setattr(b, name, getattr(self, name).ctypes.data)
        b.mem_kind = IRF_MEM_HOST
        return b

    def select(self, idx):
        """Sub-batch of the fragments idx (any order) — for permutation / shard tests."""
        idx = np.asarray(idx, dtype=np.int64)
        off = self.frag_blk_off.astype(np.int64)
        cnt = off[idx + 1] - off[idx]
        new_off = np.zeros(len(idx) + 1, np.int64)
        np.cumsum(cnt, out=new_off[1:])
        # gather block indices
        base = np.repeat(off[idx], cnt)
        within = np.arange(int(new_off[-1]), dtype=np.int64) - np.repeat(new_off[:-1], cnt)
        bidx = base + within
        return Batch(frag_chr=self.frag_chr[idx], frag_flags=self.frag_flags[idx], frag_start=self.frag_start[idx],
                     frag_end=self.frag_end[idx], frag_blk_off=new_off.astype(np.uint32), blk_start=self.blk_start[bidx],
                     blk_len=self.blk_len[bidx], blk_read=self.blk_read[bidx])

    def equals(self, o):
        return all(np.array_equal(getattr(self, n), getattr(o, n)) for n, _ in self.FIELDS)
