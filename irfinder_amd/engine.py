"""Python host mirror of the C ABI (include/irf.h) over libirfhip.so.

The library is the product; this module only marshals.  There is no CPU
fallback: if libirfhip.so is missing, or no HIP device is present, the calls
fail loudly (RuntimeError / IrfError).
"""
import ctypes as C
import os

import numpy as np

from ._abi import (STATUS, Batch, IrfBatch, IrfDreaderTiming, IrfLayout, IrfParams, IrfReaderStats, IrfResults, IrfTiming,
                   IRF_MEM_DEVICE, IRF_MEM_HOST, results_to_dict)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IRF_LIB_PATH") or os.path.join(_HERE, "libirfhip.so")   # IRF_LIB_PATH: kernel-tuning builds only
_LIB = None


class IrfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "IRF_E_?"), code, msg))
        self.code = code


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in one process
    do not both see the GPU, so when torch is installed its copy is mapped first and libirfhip.so binds to it (the
    loader matches the SONAME).  Set IRF_HIP_RUNTIME=system to keep /opt/rocm's runtime (processes without torch)."""
    if os.environ.get("IRF_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Loads libirfhip.so (built in-tree by __graft_entry__.build() / make -C irfinder_amd/csrc)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libirfhip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C irfinder_amd/csrc`" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.irf_params_default.argtypes = [C.POINTER(IrfParams)]
    L.irf_params_default.restype = None
    L.irf_ref_load_files.argtypes = [C.c_char_p] * 4 + [C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.irf_ref_load_mem.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.irf_ref_free.argtypes = [vp]
    L.irf_ref_free.restype = None
    for n in ("irf_ref_n_intron", "irf_ref_n_junction", "irf_ref_n_point", "irf_ref_n_roi"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = u32
    L.irf_create.argtypes = [C.POINTER(IrfParams), vp, C.c_int, C.POINTER(vp)]
    L.irf_destroy.argtypes = [vp]
    L.irf_destroy.restype = None
    L.irf_last_error.argtypes = [vp]
    L.irf_last_error.restype = C.c_char_p
    L.irf_chr_map_update.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(u32), C.c_void_p]
    L.irf_chr_map_update_ranges.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(u32), C.c_void_p, C.c_void_p, C.c_void_p]
    L.irf_submit_batch.argtypes = [vp, C.POINTER(IrfBatch), C.POINTER(u64)]
    L.irf_wait.argtypes = [vp, u64]
    L.irf_finalize.argtypes = [vp]
    L.irf_finalize_directional.argtypes = [vp, C.c_int32]
    L.irf_get_votes.argtypes = [vp, C.POINTER(u64)]
    L.irf_get_results.argtypes = [vp, C.POINTER(IrfResults)]
    L.irf_chr_name.argtypes = [vp, u32]
    L.irf_chr_name.restype = C.c_char_p
    L.irf_n_chr_slots.argtypes = [vp]
    L.irf_n_chr_slots.restype = u32
    L.irf_write_outputs.argtypes = [vp, C.c_char_p]
    L.irf_get_depth.argtypes = [vp, u32, u32, u32, u32, C.c_void_p]
    L.irf_reset.argtypes = [vp]
    L.irf_export_device.argtypes = [vp, u32, C.c_void_p, C.POINTER(u64)]
    L.irf_import_device.argtypes = [vp, u32, C.c_void_p, u64]
    L.irf_export_junctions.argtypes = [vp, C.POINTER(u64)] + [C.c_void_p] * 5
    L.irf_import_junctions.argtypes = [vp, u64] + [C.c_void_p] * 5
    L.irf_gather.argtypes = [C.POINTER(vp), u32, u32]
    L.irf_get_timing.argtypes = [vp, C.POINTER(IrfTiming)]
    L.irf_get_layout.argtypes = [vp, C.POINTER(IrfLayout)]
    L.irf_reader_create.argtypes = [C.POINTER(IrfParams), C.POINTER(vp)]
    L.irf_reader_destroy.argtypes = [vp]
    L.irf_reader_destroy.restype = None
    L.irf_reader_last_error.argtypes = [vp]
    L.irf_reader_last_error.restype = C.c_char_p
    L.irf_reader_feed.argtypes = [vp, C.c_void_p, C.c_size_t, C.c_int]
    L.irf_reader_set_threads.argtypes = [vp, C.c_int]
    L.irf_reader_header.argtypes = [vp, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(u32))]
    L.irf_reader_batch.argtypes = [vp, C.POINTER(IrfBatch)]
    L.irf_reader_clear_batch.argtypes = [vp]
    L.irf_reader_clear_batch.restype = None
    L.irf_reader_finish.argtypes = [vp]
    L.irf_reader_get_stats.argtypes = [vp, C.POINTER(IrfReaderStats)]
    L.irf_cigar2block.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p, u32, C.POINTER(u32), C.POINTER(u32)]
    L.irf_sync.argtypes = [vp]
    L.irf_set_input_order.argtypes = [vp, C.c_uint32]
    L.irf_dreader_create.argtypes = [C.POINTER(IrfParams), C.c_int, C.POINTER(vp)]
    L.irf_dreader_destroy.argtypes = [vp]
    L.irf_dreader_destroy.restype = None
    L.irf_dreader_last_error.argtypes = [vp]
    L.irf_dreader_last_error.restype = C.c_char_p
    L.irf_dreader_feed.argtypes = [vp, C.c_void_p, C.c_size_t]
    L.irf_dreader_set_pipeline.argtypes = [vp, C.c_int]
    L.irf_dreader_member_slots.argtypes = [vp]
    L.irf_dreader_member_slots.restype = C.c_uint32
    L.irf_dreader_header.argtypes = [vp, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(u32))]
    L.irf_dreader_batch.argtypes = [vp, C.POINTER(IrfBatch)]
    L.irf_dreader_finish.argtypes = [vp]
    L.irf_dreader_get_stats.argtypes = [vp, C.POINTER(IrfReaderStats)]
    L.irf_dreader_get_timing.argtypes = [vp, C.POINTER(IrfDreaderTiming)]
    L.irf_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.irf_gpu_bgzf_inflate.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.irf_gpu_bgzf_inflate_noalloc.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    _LIB = L
    return L


def default_params(**over):
    p = IrfParams()
    lib().irf_params_default(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def cigar2block(words):
    w = np.ascontiguousarray(words, np.uint32)
    starts = np.zeros(max(len(w), 1), np.uint32)
    lens = np.zeros(max(len(w), 1), np.uint32)
    nb, gl = C.c_uint32(), C.c_uint32()
    rc = lib().irf_cigar2block(w.ctypes.data, len(w), starts.ctypes.data, lens.ctypes.data, len(starts), C.byref(nb),
                               C.byref(gl))
    if rc != 0:
        raise IrfError(rc, "irf_cigar2block")
    return starts[:nb.value].tolist(), lens[:nb.value].tolist(), gl.value


class RefTables:
    """The four reference files (ref-cover.bed, ref-sj.ref, ref-read-continues.ref, ref-ROI.bed)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_texts(cls, texts):
        L = lib()
        bufs = [t.encode() if isinstance(t, str) else bytes(t) for t in texts]
        arr = (C.c_char_p * 4)(*bufs)
        lens = (C.c_size_t * 4)(*[len(b) for b in bufs])
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = L.irf_ref_load_mem(arr, lens, C.byref(h), err, 256)
        if rc != 0:
            raise IrfError(rc, err.value.decode())
        return cls(h)

    @classmethod
    def from_files(cls, cover, sj, read_continues, roi):
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = L.irf_ref_load_files(cover.encode(), sj.encode(), read_continues.encode(), roi.encode(), C.byref(h), err, 256)
        if rc != 0:
            raise IrfError(rc, err.value.decode())
        return cls(h)

    def counts(self):
        L = lib()
        return dict(n_intron=L.irf_ref_n_intron(self.h), n_junction=L.irf_ref_n_junction(self.h),
                    n_point=L.irf_ref_n_point(self.h), n_roi=L.irf_ref_n_roi(self.h))

    def close(self):
        if self.h:
            lib().irf_ref_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reader:
    """Host BAM reader (BAM2blocks replacement): bytes in, SoA fragment batches out."""

    def __init__(self, params=None, threads=1):
        self.L = lib()
        self.params = params or default_params()
        self.h = C.c_void_p()
        rc = self.L.irf_reader_create(C.byref(self.params), C.byref(self.h))
        if rc != 0:
            raise IrfError(rc, "irf_reader_create")
        if threads != 1:
            self._chk(self.L.irf_reader_set_threads(self.h, int(threads)))

    def _chk(self, rc):
        if rc != 0:
            raise IrfError(rc, self.L.irf_reader_last_error(self.h).decode())

    def feed(self, data, bgzf=False):
        if isinstance(data, np.ndarray):
            ptr, n = data.ctypes.data, data.nbytes
        else:
            data = bytes(data)
            ptr, n = C.cast(C.c_char_p(data), C.c_void_p), len(data)
        self._chk(self.L.irf_reader_feed(self.h, ptr, n, 1 if bgzf else 0))

    def finish(self):
        self._chk(self.L.irf_reader_finish(self.h))

    def header(self):
        names = C.POINTER(C.c_char_p)()
        lens = C.POINTER(C.c_uint32)()
        n = self.L.irf_reader_header(self.h, C.byref(names), C.byref(lens))
        if n < 0:
            raise IrfError(n, "BAM header not complete")
        return [names[i].decode() for i in range(n)], [int(lens[i]) for i in range(n)]

    def batch_struct(self):
        b = IrfBatch()
        self._chk(self.L.irf_reader_batch(self.h, C.byref(b)))
        return b

    def take_batch(self):
        """Copy of the fragments assembled so far; clears the reader's buffers."""
        out = Batch.from_struct(self.batch_struct())
        self.L.irf_reader_clear_batch(self.h)
        return out

    def stats(self):
        rs = IrfReaderStats()
        self._chk(self.L.irf_reader_get_stats(self.h, C.byref(rs)))
        return {n: int(getattr(rs, n)) for n, _ in IrfReaderStats._fields_}

    def close(self):
        if self.h:
            self.L.irf_reader_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _buf(data):
    """(pointer, byte count, keep-alive object) of bytes-like input."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data)
        return a.ctypes.data, a.nbytes, a
    data = bytes(data)
    return C.cast(C.c_char_p(data), C.c_void_p), len(data), data


def gpu_bgzf_inflate(data, device=0):
    """Whole BGZF members -> inflated bytes, inflated and CRC-checked on the device (irf_gpu_bgzf_inflate)."""
    L = lib()
    ptr, n, keep = _buf(data)
    need = C.c_size_t()
    rc = L.irf_gpu_bgzf_inflate(int(device), ptr, n, None, 0, C.byref(need))
    if rc not in (0, -3):
        raise IrfError(rc, "irf_gpu_bgzf_inflate")
    out = np.empty(need.value, np.uint8)
    if need.value:
        rc = L.irf_gpu_bgzf_inflate(int(device), ptr, n, out.ctypes.data, out.nbytes, C.byref(need))
        if rc != 0:
            raise IrfError(rc, "irf_gpu_bgzf_inflate")
    return out


NOALLOC_MAX_BYTES = 8 << 20       # include/irf.h: IRF_NOALLOC_MAX_BYTES / IRF_NOALLOC_MAX_MEMBERS
NOALLOC_MAX_MEMBERS = 8192


def gpu_bgzf_inflate_noalloc(data, out_device_ptr, out_cap, device=0):
    """irf_gpu_bgzf_inflate_noalloc: whole BGZF members (host) -> inflated bytes in CALLER-owned device memory
    (out_cap >= inflated size + 64); allocates nothing.  Returns the inflated size."""
    L = lib()
    ptr, n, keep = _buf(data)
    need = C.c_size_t()
    rc = L.irf_gpu_bgzf_inflate_noalloc(int(device), ptr, n, C.c_void_p(out_device_ptr), int(out_cap), C.byref(need))
    if rc != 0:
        raise IrfError(rc, "irf_gpu_bgzf_inflate_noalloc (needs %d bytes)" % need.value)
    return need.value


class DeviceReader:
    """Device reader (irf_dreader_*): compressed BGZF bytes in, DEVICE-resident fragment batches out."""

    def __init__(self, params=None, device=0, pipeline=False):
        """pipeline: irf_dreader_set_pipeline — the batch after feed k is that of feed k - 1, the last one appears after
        finish(); the copy of a feed overlaps the inflate of the feed before it."""
        self.L = lib()
        self.params = params or default_params()
        self.h = C.c_void_p()
        rc = self.L.irf_dreader_create(C.byref(self.params), int(device), C.byref(self.h))
        if rc != 0:
            raise IrfError(rc, "irf_dreader_create failed (no HIP device or bad parameters)")
        if pipeline:
            self._chk(self.L.irf_dreader_set_pipeline(self.h, 1))

    def _chk(self, rc):
        if rc != 0:
            raise IrfError(rc, self.L.irf_dreader_last_error(self.h).decode())

    def feed(self, data):
        ptr, n, keep = _buf(data)
        self._chk(self.L.irf_dreader_feed(self.h, ptr, n))

    def finish(self):
        self._chk(self.L.irf_dreader_finish(self.h))

    def member_slots(self):
        """BGZF members the device inflates at once (irf_dreader_member_slots)."""
        return int(self.L.irf_dreader_member_slots(self.h))

    def header(self):
        names = C.POINTER(C.c_char_p)()
        lens = C.POINTER(C.c_uint32)()
        n = self.L.irf_dreader_header(self.h, C.byref(names), C.byref(lens))
        if n < 0:
            raise IrfError(n, "BAM header not complete")
        return [names[i].decode() for i in range(n)], [int(lens[i]) for i in range(n)]

    def batch_struct(self):
        """irf_batch (device pointers) of the last feed; valid until the next-but-one feed."""
        b = IrfBatch()
        self._chk(self.L.irf_dreader_batch(self.h, C.byref(b)))
        return b

    def batch_to_host(self):
        """Copy of the last feed's fragments in host memory (tests)."""
        b = self.batch_struct()
        arrays = {}
        for name, dt in Batch.FIELDS:
            n = b.n_blk if name.startswith("blk_") else b.n_frag + (1 if name == "frag_blk_off" else 0)
            a = np.zeros(n, dt)
            if name == "frag_blk_off" and b.n_frag == 0:
                arrays[name] = a
                continue
            if a.nbytes:
                rc = self.L.irf_copy_to_host(a.ctypes.data, getattr(b, name), a.nbytes)
                if rc != 0:
                    raise IrfError(rc, "irf_copy_to_host")
            arrays[name] = a
        return Batch(**arrays)

    def stats(self):
        rs = IrfReaderStats()
        self._chk(self.L.irf_dreader_get_stats(self.h, C.byref(rs)))
        return {n: int(getattr(rs, n)) for n, _ in IrfReaderStats._fields_}

    def timing(self):
        t = IrfDreaderTiming()
        self._chk(self.L.irf_dreader_get_timing(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in IrfDreaderTiming._fields_ if n != "reserved"}

    def close(self):
        if self.h:
            self.L.irf_dreader_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One irf_ctx: mirrors ReadBlockProcessor's ChrMapUpdate / ProcessBlocks (batched) + the end-of-run writers."""

    def __init__(self, ref, params=None, device=0):
        self.L = lib()
        self.params = params or default_params()
        self.h = C.c_void_p()
        rc = self.L.irf_create(C.byref(self.params), ref.h, int(device), C.byref(self.h))
        if rc != 0:
            raise IrfError(rc, "irf_create failed (no HIP device, or bad parameters)")
        self.device = device

    def _chk(self, rc):
        if rc != 0:
            raise IrfError(rc, self.L.irf_last_error(self.h).decode())

    def chr_map_update(self, names, lens, owned=None, own_start=None, own_end=None):
        """owned: 0/1 per BAM reference; own_start / own_end: the part [start, end) of every owned reference this context
        owns (irf_chr_map_update_ranges; None = whole references)."""
        n = len(names)
        arr = (C.c_char_p * n)(*[s.encode() for s in names])
        la = (C.c_uint32 * n)(*[int(x) for x in lens])
        ow = None
        if owned is not None:
            self._owned = np.ascontiguousarray(owned, np.uint8)
            ow = self._owned.ctypes.data
        if own_start is None and own_end is None:
            self._chk(self.L.irf_chr_map_update(self.h, n, arr, la, ow))
            return
        self._own = (np.ascontiguousarray(own_start, np.uint32), np.ascontiguousarray(own_end, np.uint32))
        assert len(self._own[0]) == n and len(self._own[1]) == n
        self._chk(self.L.irf_chr_map_update_ranges(self.h, n, arr, la, ow, self._own[0].ctypes.data, self._own[1].ctypes.data))

    def submit(self, batch):
        """batch: _abi.Batch (host) or an IrfBatch struct (host or device pointers)."""
        b = batch.struct() if isinstance(batch, Batch) else batch
        bid = C.c_uint64()
        self._chk(self.L.irf_submit_batch(self.h, C.byref(b), C.byref(bid)))
        return bid.value

    def wait(self, bid):
        self._chk(self.L.irf_wait(self.h, bid))

    def process(self, batch):
        bid = self.submit(batch)
        self.wait(bid)

    def sync(self):
        self._chk(self.L.irf_sync(self.h))

    def set_input_order(self, order):
        """ORDER_SORTED (default) or ORDER_ANY: batches in no coordinate order are binned by coordinate on the device
        before they are counted (include/irf.h: irf_set_input_order)."""
        self._chk(self.L.irf_set_input_order(self.h, order))

    def finalize(self):
        self._chk(self.L.irf_finalize(self.h))

    def votes(self):
        v = (C.c_uint64 * 2)()
        self._chk(self.L.irf_get_votes(self.h, v))
        return int(v[0]), int(v[1])

    def finalize_directional(self, directionality):
        self._chk(self.L.irf_finalize_directional(self.h, int(directionality)))

    def reset(self):
        self._chk(self.L.irf_reset(self.h))

    def results(self):
        r = IrfResults()
        self._chk(self.L.irf_get_results(self.h, C.byref(r)))
        d = results_to_dict(r)
        d["chr_names"] = [self.L.irf_chr_name(self.h, i).decode() for i in range(self.L.irf_n_chr_slots(self.h))]
        return d

    def write_outputs(self, outdir):
        self._chk(self.L.irf_write_outputs(self.h, outdir.encode()))

    def depth(self, chr_id, start, end, plane):
        out = np.zeros(end - start, np.uint32)
        self._chk(self.L.irf_get_depth(self.h, chr_id, start, end, plane, out.ctypes.data))
        return out

    def timing(self):
        t = IrfTiming()
        self._chk(self.L.irf_get_timing(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in IrfTiming._fields_ if n != "reserved"}

    def layout(self):
        t = IrfLayout()
        self._chk(self.L.irf_get_layout(self.h, C.byref(t)))
        return {n: int(getattr(t, n)) for n, _ in IrfLayout._fields_}

    def export_size(self, what):
        n = C.c_uint64()
        self._chk(self.L.irf_export_device(self.h, what, None, C.byref(n)))
        return n.value

    def export_device(self, what, dst_ptr):
        n = C.c_uint64()
        self._chk(self.L.irf_export_device(self.h, what, C.c_void_p(dst_ptr), C.byref(n)))
        return n.value

    def import_device(self, what, src_ptr, n):
        """irf_import_device: add a column another shard exported (already on this device) into this context's table."""
        self._chk(self.L.irf_import_device(self.h, what, C.c_void_p(src_ptr), int(n)))

    def export_junctions(self):
        """This context's novel junctions as coordinates: dict of uint32 arrays chr, left, right, cnt_neg, cnt_pos."""
        n = C.c_uint64(0)
        self._chk(self.L.irf_export_junctions(self.h, C.byref(n), None, None, None, None, None))
        a = [np.zeros(n.value, np.uint32) for _ in range(5)]
        if n.value:
            self._chk(self.L.irf_export_junctions(self.h, C.byref(n), *[x.ctypes.data for x in a]))
        return dict(zip(("chr", "left", "right", "cnt_neg", "cnt_pos"), a))

    def import_junctions(self, rows):
        a = [np.ascontiguousarray(rows[k], np.uint32) for k in ("chr", "left", "right", "cnt_neg", "cnt_pos")]
        self._chk(self.L.irf_import_junctions(self.h, len(a[0]), *[x.ctypes.data for x in a]))

    def close(self):
        if self.h:
            self.L.irf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather(engines, root=0):
    """irf_gather: in-process collection over finalized engines with disjoint `owned` masks into engines[root]."""
    L = lib()
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    rc = L.irf_gather(arr, len(engines), int(root))
    if rc != 0:
        raise IrfError(rc, L.irf_last_error(engines[root].h).decode())


def device_batch(tensors, n_frag, n_blk):
    """irf_batch over device-resident arrays (objects with .data_ptr(), e.g. torch tensors), keyed like Batch.FIELDS."""
    b = IrfBatch()
    b.n_frag, b.n_blk = int(n_frag), int(n_blk)
    for name, _ in Batch.FIELDS:
        setattr(b, name, tensors[name].data_ptr())
    b.mem_kind = IRF_MEM_DEVICE
    return b
