"""The allocation-free entry point (irf_gpu_bgzf_inflate_noalloc) against the entry point it is a variant of
(irf_gpu_bgzf_inflate), through the C ABI: bit-exact bytes, two calls in a row, the second with a smaller input (the
library's device scratch array keeps the first call's contents).  Collected last (file name): `pytest -x` stops at the
first failure."""
import ctypes as C
import zlib

import numpy as np
import pytest

from irfinder_amd import engine
from irfinder_amd.synth.bam import bgzf_compress

pytestmark = pytest.mark.gpu


def _payload(n, seed):
    rng = np.random.default_rng(seed)
    # compressible and not: text-like runs, random bytes, a long run of one byte (overlapping matches)
    parts = [rng.integers(65, 70, n // 2, dtype=np.uint8).tobytes(), rng.integers(0, 256, n // 4, dtype=np.uint8).tobytes(),
             b"\x07" * (n - n // 2 - n // 4)]
    return b"".join(parts)


def _device_bytes(nbytes):
    import torch
    return torch.full((nbytes,), 0xA5, dtype=torch.uint8, device="cuda:0")


def test_noalloc_inflate_equals_allocating_entry_point_twice_second_smaller():
    import torch
    raw_big, raw_small = _payload(3_000_000, 1), _payload(70_001, 2)
    for raw in (raw_big, raw_small):          # first call: 46+ members; second: 2 members, shorter compressed input
        z = bgzf_compress(raw)
        assert len(z) <= engine.NOALLOC_MAX_BYTES
        want = engine.gpu_bgzf_inflate(z)                      # the existing entry point (allocates per call)
        assert want.tobytes() == raw
        out = _device_bytes(len(raw) + 64)
        n = engine.gpu_bgzf_inflate_noalloc(z, out.data_ptr(), out.numel())
        torch.cuda.synchronize()
        assert n == len(raw)
        got = out[:n].cpu().numpy()
        assert np.array_equal(got, want), "noalloc output differs from irf_gpu_bgzf_inflate"
    assert len(bgzf_compress(raw_small)) < len(bgzf_compress(raw_big))


def test_noalloc_inflate_refuses_corruption_like_the_allocating_entry_point():
    raw = _payload(200_000, 3)
    z = bytearray(bgzf_compress(raw))
    z[len(z) // 2] ^= 0x55                      # inside some member's DEFLATE data
    out = _device_bytes(len(raw) + 64)
    with pytest.raises(engine.IrfError) as e1:
        engine.gpu_bgzf_inflate(bytes(z))
    with pytest.raises(engine.IrfError) as e2:
        engine.gpu_bgzf_inflate_noalloc(bytes(z), out.data_ptr(), out.numel())
    assert e1.value.code == e2.value.code == -2
    # and the next good call is not disturbed by what the failed one left in the scratch array
    z = bgzf_compress(raw)
    n = engine.gpu_bgzf_inflate_noalloc(z, out.data_ptr(), out.numel())
    assert out[:n].cpu().numpy().tobytes() == raw


def test_noalloc_inflate_limits():
    out = _device_bytes(1024)
    L = engine.lib()
    need = C.c_size_t()
    big = np.zeros(engine.NOALLOC_MAX_BYTES + 1, np.uint8)
    assert L.irf_gpu_bgzf_inflate_noalloc(0, big.ctypes.data, big.nbytes, C.c_void_p(out.data_ptr()), 1024, C.byref(need)) == -3
    # more members than the member table holds: empty members (the 28-byte EOF marker) are members too
    eof = bgzf_compress(b"")
    assert len(eof) == 28
    many = np.frombuffer(eof * (engine.NOALLOC_MAX_MEMBERS + 1), np.uint8)
    assert L.irf_gpu_bgzf_inflate_noalloc(0, many.ctypes.data, many.nbytes, C.c_void_p(out.data_ptr()), 1024, C.byref(need)) == -3
    # output buffer too small: IRF_E_RANGE and the size needed
    z = bgzf_compress(b"x" * 5000)
    zz = np.frombuffer(z, np.uint8)
    assert L.irf_gpu_bgzf_inflate_noalloc(0, zz.ctypes.data, zz.nbytes, C.c_void_p(out.data_ptr()), 1024, C.byref(need)) == -3
    assert need.value == 5000
