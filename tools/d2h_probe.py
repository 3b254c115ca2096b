"""Diagnostic: device -> page-locked host copy rates at the IR table's sizes (torch copy_, hipMemcpyAsync kinds)."""
import ctypes, time, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
for mb in (4.8, 12.8, 17.6, 64.0):
    n = int(mb * 1e6) // 8 * 8
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    s = torch.cuda.Stream()
    res = {}
    with torch.cuda.stream(s):
        for name in ("torch", "kind_d2h", "kind_default"):
            ts = []
            for it in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if name == "torch":
                    h.copy_(d, non_blocking=True)
                else:
                    rc = hip.hipMemcpyAsync(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), 2 if name == "kind_d2h" else 4, ctypes.c_void_p(s.cuda_stream))
                    assert rc == 0
                s.synchronize()
                ts.append(time.perf_counter() - t0)
            res[name] = min(ts)
    print("%.1f MB: " % mb + ", ".join("%s %.3f ms (%.1f GB/s)" % (k, v * 1e3, n / v / 1e9) for k, v in res.items()), flush=True)
