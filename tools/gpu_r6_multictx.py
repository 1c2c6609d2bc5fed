"""round 6: several LONE contexts of one process at work at the same time on arrays with regression blocks -- the case the coefficient hand-off to
a running sweep was not protected against ("a call that starts alone and is joined by another context's call half-way").  T host threads, a context each, N calls each of
the 512^3 M-field (the sweep is FED while the host's chains run whenever a call starts alone); every stream is compared on the device with the stream of a single call.
usage: gpu_r6_multictx.py [threads [calls_per_thread [edge]]]   -> one JSON line"""
import os, sys, json, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sz_amd import api
from sz_amd.fields import m_field

def run(threads, calls, edge=512):
    dev = torch.device("cuda:0")
    meta = api.make_meta(np.float32, api.ABS, 1e-4)
    d = torch.from_numpy(m_field(edge)).to(dev)
    ctx0 = api.HipContext(0)
    ptr, n, st = ctx0.compress(d.data_ptr(), True, (edge, edge, edge), np.float32, 1e-4, meta, out_on_device=True)
    import ctypes
    ref = torch.empty(n, dtype=torch.uint8, device=dev)
    ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(n), 3)
    ctx0.close()
    res = {"threads": threads, "calls_per_thread": calls, "edge": edge, "stream_bytes": int(n), "errors": 0, "mismatches": 0, "fed": 0, "unfed": 0, "slow_calls_over_3x_median": 0}
    lock = threading.Lock()
    times = []
    def worker(tid):
        ctx = api.HipContext(0)
        out = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
        cmp_s = torch.cuda.Stream(device=dev)
        err = mis = fed = unfed = 0
        mine = []
        for it in range(calls):
            try:
                t0 = time.perf_counter()
                p, m, s = ctx.compress(d.data_ptr(), True, (edge, edge, edge), np.float32, 1e-4, meta, out_on_device=True)
                mine.append(time.perf_counter() - t0)
            except Exception as e:      # noqa: BLE001
                err += 1
                continue
            fed += int(s.chain_overlapped) == 2; unfed += int(s.chain_overlapped) != 2
            if m != n:
                mis += 1
                continue
            ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(p), ctypes.c_size_t(n), 3)
            ctypes.CDLL("libamdhip64.so").hipStreamSynchronize(None)      # (a device-to-device hipMemcpy may return before the copy is done; cmp_s does not wait for the null stream)
            with torch.cuda.stream(cmp_s):
                same = bool(torch.equal(out[:n], ref))
            mis += not same
        ctx.close()
        with lock:
            res["errors"] += err; res["mismatches"] += mis; res["fed"] += fed; res["unfed"] += unfed; times.extend(mine)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    for t in th: t.start()
    for t in th: t.join()
    res["wall_s"] = round(time.perf_counter() - t0, 2)
    if times:
        med = float(np.median(times)); res["median_call_ms"] = round(med * 1e3, 3); res["max_call_ms"] = round(max(times) * 1e3, 3)
        res["slow_calls_over_3x_median"] = int(sum(t > 3 * med for t in times))
    return res

if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    print(json.dumps(run(T, N, E)))
