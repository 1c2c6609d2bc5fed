"""Round 5: time line of one M-field compression (SZ_HIP_TIMING=1 prints the host's view, the stats the device's)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sz_amd import api
from sz_amd.fields import m_field, s_field
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
meta = api.make_meta(np.float32, api.ABS, 1e-4)
for field in (sys.argv[2] if len(sys.argv) > 2 else "m,s").split(","):
    d = torch.from_numpy(m_field(edge) if field == "m" else s_field(edge, edge, edge)).to(dev)
    ctx = api.HipContext(0)
    for it in range(6):
        if it == 4: os.environ["SZ_HIP_TIMING"] = "1"
        ptr, n, st = ctx.compress(d.data_ptr(), True, (edge, edge, edge), np.float32, 1e-4, meta, out_on_device=True)
        os.environ.pop("SZ_HIP_TIMING", None)
        if it >= 3:
            print(json.dumps({"field": field, "it": it, "total_ms": round(st.ms_total, 3), "prequant": round(st.ms_prequant, 3), "quant": round(st.ms_quant, 3), "entropy": round(st.ms_entropy, 3), "host": round(st.ms_host, 3), "kernel": int(st.quant_kernel), "size": n}), flush=True)
    ctx.close()
    # the way back: the stream stays on the device (a copy: the context's output buffer is reused), the result goes to a device array
    if os.environ.get("R5_DEC", "1") != "0":
        import ctypes
        ctx = api.HipContext(0)
        ptr, n, st = ctx.compress(d.data_ptr(), True, (edge, edge, edge), np.float32, 1e-4, meta, out_on_device=True)
        stream = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        out = torch.empty_like(d)
        torch.cuda.synchronize()
        ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(stream.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(n), 3)
        ctypes.CDLL("libamdhip64.so").hipDeviceSynchronize()     # (a device-to-device hipMemcpy may return before the copy is done, and the context's streams do not wait for the null stream: include/szhip.h)
        for it in range(5):
            if it == 3: os.environ["SZ_HIP_TIMING"] = "1"
            else: os.environ.pop("SZ_HIP_TIMING", None)
            try: st = ctx.decompress(stream.data_ptr(), True, n, 4 + 28 + 8, (edge, edge, edge), np.float32, out.data_ptr(), True)
            except Exception as ex:
                print("DEC FAILED field", field, "dec_it", it, str(ex)[-90:], flush=True); continue
            if it >= 3:
                print(json.dumps({"field": field, "dec_it": it, "total_ms": round(st.ms_total, 3), "entropy": round(st.ms_entropy, 3), "quant": round(st.ms_quant, 3), "host": round(st.ms_host, 3), "kernel": int(st.quant_kernel), "max_err": float((out - d).abs().max())}), flush=True)
        ctx.close()
