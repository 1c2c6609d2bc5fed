"""Round 5: the beam sweep (szh_beam.h) on the GPU: parity against the oracle on small arrays, against k_ribbon's stream at full size, timings of both.
Usage: python tools/gpu_r5_beam.py [edge]   (run through gpurun; prints one JSON line per check)"""
import ctypes, hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sz_amd
from sz_amd import api
from sz_amd.fields import s_field, m_field
import oracle_lib
HIP = ctypes.CDLL('libamdhip64.so')

def small_parity():
    rng = np.random.default_rng(7)
    cases = [("S-12x8x32", s_field(12, 8, 32), 1e-4), ("S-10x40x36", s_field(10, 40, 36), 1e-4), ("S-64^3", s_field(64, 64, 64), 1e-4),
             ("N-60x36x40", (s_field(60, 36, 40) + (rng.random((60, 36, 40)) - 0.5) * 3e-4).astype(np.float32), 1e-4),
             ("S-100x70x68", s_field(100, 70, 68), 1e-4), ("M40", m_field(40), 1e-4), ("M64", m_field(64), 1e-4), ("M36-f64", m_field(36, np.float64), 1e-3), ("S64-41x70x36", s_field(41, 70, 36, np.float64), 1e-3), ("S-128^3", s_field(128, 128, 128), 1e-4)]
    assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
    bad = 0
    for name, d, eb in cases:
        ref, _ = oracle_lib.compress(d, oracle_lib.ABS, eb)
        for beam in ("2", "0"):
            os.environ["SZ_HIP_BEAM"] = beam
            t0 = time.time()
            try:
                got = sz_amd.SZ_compress_args(d, sz_amd.ABS, eb)
                st = sz_amd.SZ_hip_last_stats()
                dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
                okc = got == ref
                okd = np.array_equal(dec.view(np.uint8), oracle_lib.decompress(ref, d.shape, d.dtype).view(np.uint8))
                print(json.dumps({"check": "small", "case": name, "beam": beam, "compress_ok": okc, "decompress_ok": bool(okd), "kernel": int(st.quant_kernel), "s": round(time.time() - t0, 2)}), flush=True)
                bad += (not okc) + (not okd)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"check": "small", "case": name, "beam": beam, "error": str(e)[:200]}), flush=True); bad += 1
    sz_amd.SZ_Finalize()
    os.environ["SZ_HIP_BEAM"] = "2"
    return bad

def full(edge, field="s"):
    dev = torch.device("cuda:0")
    d = torch.from_numpy(s_field(edge, edge, edge) if field == "s" else m_field(edge)).to(dev)
    out = torch.empty_like(d)
    meta = api.make_meta(np.float32, api.ABS, 1e-4)
    res = {}
    for beam in ("0", "2"):
        os.environ["SZ_HIP_BEAM"] = beam
        ctx = api.HipContext(0)
        tq, tt, td, tdq = [], [], [], []
        md5 = None; size = 0
        try:
            for it in range(8):
                t0 = time.time()
                ptr, n, st = ctx.compress(d.data_ptr(), True, (edge, edge, edge), np.float32, 1e-4, meta, out_on_device=True)
                torch.cuda.synchronize()
                tt.append((time.time() - t0) * 1e3); tq.append(st.ms_quant)
                if it == 0:
                    tmp = torch.empty(n, dtype=torch.uint8, device=dev)
                    rc = HIP.hipMemcpy(ctypes.c_void_p(tmp.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(n), 3)
                    assert rc == 0, rc
                    host = tmp.cpu().numpy()
                    md5 = hashlib.md5(host.tobytes()).hexdigest(); size = n
                    stream_dev = tmp
            for it in range(6):
                t0 = time.time()
                st = ctx.decompress(stream_dev.data_ptr(), True, size, 4 + 28 + 8, (edge, edge, edge), np.float32, out.data_ptr(), True)
                torch.cuda.synchronize()
                td.append((time.time() - t0) * 1e3); tdq.append(st.ms_quant)
            err = float((out.double() - d.double()).abs().max())
            res[beam] = {"field": field, "md5": md5, "size": size, "host_ms": round(st.ms_host, 3), "reg_blocks": int(st.n_reg_blocks), "quant_ms": sorted(tq)[len(tq) // 2], "quant_ms_min": min(tq), "call_ms": sorted(tt)[len(tt) // 2], "dec_call_ms": sorted(td)[len(td) // 2],
                         "dec_quant_ms": sorted(tdq)[len(tdq) // 2], "max_err": err, "out_md5": hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()}
        except Exception as e:  # noqa: BLE001
            res[beam] = {"error": str(e)[:300]}
        ctx.close()
        print(json.dumps({"check": "full", "edge": edge, "beam": beam, **res[beam]}), flush=True)
    same = res["0"].get("md5") is not None and res["0"].get("md5") == res["2"].get("md5") and res["0"].get("out_md5") == res["2"].get("out_md5")
    print(json.dumps({"check": "full-compare", "edge": edge, "streams_and_outputs_identical": bool(same)}), flush=True)

if __name__ == "__main__":
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    bad = small_parity()
    print(json.dumps({"check": "small-summary", "bad": bad}), flush=True)
    if bad == 0 or os.environ.get("R5_FORCE_FULL"):
        for f in os.environ.get("R5_FIELDS", "s,m").split(","):
            full(edge, f)
