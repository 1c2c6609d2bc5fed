"""Round 5: sweep time of the beam kernel by array shape (what the step costs, what the hops along j and k cost).  Run through gpurun."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sz_amd import api
from sz_amd.fields import s_field
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(512, 32, 32), (512, 64, 32), (512, 128, 32), (512, 512, 32), (512, 32, 128), (512, 32, 512), (512, 128, 128), (128, 512, 512), (256, 512, 512), (512, 512, 512)]
dev = torch.device("cuda:0")
meta = api.make_meta(np.float32, api.ABS, 1e-4)
for beam in os.environ.get("R5_BEAMS", "2").split(","):
    os.environ["SZ_HIP_BEAM"] = beam
    ctx = api.HipContext(0)
    for sh in shapes:
        d = torch.from_numpy(s_field(*sh)).to(dev)
        q = []
        for it in range(6):
            ptr, n, st = ctx.compress(d.data_ptr(), True, sh, np.float32, 1e-4, meta, out_on_device=True)
            q.append(st.ms_quant)
        q = sorted(q[1:])
        steps = 5 * (sh[0] + 8)
        print(json.dumps({"beam": beam, "shape": "x".join(map(str, sh)), "quant_ms_med": round(q[len(q) // 2], 4), "min": round(q[0], 4), "ns_per_wave_step_if_no_lag": round(q[0] * 1e6 / steps, 1), "kernel": int(st.quant_kernel)}), flush=True)
    ctx.close()
