"""GPU diagnostic: run the HIP path on a set of small/medium cases, compare every stage with the oracle and
write a report to gpurun_out/diag.txt.  Development tool (not a test, not the bench)."""
import ctypes, hashlib, os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import sz_amd
from sz_amd.fields import s_field, l_field, m_field

out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
rep = open(os.path.join(out_dir, "diag.txt"), "w")
def P(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); rep.write(s + "\n"); rep.flush()

def cases():
    rng = np.random.default_rng(0)
    z = s_field(40, 40, 40); z[np.abs(z) < 0.7] = 0.0
    yield "S40", s_field(40, 40, 40), O.ABS, 1e-4, 0
    yield "C1", np.fromfile(os.path.join(ROOT, "tests/golden/testfloat_8_8_128.dat"), dtype=np.float32).reshape(128, 8, 8), O.ABS, 1e-4, 0
    yield "M64", m_field(64), O.ABS, 1e-4, 0
    yield "L", l_field(30, 50, 70), O.ABS, 1e-4, 0
    yield "Sodd", s_field(37, 45, 70), O.ABS, 1e-4, 0
    yield "rand-mean", rng.random((33, 20, 17), dtype=np.float32), O.ABS, 1e-2, 0
    yield "zeros-mean", z, O.ABS, 1e-3, 0
    yield "M48-f64", m_field(48, np.float64), O.ABS, 1e-5, 0
    yield "S-f64-rel", s_field(32, 64, 64, np.float64), O.REL, 0, 1e-3
    yield "S128", s_field(128, 128, 128), O.ABS, 1e-4, 0
    yield "M128", m_field(128), O.ABS, 1e-4, 0
    if os.environ.get("DIAG_BIG"):
        yield "M256", m_field(256), O.ABS, 1e-4, 0

def main():
    cfg = os.path.join(ROOT, "tests/golden/sz_speed.config")
    assert sz_amd.SZ_Init(cfg) == 0
    ctx = sz_amd.HipContext(0)
    nfail = 0
    for name, d, mode, ab, rel in cases():
        try:
            t0 = time.time()
            ref, st = O.compress(d, mode, ab, rel, want_stages=True)
            ref_dec = O.decompress(ref, d.shape, d.dtype)
            t_or = time.time() - t0
            t0 = time.time()
            got = sz_amd.SZ_compress_args(d, mode, ab, rel)
            t_gpu = time.time() - t0
            stt = sz_amd.SZ_hip_last_stats()
            same = got == ref
            P(f"[{name}] shape={d.shape} {d.dtype} intervals={st['intervals']} use_mean={st['use_mean']} reg={st['reg_count']}/{st['num_blocks']} unpred={st['total_unpred']}"
              f" | stream {'IDENTICAL' if same else 'DIFFERENT'} gpu={len(got)}B ref={len(ref)}B | gpu stats: intervals={stt.intervals} use_mean={stt.use_mean} reg={stt.n_reg_blocks} unpred={stt.n_unpred}"
              f" ms_total={stt.ms_total:.2f} ms_quant={stt.ms_quant:.3f} ms_pre={stt.ms_prequant:.2f} ms_ent={stt.ms_entropy:.2f} ms_host={stt.ms_host:.2f} (oracle {t_or:.2f}s, call {t_gpu:.2f}s)")
            if not same:
                nfail += 1
                # localise: rerun through the low-level entry and fetch intermediates
                nb, ne = st['num_blocks'], st['num_elements']
                T = d.dtype
                meta = ref[:4 + (28 if T == np.float32 else 36)]
                b2, n2, s2 = ctx.compress(d.ctypes.data, False, d.shape, T, st['eb'], meta)
                P(f"   low-level stream identical to API stream: {b2 == got}; first diff byte vs ref: {next((i for i,(x,y) in enumerate(zip(b2,ref)) if x!=y), None)}")
                lor = ctx.debug_fetch(1, nb, np.uint8)
                P(f"   indicator mismatches: {int((lor != st['indicator']).sum())} / {nb}")
                coef = ctx.debug_fetch(0, 4 * nb, T).reshape(4, nb)
                regs = np.where(st['indicator'] == 0)[0]
                lorb = np.where(st['indicator'] == 1)[0]
                P(f"   fitted coef mismatches on Lorenzo blocks: {int((coef[:, lorb].view(np.uint8) != st['reg_params'][:, lorb].view(np.uint8)).any(axis=0).sum()) if len(lorb) else 0}")
                if len(regs):
                    P(f"   decoded coef mismatches on regression blocks: {int((coef[:, regs] != st['coeff_dec']).sum())}")
                cb = ctx.debug_fetch(3, ne, np.uint16).astype(np.int32)
                bad = np.where(cb != st['codes'])[0]
                P(f"   block-order code mismatches: {len(bad)} / {ne}; first {bad[:8].tolist()} gpu {cb[bad[:8]].tolist()} ref {st['codes'][bad[:8]].tolist()}")
                hist = ctx.debug_fetch(4, st['intervals'], np.uint32)
                rh = np.bincount(st['codes'], minlength=st['intervals'])[:st['intervals']]
                P(f"   histogram mismatches: {int((hist != rh).sum())}")
                if st['total_unpred']:
                    un = ctx.debug_fetch(7, st['total_unpred'], T)
                    P(f"   unpredictable value mismatches: {int((un.view(np.uint8) != st['unpred'].view(np.uint8)).sum())}")
            # decompress the REFERENCE-format stream (oracle's) on the GPU and compare bit for bit
            dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
            stt = sz_amd.SZ_hip_last_stats()
            iview = np.uint32 if d.dtype == np.float32 else np.uint64
            nbad = int((dec.view(iview) != ref_dec.view(iview)).sum())
            maxerr = float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max())
            P(f"   decompress: bit mismatches vs oracle {nbad} / {d.size}; max|x-x'|={maxerr:.6e} (eb {st['eb']:.6e}) ms_total={stt.ms_total:.2f} ms_quant={stt.ms_quant:.3f} ms_ent={stt.ms_entropy:.2f}")
            if nbad:
                nfail += 1
        except Exception:
            nfail += 1
            P(f"[{name}] EXCEPTION\n" + traceback.format_exc())
    P("FAILURES:", nfail)
    return nfail

if __name__ == "__main__":
    sys.exit(1 if main() else 0)
