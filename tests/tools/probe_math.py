"""GPU box: compare device arithmetic primitives with the oracle / numpy bit for bit."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from oracle import orc
lib = rrt.load(); O = orc.load()
def dev(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32); out = np.zeros_like(a)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    rc = rrt.load_diag().mipt_debug_eval(op, a.ctypes.data, None if bb is None else bb.ctypes.data, a.size, out.ctypes.data)
    assert rc == 0, lib.mipt_last_error()
    return out
rng = np.random.default_rng(1)
N = 2_000_000
def report(name, got, want):
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    # NaN payloads may differ: treat both-NaN as equal
    bad = [i for i in bad if not (np.isnan(got[i]) and np.isnan(want[i]))]
    print(f"{name}: {len(bad)} mismatches of {got.size}")
    return bad
# division / sqrt / mul / add over wild ranges
a = (rng.standard_normal(N) * np.exp(rng.uniform(-40, 40, N))).astype(np.float32)
b = (rng.standard_normal(N) * np.exp(rng.uniform(-40, 40, N))).astype(np.float32)
a[:8] = [0, -0.0, np.inf, -np.inf, np.nan, 1, 1e-45, 3e38]; b[:8] = [0, 1, np.inf, 0, 1, 0, 3, 1e-45]
with np.errstate(all="ignore"):
    bad = report("div", dev(3, a, b), (a / b).astype(np.float32))
    for i in bad[:5]: print("   ", a[i], b[i], dev(3, a[i:i+1], b[i:i+1]), a[i] / b[i])
    report("sqrt", dev(4, np.abs(a)), np.sqrt(np.abs(a)))
    report("mul", dev(5, a, b), a * b)
    report("add", dev(6, a, b), a + b)
# shim functions vs oracle's C
x = (rng.random(N) * 6.2832).astype(np.float32)
want = np.array([O.orc_glibc_cosf(float(v)) for v in x[:200000]], dtype=np.float32)
bad = report("cos", dev(0, x[:200000]), want)
for i in bad[:5]: print("   ", x[i], dev(0, x[i:i+1]), want[i])
r = rng.random(N).astype(np.float32)
want = np.array([O.orc_glibc_log10f(float(v)) for v in r[:200000]], dtype=np.float32)
bad = report("log10", dev(1, r[:200000]), want)
for i in bad[:5]: print("   ", r[i], dev(1, r[i:i+1]), want[i])
seeds = rng.integers(1, 2**32, 200000, dtype=np.uint32)
s = seeds.view(np.float32)
want = np.zeros(len(seeds), dtype=np.float32)
for i, sd in enumerate(seeds):
    st = C.c_uint32(int(sd)); want[i] = O.orc_rand_f32_nd(C.byref(st), 0)
bad = report("rand_nd", dev(10, s), want)
for i in bad[:5]: print("   ", seeds[i], dev(10, s[i:i+1]), want[i])
for comp in range(3):
    want = np.zeros(len(seeds), dtype=np.float32)
    for i, sd in enumerate(seeds[:50000]):
        st = C.c_uint32(int(sd)); o3 = (C.c_float * 3)(); O.orc_rand_in_unit_sphere(C.byref(st), 0, C.byref(o3)); want[i] = o3[comp]
    report(f"sphere[{comp}]", dev(11, s[:50000], np.full(50000, comp, np.float32)), want[:50000])
