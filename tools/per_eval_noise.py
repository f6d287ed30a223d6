"""Per-evaluation rounding noise of the device's workgroup program (reduced knot system solved by the Thomas recurrences, the device's
summation orders; run here through the host emulator of tests/emu) against the oracle's banded LU in reference order -- next to the
oracle's OWN noise when it is merely recompiled with FMA contraction.  VERDICT r1 item 1(c): "quantify what the knot operator costs".
usage: python tools/per_eval_noise.py   (CPU only)"""
import sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle_py as O
from uneven_planner_amd import scenes
import emu_bridge as E
cells = scenes.analytic_cells()
probs = scenes.random_problems(24, seed0=1000)
def evals(mod):
    g = mod.OracleGrid(); g.set_cells(cells)
    out = []
    for p in probs:
        a = mod.OracleALM(g); x0 = a.setup(p); a.init_scaling(x0)
        rng = np.random.default_rng(1); x = x0 + 0.01 * rng.normal(size=x0.size)
        f, gr, _ = a.eval(x); out.append((f, gr, a.coeffs()[0]))
    return out
plain = evals(O)
so = "/tmp/liboracle_fma_pe.so"
subprocess.check_call(["g++", "-O3", "-march=native", "-ffp-contract=fast", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
saved = O._LIB; O._LIB = None; real = O.os.path.join
O.os.path.join = lambda *a, _r=real: so if a[-1] == "liboracle.so" else _r(*a)
fma = evals(O)
O.os.path.join = real; O._LIB = saved
# emulator (the device's workgroup program on the host: Thomas knot solve, device summation orders)
em = E.Emu(cells, O.map_params_vec(), O.params_vec()); E.lib().emu_set_lanes(128)
g = O.OracleGrid(); g.set_cells(cells)
emu = []
for p in probs:
    a = O.OracleALM(g); x0 = a.setup(p); a.init_scaling(x0); st = a.get_state()
    rng = np.random.default_rng(1); x = x0 + 0.01 * rng.normal(size=x0.size)
    r = em.run(0, p, x, lam=np.zeros(a.S), mu=np.zeros(6 * a.S), scale_cx=st["scale_cx"], rho=1.0, scale_fx=st["scale_fx"])
    emu.append((r["f"], r["g"], r["c_xy"]))
def rel(a, b): return np.abs(a - b).max() / np.abs(a).max()
for name, other in (("oracle(FMA) vs oracle", fma), ("device program (emulated) vs oracle", emu)):
    df = [abs(a[0] - b[0]) / abs(a[0]) for a, b in zip(plain, other)]
    dg = [rel(a[1], b[1]) for a, b in zip(plain, other)]
    dc = [rel(np.asarray(a[2]).ravel(), np.asarray(b[2]).ravel()) for a, b in zip(plain, other)]
    print("%-40s f: median %.1e max %.1e   grad: median %.1e max %.1e   coeffs: median %.1e max %.1e" % (name, np.median(df), max(df), np.median(dg), max(dg), np.median(dc), max(dc)))
