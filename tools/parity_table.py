#!/usr/bin/env python3
"""Parity table of the HIP path against the golden vectors of the reference Fortran, every output field of every fixture:
rel-L2 error, number of elements whose BITS differ, the strict 1e-6 bar of BASELINE.json and the noise-aware bar the tests use
(tests/golden_util.tol).  Runs in whichever build smash_amd loads: default, or the exact-libm build with SMASHX_EXACT_LIBM=1.

    python tools/parity_table.py [--out FILE.md] [--assert-exact] [name-substring ...]

--assert-exact (used by tests/test_gpu_exact.py under SMASHX_EXACT_LIBM=1): forward outputs must be bit-identical to the
reference, every gradient field within the STRICT 1e-6."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as gu  # noqa: E402
from smash_amd import _lib  # noqa: E402
from test_gpu_parity import _run_adjoint, _run_forward  # noqa: E402


def nbits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--assert-exact", action="store_true")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    rows, fails = [], []
    worst = {"fwd": 0.0, "grad": 0.0}
    for name in gu.names():
        if a.names and not any(s in name for s in a.names):
            continue
        g = gu.load(name)
        par, sta, out = _run_forward(g)

        def add(kind, field, got, ref, bar, is_fwd):
            e = gu.rel_l2(got, ref)
            nb = nbits(got, ref)
            rows.append((name, kind, field, e, nb, np.asarray(ref).size, e <= 1e-6, bar, e <= bar))
            worst["fwd" if is_fwd else "grad"] = max(worst["fwd" if is_fwd else "grad"], e)
            if a.assert_exact and ((is_fwd and nb) or (not is_fwd and not e <= 1e-6)):
                fails.append((name, kind, field, e, nb))
        for i in range(g.mesh.ng):
            add("forward", f"qsim[{i}]", out.qsim[i], g.fwd["qsim"][i], gu.tol(g.noise["qsim"][i]), True)
        add("forward", "cost", np.float32(out.cost), np.float32(g.fwd["cost"]), gu.tol_cost(g.noise["cost"], g.fwd["cost"]) / max(abs(g.fwd["cost"]), 1e-30), True)
        for k in gu.STRUCT_STATES[g.structure]:
            add("forward", "fstates." + k, getattr(out.fstates, k), g.fwd["fstates"][k], gu.tol_fstate(k, g.noise["fstates"][k]), True)
        par, sta, out, pb, sb = _run_adjoint(g)
        add("adjoint", "qsim", out.qsim, g.adj["qsim"], gu.tol(g.noise["qsim"].max()), True)
        for k in gu.STRUCT_PARAMS[g.structure]:
            add("adjoint", k + "_b", getattr(pb, k), g.adj["parameters_b"][k], gu.tol(g.noise["parameters_b"][k]), False)
        for k in gu.STRUCT_STATES[g.structure]:
            add("adjoint", k + "_b", getattr(sb, k), g.adj["states_b"][k], gu.tol(g.noise["states_b"][k]), False)
    mode = "exact-libm build (SMASHX_EXACT_LIBM=1: glibc 2.35 expf/logf/powf/tanhf restated, IEEE division)" if _lib.EXACT else "default build"
    lines = [f"# Parity of the HIP path against the reference's golden vectors -- {mode}", "",
             f"library: `{os.path.relpath(_lib.LIB_PATH, ROOT)}`; error = rel-L2 over the field; `bits` = elements whose bit pattern differs / elements.",
             "", "| fixture | sweep | field | rel-L2 | bits differ | strict 1e-6 | noise-aware bar | within bar |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]:.2e} | {r[4]} / {r[5]} | {'yes' if r[6] else 'NO'} | {r[7]:.1e} | {'yes' if r[8] else 'NO'} |")
    nfw = [r for r in rows if r[1] == "forward" or r[2] == "qsim"]
    ngr = [r for r in rows if r not in nfw]
    summ = (f"forward outputs: {sum(r[4] == 0 for r in nfw)} of {len(nfw)} bit-identical, worst rel-L2 {worst['fwd']:.2e}, {sum(r[6] for r in nfw)} within strict 1e-6; "
            f"gradient fields: {sum(r[4] == 0 for r in ngr)} of {len(ngr)} bit-identical, worst rel-L2 {worst['grad']:.2e}, {sum(r[6] for r in ngr)} of {len(ngr)} within strict 1e-6, "
            f"{sum(r[8] for r in ngr)} within the noise-aware bar")
    lines += ["", "**Summary.** " + summ, ""]
    txt = "\n".join(lines)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt)
    print(summ)
    if fails:
        print("FAILED (exact mode):", fails[:20])
        sys.exit(1)


if __name__ == "__main__":
    main()
