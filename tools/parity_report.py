#!/usr/bin/env python3
"""Print the full parity table of the HIP path against the golden vectors (no asserts).
Usage (GPU box): python tools/parity_report.py [name-substring ...]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as gu  # noqa: E402
from test_gpu_parity import _run_adjoint, _run_forward  # noqa: E402


def maxrel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b); s = np.abs(b).max()
    return float(d.max() / s) if s > 0 else float(d.max())


def main():
    sel = sys.argv[1:]
    for name in gu.names():
        if sel and not any(s in name for s in sel):
            continue
        if "jreg" in name or "prior" in name:
            continue
        g = gu.load(name)
        par, sta, out = _run_forward(g)
        q = [f"{gu.rel_l2(out.qsim[i], g.fwd['qsim'][i]):.1e}" for i in range(g.mesh.ng)]
        print(f"== {name}: fwd qsim rel-L2 per gauge {q} maxrel {maxrel(out.qsim, g.fwd['qsim']):.1e} "
              f"cost {out.cost:.8g} vs {g.fwd['cost']:.8g} ({abs(out.cost - g.fwd['cost']) / abs(g.fwd['cost']):.1e})")
        print("   fstates rel-L2:", {k: f"{gu.rel_l2(getattr(out.fstates, k), g.fwd['fstates'][k]):.1e}" for k in gu.STRUCT_STATES[g.structure]},
              " maxrel:", {k: f"{maxrel(getattr(out.fstates, k), g.fwd['fstates'][k]):.1e}" for k in gu.STRUCT_STATES[g.structure]})
        hl, hr = np.asarray(out.fstates.hlr, np.float64), np.asarray(g.fwd["fstates"]["hlr"], np.float64)
        d = np.abs(hl - hr)
        for idx in np.argsort(d.ravel())[::-1][:4]:
            r, c = np.unravel_index(idx, d.shape)
            print(f"     hlr worst cell ({r},{c}): hip {hl[r, c]:.9g} ref {hr[r, c]:.9g} rel {d[r, c] / max(abs(hr[r, c]), 1e-300):.2e} "
                  f"lr {g.params['lr'][r, c]:.6g} flwacc {g.mesh.flwacc[r, c]} a {np.exp(-g.dt / (g.params['lr'][r, c] * 60.0)):.3e}")
        par, sta, out, pb, sb = _run_adjoint(g)
        print(f"   adj cost {out.cost:.8g} vs {g.adj['cost']:.8g}; qsim {gu.rel_l2(out.qsim, g.adj['qsim']):.1e}")
        print("   params_b rel-L2:", {k: f"{gu.rel_l2(getattr(pb, k), g.adj['parameters_b'][k]):.1e}" for k in gu.STRUCT_PARAMS[g.structure]},
              " |ref|max:", {k: f"{np.abs(g.adj['parameters_b'][k]).max():.1e}" for k in gu.STRUCT_PARAMS[g.structure]})
        print("   states_b rel-L2:", {k: f"{gu.rel_l2(getattr(sb, k), g.adj['states_b'][k]):.1e}" for k in gu.STRUCT_STATES[g.structure]},
              " |ref|max:", {k: f"{np.abs(g.adj['states_b'][k]).max():.1e}" for k in gu.STRUCT_STATES[g.structure]})
        print("   ref self-noise: qsim %.1e cost %.1e" % (g.noise["qsim"].max(), g.noise["cost"]),
              "fstates", {k: f"{g.noise['fstates'][k]:.1e}" for k in gu.STRUCT_STATES[g.structure]},
              "pb", {k: f"{g.noise['parameters_b'][k]:.1e}" for k in gu.STRUCT_PARAMS[g.structure]},
              "sb", {k: f"{g.noise['states_b'][k]:.1e}" for k in gu.STRUCT_STATES[g.structure]})
        sys.stdout.flush()


if __name__ == "__main__":
    main()
