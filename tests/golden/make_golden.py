#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference solver (oracle/_ref/libsmash_ref.so,
built by oracle/ref/build_ref.sh with flang -O2 -ffp-contract=off from /root/reference).

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
Each fixture is data only: the inputs handed to the reference's mw_forward::forward / forward_b
(smash/solver/forward/mw_forward.f90:18-68) and the outputs it returned.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)

from oracle import refbind  # noqa: E402
from oracle.refbind import GLB_P, GLB_S, GUB_P, GUB_S  # noqa: E402
from smash_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DT = 3600.0


def norm(d, names, lb, ub):
    return {k: np.asfortranarray(((d[k] - lb[i]) / (ub[i] - lb[i])).astype(np.float32)) for i, k in enumerate(names)}


CASES = [
    # name, structure, n, nt, ng, mask, gaps ppm, options
    dict(name="gr_a_12x12x48_nse", structure="gr-a", n=12, nt=48, ng=2, mask=False, gaps=0, opts={}),
    dict(name="gr_b_16x16x96_nse_gaps", structure="gr-b", n=16, nt=96, ng=3, mask=False, gaps=20000, opts={}),
    dict(name="gr_c_16x16x96_kge_se_log_mask", structure="gr-c", n=16, nt=96, ng=3, mask=True, gaps=5000,
         opts=dict(jobs_fun=("kge", "se", "logarithmic"), wjobs_fun=(1.0, 0.5, 0.25))),
    dict(name="gr_d_12x12x48_rmse_kge2_start", structure="gr-d", n=12, nt=48, ng=2, mask=False, gaps=5000,
         opts=dict(jobs_fun=("rmse", "kge2"), wjobs_fun=(1.0, 0.5), optimize_start_step=7)),
    dict(name="gr_b_24x24x120_norm_jreg", structure="gr-b", n=24, nt=120, ng=3, mask=True, gaps=5000, normalized=True,
         opts=dict(jobs_fun=("nse", "kge"), wjobs_fun=(0.7, 0.3), jreg_fun=("prior", "smoothing", "hard_smoothing"),
                   wjreg_fun=(1.0, 0.5, 0.1), wjreg=1e-3, denormalize_forward=True, optimize_start_step=13,
                   wgauge=[0.5, 0.0, 0.5])),
    dict(name="gr_a_24x24x120_norm_prior", structure="gr-a", n=24, nt=120, ng=3, mask=False, gaps=1000, normalized=True,
         opts=dict(jreg_fun=("prior",), wjreg_fun=(1.0,), wjreg=1e-2, denormalize_forward=True)),
    # negative gauge weights: the median over those gauges replaces the weighted sum (mwd_cost.f90:139-154)
    dict(name="gr_a_16x16x96_median3", structure="gr-a", n=16, nt=96, ng=4, mask=False, gaps=1000,
         opts=dict(wgauge=[-1.0, 0.3, -1.0, -1.0])),
    dict(name="gr_b_16x16x96_median2", structure="gr-b", n=16, nt=96, ng=3, mask=True, gaps=1000,
         opts=dict(jobs_fun=("nse", "kge"), wjobs_fun=(0.6, 0.4), wgauge=[-0.5, -0.5, 0.0])),
    # all eight D8 codes, interior outlet, ragged mask (synth.make_mesh_d8): what real catchments look like
    dict(name="gr_b_20x20x96_d8", structure="gr-b", n=20, nt=96, ng=3, mask=False, gaps=1000, d8=True, opts={}),
    dict(name="gr_c_32x32x240_d8_ragged", structure="gr-c", n=32, nt=240, ng=4, mask=True, gaps=1000, d8=True, radius=0.42,
         opts=dict(jobs_fun=("kge", "nse"), wjobs_fun=(0.5, 0.5))),
    # vic-a structure (md_vic_operator.f90, vic_a_forward md_forward_structure.f90:762-931)
    dict(name="vic_a_16x16x96_nse_gaps", structure="vic-a", n=16, nt=96, ng=3, mask=False, gaps=20000, opts={}),
    dict(name="vic_a_24x24x240_d8_kge", structure="vic-a", n=24, nt=240, ng=3, mask=True, gaps=1000, d8=True, radius=0.45,
         opts=dict(jobs_fun=("kge", "nse"), wjobs_fun=(0.5, 0.5))),
    dict(name="gr_a_12x12x48_nse_cold", structure="gr-a", n=12, nt=48, ng=2, mask=False, gaps=0, warm=False, opts={}),
    # larger cases: forcing is regenerated from smash_amd.synth (sha256 pinned in the fixture)
    dict(name="gr_b_64x64x720_nse", structure="gr-b", n=64, nt=720, ng=4, mask=False, gaps=1000, big=True, opts={}),
    dict(name="gr_a_64x64x720_nse_cold", structure="gr-a", n=64, nt=720, ng=4, mask=False, gaps=1000, big=True, warm=False, opts={}),
    dict(name="gr_c_48x48x480_nse", structure="gr-c", n=48, nt=480, ng=4, mask=True, gaps=1000, big=True, opts={}),
    dict(name="gr_d_48x48x480_nse", structure="gr-d", n=48, nt=480, ng=4, mask=False, gaps=1000, big=True, opts={}),
    # the reference's own real-data case (BASELINE.json configs[0]): Cance, 28 x 28 cells (383 active), 3 gauges, 1440 hourly
    # steps, read from the dataset files by cance_io.py; uniform parameters = the SBS optimum printed in model.py:784
    dict(name="gr_a_cance_28x28x1440", structure="gr-a", dataset="cance", n=28, nt=1440, ng=3, mask=True, gaps=0,
         opts=dict(wgauge=[1.0, 0.0, 0.0])),
]
CANCE_SBS = dict(cp=76.57858, cft=263.64627, exc=-1.455813, lr=30.859276)


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_case(c):
    n, nt = c["n"], c["nt"]
    if c.get("dataset") == "cance":
        import cance_io
        mesh, prcp, pet, qobs, P, S = cance_io.load()
        for k, v in CANCE_SBS.items():
            P[k][:] = v
        return mesh, prcp, pet, qobs, P, S, dict(c["opts"])
    if c.get("d8"):
        mesh = synth.make_mesh_d8(n, n, ng=c["ng"], radius=c.get("radius", 0.0))
    else:
        mesh = synth.make_mesh(n, n, ng=c["ng"], mask_corner=c["mask"])
    prcp, pet = synth.dense_forcing(mesh, nt, gap_per_million=c["gaps"])
    P = synth.make_parameters(n, n)
    S = synth.make_states(n, n, warm=c.get("warm", True))
    Pq = synth.make_parameters(n, n, perturb=0.1)
    qobs = refbind.run(c["structure"], mesh, DT, prcp, pet, np.zeros((c["ng"], nt), np.float32), Pq, S)["qsim"].copy()
    if c["gaps"]:
        qobs[-1, nt // 5: nt // 3] = -99.0          # missing observations (mwd_cost.f90:378)
    opts = dict(c["opts"])
    if c.get("normalized"):
        op = np.zeros(16, np.int32)
        op[[1, 3, 6, 15]] = 1
        os_ = np.zeros(8, np.int32)
        os_[[1, 2]] = 1
        opts.update(optim_parameters=op, optim_states=os_,
                    params_bgd=norm(Pq, synth.PARAM_NAMES, GLB_P, GUB_P),
                    states_bgd=norm(S, synth.STATE_NAMES, GLB_S, GUB_S))
        P = norm(P, synth.PARAM_NAMES, GLB_P, GUB_P)
        S = norm(S, synth.STATE_NAMES, GLB_S, GUB_S)
    return mesh, prcp, pet, qobs, P, S, opts


def main():
    if not refbind.available():
        raise SystemExit("oracle/_ref/libsmash_ref.so missing: run oracle/ref/build_ref.sh first")
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for c in CASES:
        if only and c["name"] not in only:
            continue
        mesh, prcp, pet, qobs, P, S, opts = build_case(c)
        f = refbind.run(c["structure"], mesh, DT, prcp, pet, qobs, P, S, **opts)
        b = refbind.run(c["structure"], mesh, DT, prcp, pet, qobs, P, S, adjoint=True, **opts)
        # the reference's own flag-to-flag noise: same sources built the way its makefile does
        # (-O3 + FMA contraction, makefile:6) against the parity build (-O2 -ffp-contract=off)
        f3 = refbind.run(c["structure"], mesh, DT, prcp, pet, qobs, P, S, fast=True, **opts)
        b3 = refbind.run(c["structure"], mesh, DT, prcp, pet, qobs, P, S, adjoint=True, fast=True, **opts)
        d = dict(structure=c["structure"], dt=DT, dx=mesh.dx, nrow=mesh.nrow, ncol=mesh.ncol, nt=c["nt"],
                 gaps=c["gaps"], mask=int(c["mask"]), big=int(bool(c.get("big"))),
                 flwdir=mesh.flwdir, flwacc=mesh.flwacc, path=mesh.path, active_cell=mesh.active_cell,
                 gauge_pos=mesh.gauge_pos, area=mesh.area, qobs=qobs,
                 prcp_sha=sha(prcp), pet_sha=sha(pet))
        if not c.get("big"):
            d.update(prcp=prcp, pet=pet)
        for k in synth.PARAM_NAMES:
            d["p_" + k] = P[k]
        for k in synth.STATE_NAMES:
            d["s_" + k] = S[k]
        for k, v in opts.items():
            if k in ("params_bgd", "states_bgd"):
                for kk, vv in v.items():
                    d[("pbgd_" if k == "params_bgd" else "sbgd_") + kk] = vv
            elif k in ("jobs_fun", "jreg_fun"):
                d["opt_" + k] = np.array(list(v))
            else:
                d["opt_" + k] = np.asarray(v)
        # expected outputs
        d.update(fwd_qsim=f["qsim"], fwd_cost=np.float32(f["cost"]), fwd_cost_jobs=np.float32(f["cost_jobs"]),
                 fwd_cost_jreg=np.float32(f["cost_jreg"]), adj_qsim=b["qsim"], adj_cost=np.float32(b["cost"]))
        d["noise_qsim"] = np.array([rel_l2(f3["qsim"][i], f["qsim"][i]) for i in range(c["ng"])])
        d["noise_cost"] = np.float64(abs(f3["cost"] - f["cost"]) / max(abs(f["cost"]), 1e-300))
        for k in synth.STATE_NAMES:
            d["noise_fstates_" + k] = np.float64(rel_l2(f3["fstates"][k], f["fstates"][k]))
            d["noise_states_b_" + k] = np.float64(rel_l2(b3["states_b"][k], b["states_b"][k]))
        for k in synth.PARAM_NAMES:
            d["noise_parameters_b_" + k] = np.float64(rel_l2(b3["parameters_b"][k], b["parameters_b"][k]))
        for k in synth.STATE_NAMES:
            d["fwd_fstates_" + k] = f["fstates"][k]
            d["adj_states_b_" + k] = b["states_b"][k]
            d["fwd_states_out_" + k] = f["states"][k]
        for k in synth.PARAM_NAMES:
            d["adj_parameters_b_" + k] = b["parameters_b"][k]
            d["fwd_parameters_out_" + k] = f["parameters"][k]
        path = os.path.join(OUT, c["name"] + ".npz")
        np.savez_compressed(path, **d)
        print(f"{c['name']}: cost={f['cost']:.8g} |cp_b|={np.abs(b['parameters_b']['cp']).max():.4g} "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


def tangent_direction(g):
    """The direction the tangent fixtures use: 1 % of every field, signed like the reference gradient, so that
    cost_d = sum |grad . d| is well conditioned (a random direction makes cost_d a cancelling sum)."""
    pd = {k: np.asfortranarray((0.01 * np.maximum(np.abs(v), 1e-3) * np.sign(g.adj["parameters_b"][k])).astype(np.float32))
          for k, v in g.params.items()}
    sd = {k: np.asfortranarray((0.01 * np.maximum(np.abs(v), 1e-3) * np.sign(g.adj["states_b"][k])).astype(np.float32))
          for k, v in g.states.items()}
    return pd, sd


TANGENT_CASES = ["gr_a_12x12x48_nse", "gr_b_16x16x96_nse_gaps", "gr_c_16x16x96_kge_se_log_mask", "gr_d_12x12x48_rmse_kge2_start",
                 "gr_b_24x24x120_norm_jreg", "gr_b_16x16x96_median2", "gr_c_32x32x240_d8_ragged",
                 "vic_a_16x16x96_nse_gaps", "vic_a_24x24x240_d8_kge", "gr_a_cance_28x28x1440"]


def main_tangent():
    """forward_d of the reference (mw_forward.f90:70-97) along tangent_direction(): cost_d, qsim_d and their
    flag-to-flag noise (-O3 + FMA build against the parity build)."""
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_util as gu
    os.makedirs(os.path.join(OUT, "tangent"), exist_ok=True)
    for name in TANGENT_CASES:
        g = gu.load(name)
        pd, sd = tangent_direction(g)
        r = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, params_d=pd, states_d=sd, **g.opts)
        r3 = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, params_d=pd, states_d=sd, fast=True, **g.opts)
        np.savez_compressed(os.path.join(OUT, "tangent", name + ".npz"), cost_d=np.float32(r["cost_d"]), qsim_d=r["qsim_d"],
                            noise_cost_d=np.float64(abs(r3["cost_d"] - r["cost_d"]) / abs(r["cost_d"])),
                            noise_qsim_d=np.array([rel_l2(r3["qsim_d"][i], r["qsim_d"][i]) for i in range(g.mesh.ng)]))
        print(f"tangent {name}: cost_d={r['cost_d']:.8g} noise={abs(r3['cost_d'] - r['cost_d']) / abs(r['cost_d']):.2g}")


def hyper_inputs(g, mapping):
    """Two smooth descriptors and hyper-parameter vectors whose mapped fields sit near the case's own parameters:
    intercept = logit of the normalised field mean, small descriptor coefficients (and exponents for the polynomial map)."""
    r = np.arange(g.mesh.nrow)[:, None] / g.mesh.nrow
    c = np.arange(g.mesh.ncol)[None, :] / g.mesh.ncol
    desc = np.asfortranarray(np.stack([0.2 + r + 0 * c, 0.5 + 0.5 * np.sin(3 * c) + 0 * r], axis=2).astype(np.float32))
    nh = 3 if mapping == "hyper-linear" else 5

    def mk(vals, lb, ub, names):
        out = {}
        for i, k in enumerate(names):
            t = float(np.mean(vals[k]))
            tt = min(max((t - lb[i]) / (ub[i] - lb[i]), 1e-4), 1 - 1e-4)
            h = np.zeros(nh, np.float32)
            h[0] = np.log(tt / (1 - tt))
            h[1:] = [0.05, -0.03] if mapping == "hyper-linear" else [0.05, 1.2, -0.03, 0.8]
            out[k] = h
        return out
    return desc, mk(g.params, GLB_P, GUB_P, synth.PARAM_NAMES), mk(g.states, GLB_S, GUB_S, synth.STATE_NAMES)


def hyper_direction(adj):
    """Direction of the hyper tangent fixtures: 0.01 per coefficient, signed like the reference gradient (well-conditioned cost_d)."""
    sg = lambda v: np.where(np.asarray(v) < 0, -0.01, 0.01).astype(np.float32)
    return ({k: sg(v) for k, v in adj["hyper_parameters_b"].items()}, {k: sg(v) for k, v in adj["hyper_states_b"].items()})


HYPER_CASES = [("gr_b_16x16x96_nse_gaps", "hyper-linear"), ("gr_c_32x32x240_d8_ragged", "hyper-polynomial"),
               ("vic_a_16x16x96_nse_gaps", "hyper-polynomial")]


def main_hyper():
    """mw_forward::hyper_forward / hyper_forward_b of the reference (mw_forward.f90:99-152) on hyper_inputs()."""
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_util as gu
    os.makedirs(os.path.join(OUT, "hyper"), exist_ok=True)
    for name, mapping in HYPER_CASES:
        g = gu.load(name)
        desc, hp, hs = hyper_inputs(g, mapping)
        kw = dict(descriptor=desc, hyper_params=hp, hyper_states=hs, mapping=mapping,
                  **{k: v for k, v in g.opts.items() if k in ("jobs_fun", "wjobs_fun", "optimize_start_step", "wgauge")})
        run = lambda **o: refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, **kw, **o)
        f, b, f3, b3 = run(), run(adjoint=True), run(fast=True), run(adjoint=True, fast=True)
        d = dict(mapping=mapping, fwd_qsim=f["qsim"], fwd_cost=np.float32(f["cost"]), adj_cost=np.float32(b["cost"]),
                 noise_qsim=np.array([rel_l2(f3["qsim"][i], f["qsim"][i]) for i in range(g.mesh.ng)]),
                 noise_cost=np.float64(abs(f3["cost"] - f["cost"]) / abs(f["cost"])))
        for k in synth.PARAM_NAMES:
            d["fwd_p_" + k] = f["parameters"][k]
            d["adj_hp_b_" + k] = b["hyper_parameters_b"][k]
            d["noise_hp_b_" + k] = np.float64(rel_l2(b3["hyper_parameters_b"][k], b["hyper_parameters_b"][k]))
        for k in synth.STATE_NAMES:
            d["fwd_s_" + k] = f["states"][k]
            d["adj_hs_b_" + k] = b["hyper_states_b"][k]
            d["noise_hs_b_" + k] = np.float64(rel_l2(b3["hyper_states_b"][k], b["hyper_states_b"][k]))
        # hyper_forward_d (mw_forward.f90:154-181) along hyper_direction()
        hd, sd = hyper_direction(b)
        t, t3 = run(hyper_params_d=hd, hyper_states_d=sd), run(hyper_params_d=hd, hyper_states_d=sd, fast=True)
        d.update(tan_cost_d=np.float32(t["cost_d"]), tan_qsim_d=t["qsim_d"],
                 noise_tan_cost_d=np.float64(abs(t3["cost_d"] - t["cost_d"]) / abs(t["cost_d"])),
                 noise_tan_qsim_d=np.array([rel_l2(t3["qsim_d"][i], t["qsim_d"][i]) for i in range(g.mesh.ng)]))
        np.savez_compressed(os.path.join(OUT, "hyper", f"{name}__{mapping}.npz"), **d)
        print(f"hyper {name} {mapping}: cost={f['cost']:.8g} |cp hyper_b|={np.abs(b['hyper_parameters_b']['cp']).max():.3g} "
              f"cost_d={t['cost_d']:.6g} noise={d['noise_tan_cost_d']:.2g}")


def main_lbfgsb():
    """Row f1: the reference's own optimize_lbfgsb (mw_optimize.f90:484-676) on the gr-b 24x24x120 case,
    distributed mapping over cp, cft, exc, lr: cost after 0..4 iterations and the final parameter fields."""
    c = [x for x in CASES if x["name"] == "gr_b_24x24x120_norm_jreg"][0]
    mesh = synth.make_mesh(c["n"], c["n"], ng=c["ng"], mask_corner=c["mask"])
    prcp, pet = synth.dense_forcing(mesh, c["nt"], gap_per_million=c["gaps"])
    P = synth.make_parameters(c["n"], c["n"])
    S = synth.make_states(c["n"], c["n"], warm=True)
    Pq = synth.make_parameters(c["n"], c["n"], perturb=0.1)
    qobs = refbind.run("gr-b", mesh, DT, prcp, pet, np.zeros((c["ng"], c["nt"]), np.float32), Pq, S)["qsim"].copy()
    op = np.zeros(16, np.int32)
    op[[1, 3, 6, 15]] = 1
    d = dict(optim_parameters=op, maxiters=np.array([0, 1, 2, 3, 4]))
    costs = []
    for it in d["maxiters"]:
        r = refbind.run("gr-b", mesh, DT, prcp, pet, qobs, P, S, optimize_maxiter=int(it), optim_parameters=op,
                        jobs_fun=("nse",), wjobs_fun=(1.0,))
        costs.append(r["cost"])
    d["costs"] = np.array(costs, np.float32)
    for k in ("cp", "cft", "exc", "lr"):
        d["final_" + k] = r["parameters"][k]
    d["qobs"] = qobs
    np.savez_compressed(os.path.join(OUT, "lbfgsb", "opt_gr_b_24x24x120.npz"), **d)
    print("lbfgsb costs:", costs)


def main_lbfgsb_cance():
    """The user guide's distributed calibration (doc/source/user_guide/quickstart/real_case_cance.rst:470-552) on the real
    Cance data: optimize_lbfgsb over cp, cft, exc, lr from the uniform SBS optimum, nse at the downstream gauge; cost after
    0..6 iterations of the all-CPU reference."""
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_util as gu
    g = gu.load("gr_a_cance_28x28x1440")
    op = np.zeros(16, np.int32)
    op[[1, 3, 6, 15]] = 1
    d = dict(optim_parameters=op, maxiters=np.array([0, 1, 2, 4, 6]))
    costs = []
    for it in d["maxiters"]:
        r = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, optimize_maxiter=int(it),
                        optim_parameters=op, **g.opts)
        costs.append(r["cost"])
    d["costs"] = np.array(costs, np.float32)
    for k in ("cp", "cft", "exc", "lr"):
        d["final_" + k] = r["parameters"][k]
    np.savez_compressed(os.path.join(OUT, "lbfgsb", "opt_gr_a_cance.npz"), **d)
    print("lbfgsb cance costs:", costs)


def main_lbfgsb_bounds_cance():
    """The reference's "adjust bounds" test (tests/core/test_simu.py:128-140) on its own catchment: distributed L-BFGS-B over cp and
    cft with cp bounded to [1, 300], one iteration from the model's default parameters; cost and the two calibrated fields."""
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_util as gu
    g = gu.load("gr_a_cance_28x28x1440")
    Pd = {k: np.asfortranarray(np.full((g.mesh.nrow, g.mesh.ncol), synth.PARAM_DEFAULTS[k], np.float32)) for k in synth.PARAM_NAMES}
    op = np.zeros(16, np.int32)
    op[[1, 3]] = 1                                       # cp, cft
    lb, ub = GLB_P.copy(), GUB_P.copy()
    lb[1], ub[1] = 1.0, 300.0
    r = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, Pd, g.states, optimize_maxiter=1, optim_parameters=op,
                    lb_parameters=lb, ub_parameters=ub, **g.opts)
    np.savez_compressed(os.path.join(OUT, "lbfgsb", "bounds_gr_a_cance.npz"), optim_parameters=op, lb_parameters=lb, ub_parameters=ub,
                        cost=np.float32(r["cost"]), final_cp=r["parameters"]["cp"], final_cft=r["parameters"]["cft"], maxiter=1)
    print("bounds cance: cost", r["cost"], "cp range", float(r["parameters"]["cp"].min()), float(r["parameters"]["cp"].max()))


def main_sbs_cance():
    """The user guide's first calibration (real_case_cance.rst:396-430): mw_optimize::optimize_sbs, uniform cp, cft, exc, lr from
    the Model() defaults on the real Cance data, nse at the downstream gauge: cost after 0, 1, 2 iterations of the all-CPU
    reference and the calibrated values at the outlet cell."""
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_util as gu
    g = gu.load("gr_a_cance_28x28x1440")
    pv = dict(ci=1e-6, cp=200.0, beta=1000.0, cft=500.0, cst=500.0, alpha=0.9, exc=0.0, lr=5.0)
    P = {k: (np.full_like(v, pv[k]) if k in pv else v) for k, v in g.params.items()}
    op = np.zeros(16, np.int32)
    op[[1, 3, 6, 15]] = 1
    costs = []
    for it in (0, 1, 2):
        r = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, P, g.states, optimize_sbs_maxiter=it, optim_parameters=op, **g.opts)
        costs.append(r["cost"])
    d = dict(optim_parameters=op, maxiters=np.array([0, 1, 2]), costs=np.array(costs, np.float32))
    for k in ("cp", "cft", "exc", "lr"):
        d["final_" + k] = np.float32(r["parameters"][k][20, 27])
    np.savez_compressed(os.path.join(OUT, "lbfgsb", "sbs_gr_a_cance.npz"), **d)
    print("sbs cance costs:", costs)


def _reference_lcurve_helpers():
    """The two pure-numpy helpers of the reference's L-curve (core/simulation/_optimize.py: `_compute_wjreg_range`,
    `_compute_best_lcurve_weight`), taken from the reference at generation time: the package itself does not import here
    (f90wrap, gdal, h5py missing), these two functions need numpy only.  Nothing of them is stored: the fixture holds
    their inputs and outputs."""
    import ast
    path = "/root/reference/smash/core/simulation/_optimize.py"
    tree = ast.parse(open(path).read())
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("_compute_wjreg_range", "_compute_best_lcurve_weight"):
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return ns["_compute_wjreg_range"], ns["_compute_best_lcurve_weight"]


def _auto_wjreg_case(tag, structure, mesh, prcp, pet, qobs, P, S, base_opts):
    """One fixture of the auto_wjreg cycles: the reference's own test configuration (tests/core/test_simu.py:143-170 -- control
    cp, cft, lr; prior + smoothing with weights 1, 2; maxiter 2; 8 L-curve cycles), every cycle run by the reference's
    optimize_lbfgsb."""
    from smash_amd.optimize import auto_wjreg_cycles
    op = np.zeros(16, np.int32)
    op[[1, 3, 15]] = 1                                   # cp, cft, lr
    kw = dict(base_opts)
    kw.update(optim_parameters=op, jreg_fun=("prior", "smoothing"), wjreg_fun=(1.0, 2.0))
    # output%cost_jobs_initial: the misfit of the first guess, from the forward run optimize_lbfgsb starts with
    # (mw_optimize.f90:567-573: normalised control, denormalize_forward on)
    Pn, Sn = norm(P, synth.PARAM_NAMES, GLB_P, GUB_P), norm(S, synth.STATE_NAMES, GLB_S, GUB_S)
    kf = dict(kw)
    kf.update(params_bgd=Pn, states_bgd=Sn, denormalize_forward=True, wjreg=0.0)
    jobs0 = refbind.run(structure, mesh, DT, prcp, pet, qobs, Pn, Sn, **kf)["cost_jobs"]
    d = dict(optim_parameters=op, qobs=qobs, maxiter=2, nb_wjreg_lcurve=8, cost_jobs_initial=np.float32(jobs0))
    for mode in ("fast", "lcurve"):
        log = []

        def run_cycle(w):
            kc = dict(kw)
            kc.update(optimize_maxiter=2, wjreg=float(w))
            r = refbind.run(structure, mesh, DT, prcp, pet, qobs, P, S, **kc)
            log.append((float(w), r["cost"], r["cost_jobs"], r["cost_jreg"]))
            run_cycle.last = r
            return dict(cost=r["cost"], cost_jobs=r["cost_jobs"], cost_jreg=r["cost_jreg"], cost_jobs_initial=jobs0)

        w, lcurve = auto_wjreg_cycles(run_cycle, lambda: None, mode, 8)
        d[mode + "_cycles"] = np.array(log, np.float64)          # wjreg, cost, cost_jobs, cost_jreg of every cycle, in order
        d[mode + "_wjreg"] = np.float64(w if w is not None else np.nan)
        d[mode + "_final_cp"] = run_cycle.last["parameters"]["cp"]
        if lcurve is not None:
            d["lcurve_distance"] = lcurve["distance"]
        print(tag, mode, "wjreg", w, "cycles", len(log), "final cost", log[-1][1])
    return d


def main_auto_wjreg():
    """Row f2: the calibration cycles of auto_wjreg = 'fast' / 'lcurve' (core/simulation/_optimize.py:257-453) on the synthetic
    gr-b case and on the real Cance data (the catchment of the reference's own test), plus input / output vectors of its two
    helpers."""
    c = [x for x in CASES if x["name"] == "gr_b_24x24x120_norm_jreg"][0]
    mesh = synth.make_mesh(c["n"], c["n"], ng=c["ng"], mask_corner=c["mask"])
    prcp, pet = synth.dense_forcing(mesh, c["nt"], gap_per_million=c["gaps"])
    P, S = synth.make_parameters(c["n"], c["n"]), synth.make_states(c["n"], c["n"], warm=True)
    Pq = synth.make_parameters(c["n"], c["n"], perturb=0.1)
    qobs = refbind.run("gr-b", mesh, DT, prcp, pet, np.zeros((c["ng"], c["nt"]), np.float32), Pq, S)["qsim"].copy()
    d = _auto_wjreg_case("gr-b 24x24", "gr-b", mesh, prcp, pet, qobs, P, S, dict(jobs_fun=("nse",), wjobs_fun=(1.0,)))
    # helper vectors straight from the reference's functions
    ref_range, ref_best = _reference_lcurve_helpers()
    wo = np.array([3.7e-4, 0.0123, 1.0, 25.0, 0.5], np.float64)
    nb = np.array([6, 7, 8, 9, 12], np.int64)
    d["range_w_opt"], d["range_nb"] = wo, nb
    for i in range(wo.size):
        d[f"range_out_{i}"] = ref_range(wo[i], int(nb[i]))
    rng = np.random.default_rng(7)
    for i in range(6):
        n = 5 + 2 * i
        jreg = np.sort(rng.uniform(0.0, 3.0, n)).astype(np.float32)
        jobs = (1.0 - 0.6 * (jreg / jreg.max()) ** rng.uniform(0.2, 2.0) + rng.normal(0, 0.03, n)).astype(np.float32)
        wj = np.concatenate([[0.0], np.sort(rng.uniform(1e-4, 1.0, n - 1))]).astype(np.float32)
        if i == 4:
            jobs[2] = jobs.max()                          # a cycle that removed nothing
        args = (jobs, jreg, wj, np.min(jobs), np.max(jobs), np.min(jreg), np.max(jreg))
        dist, best = ref_best(*args)
        d[f"pick_jobs_{i}"], d[f"pick_jreg_{i}"], d[f"pick_w_{i}"] = jobs, jreg, wj
        d[f"pick_dist_{i}"], d[f"pick_best_{i}"] = np.asarray(dist, np.float32), np.float64(np.nan if best is None else best)
    d["pick_n"] = 6
    np.savez_compressed(os.path.join(OUT, "lbfgsb", "auto_wjreg_gr_b_24x24x120.npz"), **d)
    # the real Cance data (60 days, hourly): from the model's default parameters (mwd_parameters.f90:150-167), as the reference's
    # test does, and from the fixture's first guess (the uniform SBS optimum), where two iterations remove less than 5 % of the
    # misfit and the L-curve has nothing to choose from (core/simulation/_optimize.py:332-341, 425-450)
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_util as gu
    g = gu.load("gr_a_cance_28x28x1440")
    Pd = {k: np.asfortranarray(np.full((g.mesh.nrow, g.mesh.ncol), synth.PARAM_DEFAULTS[k], np.float32)) for k in synth.PARAM_NAMES}
    dc = _auto_wjreg_case("cance defaults", g.structure, g.mesh, g.prcp, g.pet, g.qobs, Pd, g.states, dict(g.opts))
    del dc["qobs"]
    np.savez_compressed(os.path.join(OUT, "lbfgsb", "auto_wjreg_gr_a_cance.npz"), **dc)
    dc = _auto_wjreg_case("cance sbs optimum", g.structure, g.mesh, g.prcp, g.pet, g.qobs, g.params, g.states, dict(g.opts))
    del dc["qobs"]
    np.savez_compressed(os.path.join(OUT, "lbfgsb", "auto_wjreg_gr_a_cance_flat.npz"), **dc)


if __name__ == "__main__":
    if "--auto-wjreg" in sys.argv:
        main_auto_wjreg()
        sys.exit(0)
    if "--bounds" in sys.argv:
        main_lbfgsb_bounds_cance()
        sys.exit(0)
    main()                              # python make_golden.py [case names...]: only those cases
    if not [a for a in sys.argv[1:] if not a.startswith("-")]:
        os.makedirs(os.path.join(OUT, "lbfgsb"), exist_ok=True)
        main_lbfgsb()
        main_lbfgsb_cance()
        main_sbs_cance()
        main_tangent()
        main_hyper()
        main_auto_wjreg()
        main_lbfgsb_bounds_cance()
