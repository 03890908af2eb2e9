"""CPU: the weight search of the regularisation term (auto_wjreg = 'fast' | 'lcurve') in smash_amd/optimize.py
against tests/golden/lbfgsb/auto_wjreg_gr_b_24x24x120.npz: the two helpers against input / output vectors of the reference's own
functions (core/simulation/_optimize.py:911-1003), the cycle logic replayed on the costs the reference's optimize_lbfgsb returned
for every weight (tests/golden/make_golden.py::main_auto_wjreg)."""
import os

import numpy as np
import pytest

import golden_util as gu
from smash_amd.optimize import auto_wjreg_cycles, best_lcurve_weight, wjreg_range


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "auto_wjreg_gr_b_24x24x120.npz"))


def test_wjreg_range_equals_the_reference(z):
    for i in range(z["range_w_opt"].size):
        got = wjreg_range(float(z["range_w_opt"][i]), int(z["range_nb"][i]))
        ref = z[f"range_out_{i}"]
        assert got.dtype == np.float32 and np.array_equal(got, ref), (i, got, ref)
        assert got.size == max(5, int(z["range_nb"][i]) - 1)


def test_lcurve_corner_equals_the_reference(z):
    for i in range(int(z["pick_n"])):
        jobs, jreg, w = z[f"pick_jobs_{i}"], z[f"pick_jreg_{i}"], z[f"pick_w_{i}"]
        dist, best = best_lcurve_weight(jobs, jreg, w, np.min(jobs), np.max(jobs), np.min(jreg), np.max(jreg))
        ref = z[f"pick_dist_{i}"]
        assert np.array_equal(np.isnan(dist), np.isnan(ref)), i
        assert np.allclose(dist[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=0, atol=3e-7), (i, dist, ref)
        assert best == np.float32(z[f"pick_best_{i}"]), (i, best, z[f"pick_best_{i}"])
    # degenerate curves: nothing to choose from
    d, b = best_lcurve_weight([1.0, 0.9], [0.0, 1.0], [0.0, 0.1], 0.9, 1.0, 0.0, 1.0)
    assert d.size == 0 and b is None
    d, b = best_lcurve_weight([1.0, 1.0, 1.0], [0.0, 1.0, 2.0], [0.0, 0.1, 0.2], 1.0, 1.0, 0.0, 2.0)
    assert d.size == 0 and b is None


CASES = ["auto_wjreg_gr_b_24x24x120", "auto_wjreg_gr_a_cance", "auto_wjreg_gr_a_cance_flat"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["fast", "lcurve"])
def test_cycle_logic_replayed_on_the_reference_costs(case, mode):
    """run_cycle answers with what the reference's optimize_lbfgsb returned for that weight: the weights asked for, their order,
    the restores in between and the weight chosen must be the recorded ones.  Synthetic gr-b case, the real Cance data from the
    model's default parameters, and Cance from the SBS optimum, where the L-curve finds nothing to try."""
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", case + ".npz"))
    rec = z[mode + "_cycles"]
    asked, restores = [], []

    def run_cycle(w):
        i = len(asked)
        assert i < len(rec) and np.float32(w) == np.float32(rec[i, 0]), (i, w, rec[i, 0])
        asked.append(float(w))
        return dict(cost=rec[i, 1], cost_jobs=rec[i, 2], cost_jreg=rec[i, 3], cost_jobs_initial=float(z["cost_jobs_initial"]))

    w, lcurve = auto_wjreg_cycles(run_cycle, lambda: restores.append(len(asked)), mode, int(z["nb_wjreg_lcurve"]))
    assert len(asked) == len(rec)
    if np.isnan(z[mode + "_wjreg"]):                       # nothing chosen: one cycle, the first guess restored, no final cycle
        assert w is None and len(rec) == 1 and restores == [1] and lcurve["wjreg_fast"] == 0.0 and lcurve["distance"].size == 0
        return
    assert np.float32(w) == np.float32(z[mode + "_wjreg"])
    assert restores == list(range(1, len(rec)))            # the first guess is restored before every cycle but the first
    if mode == "fast":
        assert lcurve is None and w == pytest.approx((float(z["cost_jobs_initial"]) - rec[0, 2]) / rec[0, 3])
    else:
        assert np.array_equal(np.isnan(lcurve["distance"]), np.isnan(z["lcurve_distance"]))
        assert np.allclose(np.nan_to_num(lcurve["distance"]), np.nan_to_num(z["lcurve_distance"]), atol=3e-7)
        assert lcurve["wjreg"].size == int(z["nb_wjreg_lcurve"]) and lcurve["wjreg"][0] == 0.0
        assert w in lcurve["wjreg"] and lcurve["wjreg_lcurve_opt"] == w


def test_lcurve_without_a_usable_first_cycle():
    """The unregularised calibration removed < 5 % of the misfit: no weights are tried, nothing is chosen, the model is left as
    it is (core/simulation/_optimize.py:332-341, 425-450)."""
    calls = []

    def run_cycle(w):
        calls.append(w)
        return dict(cost=0.99, cost_jobs=0.99, cost_jreg=0.5, cost_jobs_initial=1.0)

    w, lcurve = auto_wjreg_cycles(run_cycle, lambda: None, "lcurve", 6)
    assert calls == [0.0] and w is None and lcurve["wjreg_fast"] == 0.0 and lcurve["distance"].size == 0
    with pytest.raises(ValueError):
        auto_wjreg_cycles(run_cycle, lambda: None, "slow", 6)
