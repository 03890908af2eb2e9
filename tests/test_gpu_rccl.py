"""GPU: the multi-process tile path over REAL RCCL (BASELINE.json configs[4] in the small).  Every rank is its
own process (torch.distributed.run, backend nccl); the boundary discharge series travel as grouped ncclSend / ncclRecv posted by
libsmashx on its routing stream (smashx_set_exchange) or, for comparison, through the torch.distributed callback.  Each rank
checks bit-identity with the single-domain run (tests/mp_tile_worker.py).  With fewer GPUs than ranks the ranks share the card
(every rank claims its own host id, so RCCL's duplicate-GPU check passes and the loopback socket transport carries the data):
the RCCL code paths -- communicator set-up, grouping, stream ordering, buffer reuse across sweeps -- are the production ones,
only the wire differs from xGMI."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(world, extra, port_off):
    port = 29600 + (os.getpid() % 1500) + port_off
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "mp_tile_worker.py")] + extra
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-3000:])
    sys.stderr.write(r.stderr[-3000:])
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("BIT-IDENTICAL") == world


@pytest.mark.parametrize("world,chunk,pipe,exchange", [(2, 96, 16, "rccl"), (4, 32, 16, "rccl"), (2, 96, 32, "torch")])
def test_ranks_over_rccl_equal_single_domain(world, chunk, pipe, exchange):
    _launch(world, ["--chunk", str(chunk), "--pipe", str(pipe), "--exchange", exchange], world)


def test_subcatchment_parts_over_rccl_equal_single_domain():
    """All eight D8 codes: the river tree cut into 3 parts (owner masks), one process per part, native exchange."""
    _launch(3, ["--case", "gr_c_32x32x240_d8_ragged", "--cut", "sub", "--chunk", "96", "--pipe", "16"], 7)


def test_regularisation_and_median_across_ranks_over_rccl():
    """Cost terms that span the ranks (smashx.h): the regulariser's ordered sums over the whole grid on every rank of a 2 x 2
    decomposition (cost_jreg and gradients bit-identical to the single domain), and the median over negative-weight gauges that sit
    on different ranks, its slots summed by ncclAllReduce on the routing stream between the two phases of the cost kernel."""
    _launch(4, ["--case", "gr_b_24x24x120_norm_jreg", "--opts", "--chunk", "64", "--pipe", "16"], 11)
    _launch(2, ["--case", "gr_b_16x16x96_median2", "--opts", "--cut", "sub", "--chunk", "96", "--pipe", "32"], 13)


def test_calibration_across_ranks_over_rccl():
    """smash_amd.optimize_lbfgsb(decomposition=tiles.TorchDecomposition) on two processes: rank 0 runs the library's L-BFGS-B, every
    trial point is one collective forward_b (boundary series over RCCL, costs and gradients put together by all-reduce), with the
    normalised control and the regularisers of the fixture -- same iterations, evaluations and costs as the single-domain calibration."""
    _launch(2, ["--case", "gr_b_24x24x120_norm_jreg", "--opts", "--calibrate", "2", "--chunk", "64", "--pipe", "16"], 17)


def test_tangent_model_across_ranks_over_rccl():
    """smashx_forward_d on a decomposition with one process per part: the boundary series of the value pass and of the tangent pass as
    grouped ncclSend / ncclRecv on the routing stream (rectangles, and sub-catchment parts of a D8 mesh); discharge and its tangent at
    every rank's gauges bit-identical to the single domain, cost_d to the rounding of the sum over the ranks."""
    _launch(2, ["--chunk", "32", "--pipe", "16", "--tangent"], 19)
    _launch(3, ["--case", "gr_c_32x32x240_d8_ragged", "--cut", "sub", "--chunk", "96", "--pipe", "16", "--tangent"], 23)


def test_bench_self_launch_two_ranks():
    """python bench.py --gpus 2 with no launcher starts both ranks itself and prints one JSON line with n_gpus = 2."""
    import json
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--tile-rows", "64", "--tile-cols", "64", "--nt", "240", "--pipe", "64", "--ng", "4", "--raw-forcing"],
                       capture_output=True, text=True, timeout=900)
    sys.stderr.write(r.stderr[-3000:])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["grid"] == [64, 128]
    assert "ncclSend" in line["config"]["parallelism"] and line["value"] > 0


_TEARDOWN = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import torch
torch.zeros(1, device="cuda:0")
import golden_util as gu
import smash_amd
from smash_amd.solver import Comm, Solver
from test_gpu_parity import _run_adjoint, _types
g = gu.load("gr_b_16x16x96_nse_gaps")
ref = _run_adjoint(g)[2].qsim.copy()
empty = np.zeros(0, np.int32)

def plan():
    setup, mesh, inp, par, sta, out = _types(g)
    sol = Solver(setup, mesh, chunk_steps=32, device=0)
    sol.set_forcing(g.prcp, g.pet); sol.set_qobs(g.qobs); sol.set_options(setup.optimize)
    def run():
        sol.upload(par, sta); sol.sweep(True, 1.0); sol.download(True, par, sta, out, par.copy(), sta.copy())
        return out.qsim.copy()
    return sol, run

c1, c2 = Comm(Comm.unique_id(), 0, 1, 0), Comm(Comm.unique_id(), 0, 1, 0)
# unset, destroy the plan, then the communicator (the communicator used to keep the freed plan in its list)
sol, run = plan()
sol.set_exchange(c1, empty, empty); sol.set_exchange(None, empty, empty); sol.close(); c1.close()
# switch communicators: the first one must forget the plan, the plan must keep the second
c1 = Comm(Comm.unique_id(), 0, 1, 0)
sol, run = plan()
sol.set_exchange(c1, empty, empty); sol.set_exchange(c2, empty, empty); c1.close()
assert np.array_equal(run(), ref)               # sweeps (their stall agreement runs on c2) still work
sol.close(); c2.close()
print("TEARDOWN-OK")
"""


def test_communicator_teardown_orders():
    """smashx_set_exchange(plan, NULL) and a switch to another communicator take the plan off the previous communicator's list:
    set, unset, destroy plan, destroy communicator -- and set c1, set c2, destroy c1, sweep, destroy -- with one-rank communicators."""
    r = subprocess.run([sys.executable, "-c", _TEARDOWN, os.path.dirname(HERE)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    sys.stderr.write(r.stderr[-3000:])
    assert r.returncode == 0 and "TEARDOWN-OK" in r.stdout, r.stderr[-2000:]
