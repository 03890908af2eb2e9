"""CPU: the parts of bench.py that need no GPU -- the chunk plan every rank of a decomposition must agree on, the roofline
arithmetic, and the self-launch of N ranks (`python bench.py --gpus 2 --backend gloo` with no launcher): both ranks start as
fresh children, complete the host-side rendezvous, and then stop loudly because there is no HIP device (libsmashx has no CPU path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def test_chunk_plan_is_a_function_of_shared_quantities_only():
    import bench
    hbm = 309e9                                                    # 288 GiB card
    cells = 2048 * 1024
    # the metric's tile: compact forcing 40 GB -> two storage chunks; fp32 rows 147 GB -> four
    c = bench.chunk_plan(8760, cells, hbm, cells * 8760 * (2 + 4 / 24), "gr-b")
    assert c % 16 == 0 and -(-8760 // c) == 2
    r = bench.chunk_plan(8760, cells, hbm, cells * 8760 * 8.0, "gr-b")
    assert r % 16 == 0 and -(-8760 // r) == 4
    # the 1024^2 tile stays store-all; a forced length is taken as is; more taped levels (gr-c) never lengthen a chunk
    assert bench.chunk_plan(8760, 1024 * 1024, hbm, 1024 * 1024 * 8760 * 2.17, "gr-b") >= 8760
    assert bench.chunk_plan(8760, cells, hbm, 0.0, "gr-b", forced=1104) == 1104
    assert bench.chunk_plan(8760, cells, hbm, 40e9, "gr-c") <= c


def test_roofline_uses_the_cell_steps_one_launch_processes():
    """Sub-chunked launches are shorter than the period: the per-launch figures must come from the cell-step counters of the
    launches themselves (the roofline of a 4-rank rehearsal was 8x too high)."""
    import bench
    cs = 1024.0 * 1024 * 8760
    tm = {"vert_fwd_ms": 40.0, "route_fwd_ms": 25.0, "route_adj_ms": 28.0, "vert_adj_ms": 72.0, "sweep_ms": 170.0,
          "vert_fwd_launches": 8, "route_fwd_launches": 16, "route_adj_launches": 16, "vert_adj_launches": 8,
          "vert_fwd_cellsteps": cs, "route_fwd_cellsteps": cs, "route_adj_cellsteps": cs, "vert_adj_cellsteps": cs, "n_chunks": 1}
    r = bench.roofline(tm, True, "gr-b", [1024, 1024])
    assert r["kernel"] == "sx_k_vert_adj" and r["launches_per_step"] == 8
    assert abs(r["cellsteps_per_launch"] - cs / 8) < 1 and abs(r["algorithmic_bytes_per_launch"] - 8 * cs / 8) < 8
    assert abs(r["achieved"] - 8 * cs / 72e-3 / 1e9) < 1e-6 * r["achieved"]          # same GB/s as one whole-period launch of 72 ms
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    if "valu" in r:
        assert 0 < r["valu"]["frac"] < 1.5
        if "weighted" in r["valu"]:          # fp64 / transcendental instructions only ever raise the ceiling's time
            assert r["valu"]["weighted"]["issue_bound_ms"] >= r["valu"]["issue_bound_ms"]


def test_pmc_profile_is_matched_to_the_workload():
    """A committed counter file is only attached to a run of the workload it was taken on (the 2048^2 line carried the
    1024^2 store-all traffic figure)."""
    import bench
    src, prof = bench.pmc_profile([1024, 1024], 1)
    assert src is not None and "sx_k_vert_adj" in prof
    src2, prof2 = bench.pmc_profile([4096, 4096], 1)
    assert src2 is None and prof2 == {}


def test_default_single_gpu_workload_is_the_largest_configuration():
    import bench
    a = bench.parse([])
    assert a.gpus == 1 and a.grid == 0 and a.secondary_grid == 1024 and a.mesh == "synth"
    assert not (a.no_forward_only or a.no_real_d8 or a.no_exact)          # the line carries configs[1], the real river network, the exact build
    p = bench.parse(["--profile"])
    assert p.no_secondary and p.no_tile_solo and p.no_exact and p.no_cpu_baseline and p.no_inclusive and p.no_forward_only and p.no_real_d8


def test_forward_only_runs_find_their_own_counter_file():
    """The forward-only case (BASELINE.json configs[1]) has its own PMC passes: the untaped forward kernel of a forward sweep is not
    priced with the traffic of the taped one of an adjoint sweep (and the other way round)."""
    import bench
    src, prof = bench.pmc_profile([1024, 1024], 1, forward_only=True)
    assert src is not None and "forward_only" in src and "sx_k_vert_fwd_untaped" in prof and "sx_k_vert_adj" not in prof
    src2, prof2 = bench.pmc_profile([1024, 1024], 1)
    assert src2 != src and "sx_k_vert_adj" in prof2
    cs = 1024.0 * 1024 * 8760
    tm = {"vert_fwd_ms": 30.0, "route_fwd_ms": 17.0, "route_adj_ms": 0.0, "vert_adj_ms": 0.0, "sweep_ms": 48.0,
          "vert_fwd_launches": 1, "route_fwd_launches": 2, "route_adj_launches": 0, "vert_adj_launches": 0,
          "vert_fwd_cellsteps": cs, "route_fwd_cellsteps": cs, "route_adj_cellsteps": 0.0, "vert_adj_cellsteps": 0.0, "n_chunks": 1}
    r = bench.roofline(tm, False, "gr-b", [1024, 1024])
    assert r["kernel"] == "sx_k_vert_fwd" and abs(r["algorithmic_bytes_per_launch"] - 8 * cs) < 8       # 8 B per cell-step: prcp + pet once
    assert r["traffic"] is not None and 5.5 * cs < r["traffic"] < 7.0 * cs                                 # the compact forcing + qt: 6.2 B moved
    assert abs(r["sweep_frac"] - 8 * cs / 48e-3 / 1e9 / 8000.0) < 1e-9


def test_bench_self_launch_reaches_the_rendezvous_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a HIP device is present (tests/test_gpu_rccl.py covers the real run)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu-baseline",
                        "--tile-rows", "32", "--tile-cols", "32", "--nt", "48"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert r.stderr.count("handshake ok (gloo)") == 2 and "rank 0/2" in r.stderr and "rank 1/2" in r.stderr
    assert "no HIP device" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())     # (gloo prints a banner on stdout)


def test_pmc_summary_divides_every_family_by_the_cell_steps_it_swept(tmp_path):
    """tools/pmc_summary.py on a synthetic counter file: FETCH_SIZE (KiB, x2 on gfx950) + WRITE_SIZE summed over all dispatches of a
    kernel family and divided by the cell-steps bench.py says that family swept -- taped / untaped forward, reverse, and the copy
    passes of the chained launches (gather: every forward pass, scatter: the reverse pass)."""
    import json
    root = ROOT
    acct = {"grid": [64, 64], "n_chunks": 2, "chunk_steps": 48, "cellsteps": 64 * 64 * 96, "adjoint_sweeps": 2, "forward_sweeps": 1,
            "per_adjoint_sweep": {"taped_forward": 1000.0, "untaped_forward": 500.0, "reverse": 1000.0},
            "per_forward_sweep": {"taped_forward": 0.0, "untaped_forward": 1000.0}}
    bench = tmp_path / "bench.json"
    bench.write_text(json.dumps({"profile_accounting": acct, "config": {"workload": "synthetic"}}))
    rows = [("void sx_k_vert_fwd<2, true, true>(SxDeviceArrays, int, int)", 10.0, 20.0),        # taped forward: 2000 cell-steps
            ("void sx_k_vert_fwd<2, false, true>(SxDeviceArrays, int, int)", 4.0, 8.0),         # untaped: 2 x 500 + 1000 = 2000
            ("void sx_k_vert_adj<2, true>(SxDeviceArrays, int, int)", 6.0, 2.0),                # reverse: 2000
            ("void sx_k_chain_transpose<true>(SxDeviceArrays, SxStageTables, int, int)", 1.0, 2.0),    # gather: 4000
            ("void sx_k_chain_transpose<false>(SxDeviceArrays, SxStageTables, int, int)", 2.0, 4.0)]   # scatter: 2000
    for i, counter in enumerate(("FETCH_SIZE", "WRITE_SIZE"), 1):
        d = tmp_path / f"pmc{i}" / "host"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for name, fetch, write in rows:
                for _ in range(2):                                 # two dispatches each: summed
                    f.write(f"\"{name}\",{counter},{fetch if counter == 'FETCH_SIZE' else write}\n")
    out = tmp_path / "out.json"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(tmp_path), "--bench", str(bench), "--json", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(out.read_text())

    def per(fetch, write, cs):
        return (2.0 * 2 * fetch + 2 * write) * 1024.0 / cs
    assert d["sx_k_vert_fwd"]["hbm_bytes_per_cellstep_corrected"] == pytest.approx(per(10.0, 20.0, 2000.0))
    assert d["sx_k_vert_fwd_untaped"]["hbm_bytes_per_cellstep_corrected"] == pytest.approx(per(4.0, 8.0, 2000.0))
    assert d["sx_k_vert_adj"]["hbm_bytes_per_cellstep_corrected"] == pytest.approx(per(6.0, 2.0, 2000.0))
    assert d["sx_k_chain_transpose_gather"]["hbm_bytes_per_cellstep_corrected"] == pytest.approx(per(1.0, 2.0, 4000.0))
    assert d["sx_k_chain_transpose_scatter"]["hbm_bytes_per_cellstep_corrected"] == pytest.approx(per(2.0, 4.0, 2000.0))
    assert d["sx_k_vert_fwd"]["dispatches"] == 2 and d["workload"]["forward_only"] is False
