#!/usr/bin/env python3
"""Timeline model of one forward + adjoint sweep of a tile decomposition: what the rank graph, the pipeline sub-chunks and
the fill of the chained routing rounds do to the sweep time on N GPUs -- which cannot be measured here (one-GPU boxes).

    python tools/pipeline_model.py [--out profiles/r3_pipeline_model.json]

A discrete-event simulation of the very launch sequence smashx_sweep issues (smash_amd/csrc/smashx.hip, sweep_once): per rank a V stream
(vertical kernels) and an R stream (round 0, chained rounds, pack / send), the dependencies between them, the storage chunks with their
recomputation, and the messages along the rank graph (forward: downstream after the sender's chained launch of the sub-chunk; reverse:
upstream after the sender's round 0).  Kernels that run together on one GPU share it (processor sharing with a co-execution bonus):
that is what the ONE-GPU rehearsals measure (N processes on one card), and it is how the model is validated -- same code path, same
parameters, `shared=True`; the N-GPU prediction is the same simulation with one GPU per rank.

Kernel costs come from this round's measurements (profiles/r3_*): per cell-step rates of the vertical kernels and of round 0, and
the chained launch as (time blocks + longest cell path of the tile) x the cost of a super-step -- the fill every sub-chunk launch pays.
Three parameters are fitted to the single-GPU measurements (solo tile at three sub-chunk lengths, rehearsals with 2 / 4 / 6 ranks): the
share of a GPU a chained launch occupies, the co-execution bonus of overlapping streams, and the latency of a message."""
from __future__ import annotations

import argparse
import heapq
import itertools
import json
import os

import numpy as np

CS = 9.185525760e9            # cell-steps of the 1024^2 x 8760 case: the unit the rates below are quoted on

# measured kernel times, ms per CS cell-steps (profiles/r3_rocprofv3_kernel_stats_2048.csv: launches of 2192 steps x 4.19 M cells;
# profiles/r2_rocprofv3_kernel_stats.csv for the store-all 1024^2 case)
RATES = {
    "vf_u": 27.1, "vf_t": 34.4, "va": 60.4,                  # vertical: forward untaped / taped, reverse (hi tape on) -- round 4, final schedule (profiles/r4_2048_kernel_stats.csv)
    "r0f_u": 8.91, "r0f_t": 12.68, "r0a": 17.24,             # round 0 at 548 time blocks per launch (2048^2), forward untaped / taped, reverse
    "r0_nb_ref": 548, "r0_depth": 160,                       # ... whose groups are up to 160 stages deep: a launch of nb blocks runs nb + depth super-steps
    "tau_f": 1.30e-3, "tau_a": 1.31e-3,                      # ms per super-step of a chained launch WITH its copy pass (2048^2: 4.88 + 3.05 / 5.16 + 2.82 ms for 548 blocks + 5530 stages)
    "path_per_side": 1.35,                                   # longest cell path of a tile ~ 1.35 x (rows + cols) stages (2900 at 1024^2, 5250 at 2048^2)
    # ... except where the chained launches of the very tile were timed (per-round timelines, --trace-groups): the 2048 x 1024 tile of
    # the 8-rank split has 7 routing rounds and its chained launch of a 1104-step sub-chunk takes 2.7 ms forward / 3.2 ms reverse
    # = 276 blocks + ~1680 stages (the rule above would say 4147 and made the model 11 % pessimistic on that tile's sub-chunks)
    "path_measured": {(2048, 1024): 1680.0},
}


class Task:
    __slots__ = ("name", "rank", "stream", "work", "demand", "deps", "lat_deps", "done_at", "left", "started", "ndeps")

    def __init__(self, name, rank, stream, work, demand):
        self.name, self.rank, self.stream, self.work, self.demand = name, rank, stream, work, demand
        self.deps, self.lat_deps = [], []          # tasks that must have ended; ... ended at least `lat` ms ago (messages)
        self.done_at, self.left, self.started = None, work, None


def ramp_schedule(T, pipe, ramp):
    """Sub-chunk lengths of a storage chunk of T steps (multiples of 16 except the last): `pipe` steps each, but the first and the last
    `pipe` steps are cut into ramp pieces each -- short sub-chunks where a sweep direction starts (the rank pipeline fills faster),
    long ones in between (fewer fills of the chained routing rounds).  ramp <= 1: equal lengths (what smashx_sweep does).  Evaluated
    with this model and NOT built: every extra sub-chunk pays the fill of a chained launch (6 + 7 ms on a 2048 x 1024
    tile), which is what a short sub-chunk saves the rank pipeline: N = 8 at sub-chunks of 2192 steps 601 -> 603 / 646 / 696 ms for
    ramp 2 / 3 / 4."""
    n = max(1, (T + pipe // 2) // pipe)
    tp = -(-T // n)
    tp = -(-tp // 16) * 16
    if ramp <= 1 or n < 2:
        return [min(tp, T - j * tp) for j in range(n) if T - j * tp > 0]
    small = max(16, (tp // ramp) // 16 * 16)
    head = [small] * ramp
    head[-1] = tp - small * (ramp - 1)
    mid = T - 2 * tp
    out = list(head)
    while mid > 0:
        out.append(min(tp, mid))
        mid -= tp
    out += head[::-1]
    # fix the total (rounding): the piece before the tail absorbs the difference
    d = T - sum(out)
    out[len(head)] += d if len(out) > 2 * len(head) else 0
    if len(out) == 2 * len(head):
        out[len(head) - 1] += d
    return [v for v in out if v > 0]


SCHEDULE = {"ramp": 1, "keep_inlets": True}
# keep_inlets (round 4): a rank keeps the inlet series it received for a storage chunk in the first pass, so the recomputation of that
# chunk in the reverse sweep waits for no neighbour and sends nothing (smashx.hip, inlet_series); False = round 3's sweep, which
# exchanged the series again.  Also round 4: the V stream waits for the R stream sub-chunk by sub-chunk between storage chunks
# (buf_free), not for the whole stream.


def build(N, tile, nt, chunk, pipe, graph, lat, delta):
    """Tasks of one sweep for every rank.  graph[r] = ranks upstream of r (it receives their boundary discharge)."""
    nr, nc = tile
    cells = nr * nc
    nch = -(-nt // chunk)
    P = RATES["path_measured"].get((nr, nc), RATES["path_per_side"] * (nr + nc))
    down = {r: [s for s in range(N) if r in graph[s]] for r in range(N)}
    tasks = []
    last = {}            # (rank, stream) -> last task queued on that stream
    book = {}            # (kind, rank, pass id, sub-chunk) -> task

    def add(name, r, stream, work, demand, deps=(), lat_deps=()):
        t = Task(name, r, stream, work, demand)
        if (r, stream) in last:
            t.deps.append(last[(r, stream)])
        t.deps += [d for d in deps if d is not None]
        t.lat_deps += [d for d in lat_deps if d is not None]
        last[(r, stream)] = t
        tasks.append(t)
        return t

    def subs(c):
        T = min(chunk, nt - c * chunk)
        if SCHEDULE["ramp"] > 1:
            return ramp_schedule(T, pipe, SCHEDULE["ramp"])
        n = max(1, (T + pipe // 2) // pipe)
        tp = -(-T // n)
        return [min(tp, T - j * tp) for j in range(n)]

    def r0_scale(T):
        nb = T / 4.0
        return (1.0 + RATES["r0_depth"] / nb) / (1.0 + RATES["r0_depth"] / RATES["r0_nb_ref"])

    order_f = topo(N, graph)      # producers are queued before their consumers: upstream ranks first going forward, downstream first going back
    order_b = order_f[::-1]

    pid = 0
    for c in range(nch):
        _forward(pid, c, c == nch - 1, order_f, N, graph, subs, add, book, cells, P, delta, last, r0_scale, True, c > 0)
        pid += 1
    for c in range(nch - 1, -1, -1):
        if c < nch - 1:
            _forward(pid, c, True, order_f, N, graph, subs, add, book, cells, P, delta, last, r0_scale, not SCHEDULE["keep_inlets"], False)
            pid += 1
        _reverse(pid, c, order_b, N, down, subs, add, book, cells, P, delta, r0_scale)
        pid += 1
    return tasks


def _forward(pid, c, taped, order, N, graph, subs, add, book, cells, P, delta, last, r0_scale, exchange, after_forward):
    """exchange: the inlet series come from the upstream ranks (False: kept from the first pass); after_forward: the previous pass over
    the chunk buffers was a forward pass, whose routing of sub-chunk j must have finished before V(j) overwrites its part."""
    for r in order:
        vs = []
        for j, T in enumerate(subs(c)):
            w = RATES["vf_t" if taped else "vf_u"] * cells * T / CS
            prev = book.get(("CHf", r, pid - 1, j)) if after_forward else None
            vs.append(add(f"Vf{pid}.{j}", r, "V", w, 1.0, deps=[prev] if prev is not None else ()))
        for j, T in enumerate(subs(c)):
            up = [book[("CHf", s, pid, j)] for s in graph[r]] if exchange else []
            add(f"R0f{pid}.{j}", r, "R", RATES["r0f_t" if taped else "r0f_u"] * cells * T / CS * r0_scale(T), 1.0, deps=[vs[j]], lat_deps=up)
            book[("CHf", r, pid, j)] = add(f"CHf{pid}.{j}", r, "R", (T / 4.0 + P) * RATES["tau_f"], delta)


def _reverse(pid, c, order, N, down, subs, add, book, cells, P, delta, r0_scale):
    for r in order:
        ss = subs(c)
        for j in range(len(ss) - 1, -1, -1):
            T = ss[j]
            dn = [book[("R0a", s, pid, j)] for s in down[r]]
            add(f"CHa{pid}.{j}", r, "R", (T / 4.0 + P) * RATES["tau_a"], delta, lat_deps=dn)
            r0 = add(f"R0a{pid}.{j}", r, "R", RATES["r0a"] * cells * T / CS * r0_scale(T), 1.0)
            book[("R0a", r, pid, j)] = r0
            add(f"Va{pid}.{j}", r, "V", RATES["va"] * cells * T / CS, 1.0, deps=[r0])


def topo(N, graph):
    order, seen = [], set()

    def visit(r):
        if r in seen:
            return
        seen.add(r)
        for s in graph[r]:
            visit(s)
        order.append(r)
    for r in range(N):
        visit(r)
    return order


def simulate(tasks, gpu_of, kappa, lat):
    """Processor sharing per GPU: the running tasks of a GPU advance at rate min(1, kappa / sum of their demands)."""
    t = 0.0
    pending = set(range(len(tasks)))
    index = {id(x): i for i, x in enumerate(tasks)}
    running = []
    stream_busy = {}
    ready_time = {}

    def can_start(x):
        if any(d.done_at is None for d in x.deps) or any(d.done_at is None for d in x.lat_deps):
            return None
        return max([0.0] + [d.done_at for d in x.deps] + [d.done_at + lat for d in x.lat_deps])

    # streams are in-order: only the head of each stream can start
    heads = {}
    for i, x in enumerate(tasks):
        heads.setdefault((x.rank, x.stream), []).append(i)
    pos = {k: 0 for k in heads}
    while True:
        # start every stream head that is ready at time t
        started = True
        while started:
            started = False
            for k, lst in heads.items():
                if pos[k] < len(lst) and stream_busy.get(k) is None:
                    x = tasks[lst[pos[k]]]
                    rt = can_start(x)
                    if rt is not None and rt <= t + 1e-12:
                        x.started = t
                        running.append(x)
                        stream_busy[k] = x
                        started = True
        if not running:
            # nothing runs: jump to the earliest time a head becomes ready
            nxt = None
            for k, lst in heads.items():
                if pos[k] < len(lst):
                    rt = can_start(tasks[lst[pos[k]]])
                    if rt is not None:
                        nxt = rt if nxt is None else min(nxt, rt)
            if nxt is None:
                break
            t = max(t, nxt)
            continue
        # rates
        load = {}
        for x in running:
            load[gpu_of[x.rank]] = load.get(gpu_of[x.rank], 0.0) + x.demand
        rate = {g: min(1.0, kappa / max(l, 1e-9)) for g, l in load.items()}
        # next event: a completion, or a head becoming ready through a latency
        dt = min(x.left / rate[gpu_of[x.rank]] for x in running)
        for k, lst in heads.items():
            if pos[k] < len(lst) and stream_busy.get(k) is None:
                rt = can_start(tasks[lst[pos[k]]])
                if rt is not None and rt > t:
                    dt = min(dt, rt - t)
        for x in running:
            x.left -= dt * rate[gpu_of[x.rank]]
        t += dt
        for x in [x for x in running if x.left <= 1e-9]:
            x.done_at = t
            running.remove(x)
            k = (x.rank, x.stream)
            stream_busy[k] = None
            pos[k] += 1
    assert all(x.done_at is not None for x in tasks), "deadlock in the task graph"
    return t


def rect_graph(pr, pc):
    """E / SE / S drainage: tile (i, j) receives from (i-1, j), (i, j-1), (i-1, j-1)."""
    g = {}
    for i in range(pr):
        for j in range(pc):
            g[i * pc + j] = [a * pc + b for a, b in ((i - 1, j), (i, j - 1), (i - 1, j - 1)) if a >= 0 and b >= 0]
    return g


def sweep_ms(N, pr, pc, tile, nt, chunk, pipe, shared, p):
    graph = rect_graph(pr, pc)
    tasks = build(N, tile, nt, chunk, pipe, graph, p["lat"], p["delta"])
    gpu_of = {r: (0 if shared else r) for r in range(N)}
    return simulate(tasks, gpu_of, p["kappa"], p["lat"])


# measurements the model is held to: (label, N, pr, pc, tile, chunk, pipe, shared GPU, measured ms, source)
MEASURED = [
    ("solo 2048x1024 tile, no sub-chunks", 1, 1, 1, (2048, 1024), 4384, 4384, True, 323.4, "profiles/r4_solo_p4384_final_schedule.json"),
    ("solo 2048x1024 tile, sub-chunks of 2192", 1, 1, 1, (2048, 1024), 4384, 2192, True, 325.1, "profiles/r4_solo_p2192_final_schedule.json"),
    ("solo 2048x1024 tile, sub-chunks of 1104", 1, 1, 1, (2048, 1024), 4384, 1104, True, 336.0, "profiles/r4_solo_p1104_final_schedule.json"),
    ("2048^2 single domain, 4 storage chunks", 1, 1, 1, (2048, 2048), 2192, 2192, True, 694.5, "profiles/r4_bench_2048x2048x8760.json (674-700 across boxes)"),
    ("1024^2 single domain, store-all", 1, 1, 1, (1024, 1024), 8768, 8768, True, 149.7, "profiles/r4_bench_2048x2048x8760.json (secondary)"),
    ("rehearsal: 2 ranks on ONE GPU, 1x2 tiles of 1024x512", 2, 1, 2, (1024, 512), 8768, 2192, True, 155.8, "profiles/r4_reh2_final_schedule.json"),
    ("rehearsal: 4 ranks on ONE GPU, 2x2 tiles of 512x512", 4, 2, 2, (512, 512), 8768, 2192, True, 154.8, "profiles/r4_reh4_final_schedule.json"),
    ("rehearsal: 6 ranks on ONE GPU, 1x6 tiles of 1024x176", 6, 1, 6, (1024, 176), 8768, 1104, True, 240.5, "profiles/r4_reh6_final_schedule.json"),
    # round 4: CHUNKED rehearsals (4 storage chunks: three of them recomputed in the reverse sweep -- without exchange since this round)
    ("rehearsal: 2 ranks on ONE GPU, 1x2 tiles of 1024x512, 4 storage chunks", 2, 1, 2, (1024, 512), 2192, 1104, True, 198.4, "profiles/r4_reh2_chunked_final_schedule.json"),
    ("rehearsal: 4 ranks on ONE GPU, 2x2 tiles of 512x512, 4 storage chunks", 4, 2, 2, (512, 512), 2192, 1104, True, 184.1, "profiles/r4_reh4_chunked_final_schedule.json"),
]


def fit():
    best = None
    for delta, kappa, lat, pps in itertools.product((0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5), (1.0, 1.05, 1.1, 1.15, 1.2, 1.3), (0.05, 0.2, 0.5, 1.0, 2.0), (0.7, 1.0, 1.35)):
        p = {"delta": delta, "kappa": kappa, "lat": lat, "path_per_side": pps}
        RATES["path_per_side"] = pps
        err = []
        for (_, N, pr, pc, tile, chunk, pipe, shared, ms, _src) in MEASURED:
            err.append(sweep_ms(N, pr, pc, tile, 8760, chunk, pipe, shared, p) / ms - 1.0)
        score = float(np.sqrt(np.mean(np.square(err))))
        if best is None or score < best[0]:
            best = (score, p, err)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    score, p, err = fit()
    RATES["path_per_side"] = p["path_per_side"]
    rep = {"what": __doc__.split("\n\n")[0], "rates_ms_per_9.19e9_cellsteps": {k: (v if not isinstance(v, dict) else {str(a): b for a, b in v.items()}) for k, v in RATES.items()}, "fitted": p, "rms_relative_error": score, "validation": [], "prediction": []}
    for (label, N, pr, pc, tile, chunk, pipe, shared, ms, src), e in zip(MEASURED, err):
        rep["validation"].append({"case": label, "measured_ms": ms, "model_ms": round(ms * (1.0 + e), 1), "error": round(e, 3), "source": src})
    # the metric's decomposition: 2048 x 1024 cells per GPU, 2 storage chunks; N = 1 x 2, 2 x 2, 2 x 4 tiles
    solo = sweep_ms(1, 1, 1, (2048, 1024), 8760, 4384, 1104, True, p)
    for N, pr, pc in ((2, 1, 2), (4, 2, 2), (8, 2, 4)):
        for pipe in (4384, 2192, 1104, 560):
            t = sweep_ms(N, pr, pc, (2048, 1024), 8760, 4384, pipe, False, p)
            rep["prediction"].append({"n_gpus": N, "tiles": [pr, pc], "grid": [pr * 2048, pc * 1024], "pipe_steps": pipe, "sweep_ms": round(t, 1),
                                      "cell_timesteps_per_s": round(N * 2048 * 1024 * 8760 / (t * 1e-3), -8),
                                      "efficiency_vs_solo_tile": round(solo / t, 3)})
    rep["solo_tile_model_ms"] = round(solo, 1)
    # what keeping the received inlet series buys: the same predictions with the recomputation exchanging them again (round 3's sweep)
    SCHEDULE["keep_inlets"] = False
    rep["prediction_if_recomputation_exchanged_again"] = []
    for N, pr, pc in ((2, 1, 2), (4, 2, 2), (8, 2, 4)):
        t = sweep_ms(N, pr, pc, (2048, 1024), 8760, 4384, 1104, False, p)
        rep["prediction_if_recomputation_exchanged_again"].append({"n_gpus": N, "pipe_steps": 1104, "sweep_ms": round(t, 1), "efficiency_vs_solo_tile": round(solo / t, 3)})
    SCHEDULE["keep_inlets"] = True
    txt = json.dumps(rep, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
