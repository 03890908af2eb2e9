#!/usr/bin/env python3
"""A/B matrix on ONE box (box-to-box spread is +-3 %): every configuration is one `python3 bench.py --profile <args>` child process.

    tools/ab_matrix.py [--tag T] [--timeout S] [common bench args] -- "label|lib|ENV=a ENV2=b|extra bench args" ...

lib: "base" = smash_amd/libsmashx.so, "exact" = libsmashx_exact.so, anything else = variants/lib_<lib>.so.  Writes
gpurun_out/<T>_<label>.json (the bench line) and prints one compact row per configuration as it completes (a long matrix keeps
talking).  A configuration that fails or times out is reported and the matrix goes on with the next one -- unless the failure
was a timeout (a GPU step killed at its limit ends the call: no further GPU step)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    argv = sys.argv[1:]
    tag, tmo = "abm", 420
    while argv and argv[0] in ("--tag", "--timeout"):
        if argv[0] == "--tag":
            tag = argv[1]
        else:
            tmo = int(argv[1])
        argv = argv[2:]
    cut = argv.index("--")
    common, specs = argv[:cut], argv[cut + 1:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rows = []
    for spec in specs:
        parts = (spec.split("|") + ["", "", ""])[:4]
        label, lib, envs, extra = (x.strip() for x in parts)
        path = {"base": "smash_amd/libsmashx.so", "": "smash_amd/libsmashx.so", "exact": "smash_amd/libsmashx_exact.so"}.get(lib, f"variants/lib_{lib}.so")
        env = dict(os.environ, SMASHX_LIB=os.path.join(ROOT, path))
        for kv in envs.split():
            k, v = kv.split("=", 1)
            env[k] = v
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--profile"] + common + extra.split()
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=tmo, cwd=ROOT)
        except subprocess.TimeoutExpired:
            print(f"{label}: TIMEOUT after {tmo} s -- stopping the matrix", flush=True)
            break
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            print(f"{label}: FAILED rc={r.returncode}: {r.stderr.strip().splitlines()[-3:]}", flush=True)
            continue
        d = json.loads(lines[-1])
        with open(os.path.join(ROOT, "gpurun_out", f"{tag}_{label}.json"), "w") as f:
            f.write(lines[-1] + "\n")
        k = d.get("kernel_ms_per_step", {})
        row = {"label": label, "lib": lib or "base", "env": envs, "args": extra, "ms": round(d["ms_per_step"], 2),
               "kernels": {a.replace("sx_k_", ""): round(b, 1) for a, b in k.items()}, "cost": d.get("cost"),
               "chunks": d["config"].get("n_chunks"), "pipe": d["config"].get("pipe_steps"), "rounds": d["config"].get("routing_rounds"),
               "wall_s": round(time.time() - t0, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_matrix.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
