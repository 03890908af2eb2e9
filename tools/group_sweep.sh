for g in 512 384 256 128; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exact --group $g --trace-groups gpurun_out/grp_$g.json > gpurun_out/grpb_$g.json 2>/dev/null; done
python - <<'PY'
import json
for g in (512,384,256,128):
    d=json.loads(open(f"gpurun_out/grpb_{g}.json").read().strip().splitlines()[-1]); t=json.load(open(f"gpurun_out/grp_{g}.json"))
    f=t["forward"]["rounds"]; a=t["adjoint"]["rounds"]
    print("group",g, round(d["ms_per_step"],2), d["kernel_ms_per_step"], "rounds", d["config"]["routing_rounds"], "groups", d["config"]["routing_groups"], "fwd r0 end", round(f[0]["last_end_ms"],1), "fwd span", round(t["forward"]["span_ms"],1), "adj chain end", round(max(r["last_end_ms"] for r in a[1:]),1), "adj span", round(t["adjoint"]["span_ms"],1))
PY
