#!/usr/bin/env bash
# round-4 GPU session 8: rank rehearsals on one GPU (store-all and chunked: recomputation without exchange), solo tile at three
# sub-chunk lengths (pipeline model inputs), parts of the real river network alone, parity tables of both builds
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/s8_$name.json 2> gpurun_out/s8_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/s8_$name.err; return 0; }
  python3 - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/s8_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("n_chunks"), d["config"].get("pipe_steps"), d.get("kernel_ms_per_step"), (d.get("tile_solo") or {}).get("ms_per_step"), d.get("efficiency_vs_solo_tile"), d.get("rccl"))
PY
}
C="--steps 3 --warmup 1 --no-cpu-baseline"
run solo_p4384 --of 8 --as-rank 0 --pipe 4384 $C
run solo_p2192 --of 8 --as-rank 0 --pipe 2192 $C
run solo_p1104 --of 8 --as-rank 0 --pipe 1104 $C
run reh2 --gpus 2 --tile-rows 1024 --tile-cols 512 $C --no-tile-solo
run reh4 --gpus 4 --tile-rows 512 --tile-cols 512 $C --no-tile-solo
run reh6 --gpus 6 --tile-rows 1024 --tile-cols 176 --pipe 1104 $C --no-tile-solo
run reh2_chunked --gpus 2 --tile-rows 1024 --tile-cols 512 --chunk 2192 --pipe 1104 $C
run reh4_chunked --gpus 4 --tile-rows 512 --tile-cols 512 --chunk 2192 --pipe 1104 $C --no-tile-solo
for R in 0 3 7; do
  run fr_sub_r$R --mesh france:all --of 8 --as-rank $R --partition sub $C
  run fr_trunk_r$R --mesh france:all --of 8 --as-rank $R --partition trunk $C
done
python3 tools/parity_table.py > gpurun_out/r4_parity_default.md 2> gpurun_out/s8_pt_default.log; echo "parity default rc=$?"
SMASHX_EXACT_LIBM=1 python3 tools/parity_table.py --assert-exact > gpurun_out/r4_parity_exact.md 2> gpurun_out/s8_pt_exact.log; echo "parity exact rc=$?"
tail -3 gpurun_out/r4_parity_default.md | cut -c1-300; tail -3 gpurun_out/r4_parity_exact.md | cut -c1-300
