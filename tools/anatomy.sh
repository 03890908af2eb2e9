#!/usr/bin/env bash
# Timing-only builds of libsmashx (results void, timings valid): what each ingredient of the kernels costs.
#   here (container):   tools/anatomy.sh build                  -> variants/lib_<name>.so  (git-ignored, travels with gpurun)
#   on the GPU box:     tools/anatomy.sh run [names...]          -> gpurun_out/anatomy_<name>.json (+ per-round routing traces)
# Switches: the timing-only ones (SX_ABL_*: one ingredient of a kernel removed, results void) and SX_WAVE_BRANCH are NOT in the product
# sources (round 4): `build` compiles from a scratch copy of smash_amd/csrc with tools/anatomy/timing_switches.patch applied (the patch
# re-adds them to sx_kernels.h, sx_math.h and sx_ops.h; regenerate it with `diff -u` if those files move under it).
# Numbers quoted in DESIGN.md 12 and profiles/r2_routing_anatomy*.json come from this.  The round-3 switches (still0, tanh0, wb15, onesub,
# exact_*: valid results, different instruction streams) are compared on one box with tools/ab_variants.sh.
set -u
cd "$(dirname "$0")/.."
declare -A V=(
  [base]="" [tanh]="-DSX_ABL_TANH" [div]="-DSX_ABL_DIV" [pow]="-DSX_ABL_POW" [arith]="-DSX_ABL_TANH -DSX_ABL_DIV -DSX_ABL_POW"
  [nobar]="-DSX_ABL_NOBAR" [norel]="-DSX_ABL_NOREL" [nolds]="-DSX_ABL_NOLDS" [nost]="-DSX_ABL_NOST=1" [nold]="-DSX_ABL_NOLD=1"
  [alu]="-DSX_ABL_NOLDS -DSX_ABL_NOST=1 -DSX_ABL_NOLD=1 -DSX_ABL_NOBAR"
  [still0]="-DSX_STILL=0" [tanh0]="-DSX_TANH_FAST=0" [wb15]="-DSX_WAVE_BRANCH=15" [onesub]="-DSX_ABL_ONE_SUBLEVEL"
  [exact]="-DSX_EXACT_LIBM=1" [exact_d0]="-DSX_EXACT_LIBM=1 -DSX_EXACT_DIV=0" [exact_d1]="-DSX_EXACT_LIBM=1 -DSX_EXACT_DIV=1" [exact_noguard]="-DSX_EXACT_LIBM=1 -DSX_ABL_NOGUARD"
  [anoq]="-DSX_ABL_A_NOQ=1" [anox]="-DSX_ABL_A_NOX=1" [anohr]="-DSX_ABL_A_NOHR=1" [pk4]="-DSX_PK=4" [pk64]="-DSX_PK=64" [mu2]="-DSX_MU=2" [mu8]="-DSX_MU=8"
)
case "${1:-}" in
build)
  mkdir -p variants
  F="-O3 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -std=c++17"
  W=$(mktemp -d); cp -r smash_amd include "$W"/
  (cd "$W" && patch -p1 -s < "$OLDPWD/tools/anatomy/timing_switches.patch") || { echo "timing_switches.patch does not apply"; exit 1; }
  n=0
  for k in "${!V[@]}"; do
    /opt/rocm/bin/hipcc $F ${V[$k]} -o variants/lib_$k.so "$W"/smash_amd/csrc/smashx.hip "$W"/smash_amd/csrc/sx_plan.cpp "$W"/smash_amd/csrc/sx_lbfgsb.cpp "$W"/smash_amd/csrc/sx_hyper.cpp -pthread -ldl 2>/dev/null &
    n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
  done
  wait; ls variants/*.so ;;
run)
  shift
  names=("$@"); [ ${#names[@]} -eq 0 ] && names=("${!V[@]}")
  mkdir -p gpurun_out
  for k in "${names[@]}"; do
    [ -f variants/lib_$k.so ] || { echo "variants/lib_$k.so missing: run 'tools/anatomy.sh build' first"; continue; }
    SMASHX_TRACE_GROUPS=1 SMASHX_LIB=$PWD/variants/lib_$k.so timeout -k 10 300 python3 bench.py --no-secondary --no-exact --no-cpu-baseline \
        --trace-groups gpurun_out/anatomy_trace_$k.json > gpurun_out/anatomy_$k.log 2>&1 || { echo "$k failed"; continue; }
    tail -1 gpurun_out/anatomy_$k.log > gpurun_out/anatomy_$k.json
    python3 - "$k" <<'PY'
import json, sys
k = sys.argv[1]
d = json.load(open(f"gpurun_out/anatomy_{k}.json")); t = json.load(open(f"gpurun_out/anatomy_trace_{k}.json"))
out = [k, round(d["ms_per_step"], 2), d["kernel_ms_per_step"]]
for ps in ("forward", "adjoint"):
    f = t[ps]["rounds"]; r0 = [r for r in f if r["round"] == 0][0]; ch = [r for r in f if r["round"] > 0]
    out.append({ps: {"round0_ms": round(r0["last_end_ms"] - r0["first_start_ms"], 2),
                     "chained_ms": round(max(r["last_end_ms"] for r in ch) - min(r["first_start_ms"] for r in ch), 2) if ch else 0.0}})
print(*out)
PY
  done ;;
*) sed -n 2,7p "$0" ;;
esac
