#!/usr/bin/env bash
# Build the UNMODIFIED reference solver (Fortran 90 + adStack.c + lbfgsb.f) from the sources where
# they lie under /root/reference into oracle/_ref/ (git-ignored), plus our own bind(C) driver
# (oracle/ref/ref_capi.f90) -> oracle/_ref/libsmash_ref.so.
#
# Test infrastructure only.  No reference source is copied: the compiler reads the files in place.
# Compile order = the reference's makefile.dep:11-41.  Flags per SURVEY Appendix D:
#   -O2 -ffp-contract=off  (bit-identical to -O0; no FMA contraction) = the parity oracle build.
# A second build with the reference's own optimisation level (-O3 -march=native, makefile:6) is
# produced as libsmash_ref_fast.so and is only used as the timed CPU baseline ("reference" kind).
set -euo pipefail
REF=${SMASH_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../_ref"
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
CC=${CC:-gcc}
S="$REF/smash/solver"
if [ ! -d "$S" ]; then echo "build_ref: $S not present, skipping (prebuilt oracle/_ref is used)"; exit 0; fi

ORDER="global/md_constant derived_type/mwd_setup derived_type/mwd_mesh derived_type/mwd_input_data
derived_type/mwd_states derived_type/mwd_output derived_type/mwd_parameters forward/mw_forward
operator/md_gr_operator operator/md_routing_operator operator/md_vic_operator forward/md_forward_structure
routine/m_array_creation routine/m_array_manipulation routine/m_sort routine/m_statistic
routine/mw_derived_type_copy routine/mw_derived_type_update routine/mw_mask routine/mw_sparse_storage
routine/mw_interception_store routine/mw_forcing_statistic routine/mwd_parameters_manipulation
routine/mwd_states_manipulation optimize/mwd_cost routine/mw_multiple_run optimize/mw_optimize
optimize/mw_adjoint_test forward/forward_db forward/forward"

build_variant () {  # $1 = subdir, $2 = lib name, rest = flags
  local sub=$1 lib=$2; shift 2
  local obj="$OUT/$sub"; mkdir -p "$obj"
  $CC -O2 -fPIC -c "$S/tapenade/adStack.c" -o "$obj/adStack.o"
  $FC "$@" -fPIC -c "$S/optimize/lbfgsb.f" -o "$obj/lbfgsb.o"
  for f in $ORDER; do
    b=$(basename "$f")
    $FC -cpp "$@" -fPIC -module-dir "$obj" -I"$obj" -c "$S/$f.f90" -o "$obj/$b.o"
  done
  $FC -cpp "$@" -fPIC -module-dir "$obj" -I"$obj" -c "$HERE/ref_capi.f90" -o "$obj/ref_capi.o"
  $FC -shared -o "$OUT/$lib" "$obj"/*.o
  echo "built $OUT/$lib"
}

build_variant obj_parity libsmash_ref.so      -O2 -ffp-contract=off

# Drop-in library: the reference with base_forward / base_forward_b replaced by our ISO_C_BINDING shim
# (fortran/smashx_dropin.f90) calling libsmashx.  The reference's own definitions are weakened in COPIES of
# its objects, so everything else (mw_forward, mw_optimize's L-BFGS-B loop, base_forward_d, ...) is unchanged.
build_dropin () {
  local src="$OUT/obj_parity" obj="$OUT/obj_dropin" repo
  repo="$(cd "$HERE/../.." && pwd)"
  [ -f "$repo/smash_amd/libsmashx.so" ] || { echo "build_ref: libsmashx.so not built yet, skipping the drop-in"; return 0; }
  rm -rf "$obj"; mkdir -p "$obj"
  cp "$src"/*.o "$src"/*.mod "$obj"/
  local OC=/opt/rocm/lib/llvm/bin/llvm-objcopy
  $OC --weaken-symbol=base_forward_ --weaken-symbol=base_hyper_forward_ "$obj/forward.o"
  $OC --weaken-symbol=base_forward_b_ --weaken-symbol=base_forward_d_ --weaken-symbol=base_hyper_forward_b_ --weaken-symbol=base_hyper_forward_d_ "$obj/forward_db.o"
  $FC -cpp -O2 -ffp-contract=off -fPIC -module-dir "$obj" -I"$obj" -c "$repo/fortran/smashx_dropin.f90" -o "$obj/smashx_dropin.o"
  $FC -shared -o "$OUT/libsmash_dropin.so" "$obj/smashx_dropin.o" $(ls "$obj"/*.o | grep -v smashx_dropin.o) \
      -L"$repo/smash_amd" -lsmashx -Wl,-rpath,'$ORIGIN/../../smash_amd'
  echo "built $OUT/libsmash_dropin.so"
}
build_dropin

# The same two libraries with `setulb` of lbfgsb.f replaced by fortran/smashx_setulb.f90 (the library's own L-BFGS-B behind the
# reference's argument list): the reference's setulb_ is weakened in a copy of lbfgsb.o, the shim's definition takes over, and
# mw_optimize's loop runs unchanged.  libsmash_ref_lbfgsb.so = all-CPU reference + that optimiser (CPU tests: the golden cost
# trajectory), libsmash_dropin_lbfgsb.so = GPU sweeps + that optimiser (the whole calibration on the new library).
build_native_lbfgsb () {
  local repo OC=/opt/rocm/lib/llvm/bin/llvm-objcopy
  repo="$(cd "$HERE/../.." && pwd)"
  [ -f "$repo/smash_amd/libsmashx.so" ] || { echo "build_ref: libsmashx.so not built yet, skipping the setulb variants"; return 0; }
  for pair in "obj_parity:obj_parity_lb:libsmash_ref_lbfgsb.so" "obj_dropin:obj_dropin_lb:libsmash_dropin_lbfgsb.so"; do
    IFS=: read -r src dst lib <<< "$pair"
    [ -d "$OUT/$src" ] || continue
    rm -rf "$OUT/$dst"; mkdir -p "$OUT/$dst"
    cp "$OUT/$src"/*.o "$OUT/$src"/*.mod "$OUT/$dst"/
    $OC --weaken-symbol=setulb_ "$OUT/$dst/lbfgsb.o"
    $FC -cpp -O2 -ffp-contract=off -fPIC -c "$repo/fortran/smashx_setulb.f90" -o "$OUT/$dst/smashx_setulb.o"
    $FC -shared -o "$OUT/$lib" "$OUT/$dst/smashx_setulb.o" $(ls "$OUT/$dst"/*.o | grep -v smashx_setulb.o) \
        -L"$repo/smash_amd" -lsmashx -Wl,-rpath,'$ORIGIN/../../smash_amd'
    echo "built $OUT/$lib"
  done
}
build_native_lbfgsb
if [ "${REF_FAST:-1}" = "1" ]; then
  build_variant obj_fast   libsmash_ref_fast.so -O3 -march=x86-64-v3 -funroll-loops
fi
