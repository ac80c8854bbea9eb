#!/bin/bash
# Build an experimental variant of the library OUTSIDE the product tree:
#   bash scripts/experiments/build_variant.sh <name> [-DSERHIP_EXP_K1_PLAIN_STORES=1 ...]
# copies ser_amd/csrc + include to a temp directory, applies scripts/experiments/variants/*.patch (which put the
# SERHIP_EXP_* / SERHIP_K1_* / SERHIP_K2_* switch sites back into the kernel headers) and any extra patch named in
# $EXTRA_PATCHES, and compiles scripts/experiments/ab/<name>.so with the given -D flags.  The shipped headers hold
# no such switches (several variants produce garbage on purpose).
set -euo pipefail
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/ser_amd" "$ROOT/scripts/experiments/ab"
cp -r "$ROOT/ser_amd/csrc" "$TMP/ser_amd/csrc"
cp -r "$ROOT/include" "$TMP/include"
for p in "$ROOT"/scripts/experiments/variants/*.patch ${EXTRA_PATCHES:-}; do
  patch -s -d "$TMP/ser_amd/csrc" -p1 < "$p"
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -shared -fPIC "$@" \
      -o "$ROOT/scripts/experiments/ab/$NAME.so" "$TMP/ser_amd/csrc/serhip.hip"
echo "built scripts/experiments/ab/$NAME.so"
