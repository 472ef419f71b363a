#!/bin/bash
# Builds the library as of a git revision (or the working tree: rev "WORK") into recommender_amd/lib/ab/<name>.so for same-box
# A/B runs through RECO_HIP_LIB (tools/ab_probe.py).  Extra arguments are appended to the compiler flags.
#   bash tools/build_variant.sh HEAD base ;  bash tools/build_variant.sh WORK new -DRECO_SOMETHING=1
set -e -o pipefail
REV=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/reco_variant.XXXXXX)
mkdir -p "$TMP/recommender_amd/csrc" "$TMP/include" "$ROOT/recommender_amd/lib/ab"
if [ "$REV" = WORK ]; then
  cp "$ROOT"/recommender_amd/csrc/*.hip "$ROOT"/recommender_amd/csrc/*.h "$ROOT"/recommender_amd/csrc/Makefile "$TMP/recommender_amd/csrc/"
  cp "$ROOT"/include/*.h "$TMP/include/"
else
  git -C "$ROOT" archive "$REV" recommender_amd/csrc include | tar -x -C "$TMP"
fi
make -C "$TMP/recommender_amd/csrc" -j8 EXTRA="$*" > "$TMP/build.log" 2>&1 || { tail -30 "$TMP/build.log"; exit 1; }
cp "$TMP/recommender_amd/lib/libreco_hip.so" "$ROOT/recommender_amd/lib/ab/$NAME.so"
rm -rf "$TMP"
echo "built recommender_amd/lib/ab/$NAME.so from $REV $*"
