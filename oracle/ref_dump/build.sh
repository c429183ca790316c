#!/bin/bash
# Tries to build oracle/_ref/ref_dump from the real reference sources; records the outcome in oracle/_ref/STATUS.
# (In the round-2 image: "unavailable" -- Eigen3 and ifopt are not installed and there is no network.)
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
mkdir -p "$OUT/build"
REF=${TOWR_REFERENCE_DIR:-/root/reference/towr}
if [ ! -d "$REF/src" ]; then
  echo "unavailable: reference sources not present at $REF" > "$OUT/STATUS"
elif cmake -S "$HERE" -B "$OUT/build" -DTOWR_REFERENCE_DIR="$REF" -DTOWR_AMD_ROOT="$HERE/../.." -DCMAKE_BUILD_TYPE=Release > "$OUT/configure.log" 2>&1 \
     && cmake --build "$OUT/build" -j4 > "$OUT/build.log" 2>&1; then
  cp "$OUT/build/ref_dump" "$OUT/ref_dump"
  echo "available" > "$OUT/STATUS"
else
  echo "unavailable: $(grep -m1 -E 'Could not find|Could NOT find|error' "$OUT/configure.log" "$OUT/build.log" 2>/dev/null | head -1 | cut -c1-200)" > "$OUT/STATUS"
fi
cat "$OUT/STATUS"
