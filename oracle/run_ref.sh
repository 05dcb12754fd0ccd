#!/bin/bash
# run_ref.sh -- ONE command that pins the oracle against output of the reference itself:
#   1. probes a Go toolchain (>= 1.18; the reference pins 1.18.3) -- without one prints "SKIP: no go" and exits 0;
#   2. copies the reference tree to oracle/_ref/work/src (never writes to /root/reference), applies
#      integration/go/parity/determinise.patch (first-maximum selectHost, one filter worker, a trace hook in schedulePods) and
#      drops integration/go/parity/parity_test.go into pkg/simulator/;
#   3. `go test -mod=vendor ./pkg/simulator/ -run TestParityDump` over the manifest oracle/compare_ref.py writes (the reference's
#      own example/ inputs + seeded random clusters as YAML): per pod, in submission order, the node it was bound to / the FitError;
#   4. oracle/compare_ref.py compare: the same inputs through the Python mirror + the C oracle with the reference's pod order as
#      data; placements and reasons diffed pod by pod.  Exit code 0 = identical (or SKIP), 1 = differences, 2 = the job broke.
#      SIMON_PARITY_ENGINE=hip|both (GPU box) compares the HIP library's placements / reasons instead of / next to the oracle's.
# Everything it produces stays under oracle/_ref/ (git-ignored).  Optional: REF=/path/to/open-simulator  GO=/path/to/go
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${REF:-/root/reference}
WORK=$HERE/_ref/work
GO_BIN=${GO:-}
if [ -z "$GO_BIN" ]; then
  for cand in go /usr/local/go/bin/go /usr/lib/go/bin/go /usr/lib/go-1.*/bin/go "$HOME/go/bin/go" "$HOME/sdk"/go*/bin/go; do
    if command -v "$cand" >/dev/null 2>&1; then GO_BIN=$(command -v "$cand"); break; fi
  done
fi
if [ -z "$GO_BIN" ]; then
  echo "SKIP: no go (no Go toolchain on this box: the determinised reference cannot be built; parity stays pinned to the oracle only)"
  exit 0
fi
VER=$("$GO_BIN" version 2>/dev/null | sed -E 's/.*go([0-9]+)\.([0-9]+).*/\1 \2/')
set -- $VER
if [ "${1:-0}" -lt 1 ] || { [ "${1:-0}" -eq 1 ] && [ "${2:-0}" -lt 18 ]; }; then
  echo "SKIP: no go (found $("$GO_BIN" version), the reference needs >= 1.18)"
  exit 0
fi
if [ ! -d "$REF/vendor" ] || [ ! -d "$REF/pkg/simulator" ]; then
  echo "SKIP: no reference tree at $REF (set REF=...)"
  exit 0
fi
echo "go: $("$GO_BIN" version)   reference: $REF"
mkdir -p "$WORK" || exit 2
rm -rf "$WORK/src"
cp -r "$REF" "$WORK/src" || exit 2
( cd "$WORK/src" && patch -p1 < "$ROOT/integration/go/parity/determinise.patch" ) || { echo "FAILED: determinise.patch does not apply"; exit 2; }
cp "$ROOT/integration/go/parity/parity_test.go" "$WORK/src/pkg/simulator/zz_parity_test.go" || exit 2
python3 "$HERE/compare_ref.py" cases --ref "$WORK/src" --work "$WORK" || exit 2
( cd "$WORK/src" && CGO_ENABLED=0 GOFLAGS=-mod=vendor SIMON_PARITY_MANIFEST="$WORK/cases.json" SIMON_PARITY_OUT="$WORK/ref_out.json" \
    "$GO_BIN" test ./pkg/simulator/ -run TestParityDump -count=1 -timeout 30m ) > "$WORK/go_test.log" 2>&1
rc=$?
tail -5 "$WORK/go_test.log"
if [ $rc -ne 0 ] || [ ! -s "$WORK/ref_out.json" ]; then echo "FAILED: the determinised reference did not run (see $WORK/go_test.log)"; exit 2; fi
python3 "$HERE/compare_ref.py" compare --work "$WORK" --ref-out "$WORK/ref_out.json" --engine "${SIMON_PARITY_ENGINE:-oracle}"
