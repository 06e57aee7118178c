#!/bin/bash
# Build the C-ABI library in-tree; exits non-zero (and says so) when any compile or the link fails, so a stale .so is never mistaken for a fresh one.
cd "$(dirname "$0")/.." || exit 1
out=$(python -c "import layoutdetr_amd.build as b; b.build()" 2>&1); rc=$?
echo "$out" | tail -${1:-3}
if [ $rc -ne 0 ]; then echo "BUILD FAILED rc=$rc"; exit $rc; fi
echo "build ok: $(ls -la --time-style=+%H:%M:%S layoutdetr_amd/lib/libldetr_hip.so | awk '{print $6, $7}')"
