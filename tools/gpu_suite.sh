#!/bin/bash
# Run the GPU test suite in ONE process on the box and keep both streams: gpurun_out/<tag>/pytest.log (stdout) and
# gpurun_out/<tag>/pytest.err (stderr: HIP / ROCr messages of a fault or abort end up here -- round 2 lost them).
# usage: tools/gpu_suite.sh <tag> [pytest args...]
set -o pipefail
tag=${1:-suite}; shift
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 ${GPU_SUITE_TIMEOUT:-900} python -X faulthandler -m pytest tests -m gpu -x -q "$@" > "$out/pytest.log" 2> "$out/pytest.err"
rc=$?
echo "pytest rc=$rc" >> "$out/pytest.log"
tail -5 "$out/pytest.log"
[ -s "$out/pytest.err" ] && { echo "--- stderr (tail)"; tail -20 "$out/pytest.err"; }
exit $rc
