#!/bin/bash
# Run the GPU test suite in ONE process on the box and keep both streams: gpurun_out/<tag>/pytest.log (stdout) and
# gpurun_out/<tag>/pytest.err (stderr: HIP / ROCr messages of a fault or abort end up here -- round 2 lost them).
# usage: tools/gpu_suite.sh <tag> [--ab] [pytest args...]
#   --ab   build the AB library on the box (make AB=1, ~2 min) and run the `ab`-marked variant matrix against it instead
set -o pipefail
tag=${1:-suite}; shift
out=gpurun_out/$tag
mkdir -p "$out"
sel="gpu"
if [ "$1" = "--ab" ]; then
    shift
    make -C mlvectordb_amd/csrc AB=1 -j8 > "$out/make_ab.log" 2>&1 || { tail -20 "$out/make_ab.log"; exit 1; }
    export MLVDB_HIP_LIBRARY=$PWD/mlvectordb_amd/csrc/libmlvdb_hip_ab.so
    sel="ab"
fi
timeout -k 10 ${GPU_SUITE_TIMEOUT:-900} python -X faulthandler -m pytest tests -m "$sel" ${GPU_SUITE_X--x} -q --durations=60 "$@" > "$out/pytest.log" 2> "$out/pytest.err"
rc=$?
echo "pytest rc=$rc" >> "$out/pytest.log"
tail -5 "$out/pytest.log"
[ -s "$out/pytest.err" ] && { echo "--- stderr (tail)"; tail -20 "$out/pytest.err"; }
exit $rc
