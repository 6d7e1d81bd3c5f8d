#!/bin/bash
# One gpurun call: the GPU test suite, then (unless the tests were killed by their time limit) the default bench.
# usage: scripts/gpu_round.sh <tag> [pytest args...]
TAG=${1:-r3}; shift
mkdir -p gpurun_out
timeout -k 10 ${TEST_LIMIT:-800} python -m pytest tests -m gpu -q "$@" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -n 15 gpurun_out/${TAG}_tests.log
echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed at their limit: no further GPU step"; exit $rc; fi
timeout -k 10 ${BENCH_LIMIT:-500} python bench.py ${BENCH_ARGS} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
brc=$?
echo "bench rc=$brc"; tail -c 3000 gpurun_out/${TAG}_bench.json; tail -n 5 gpurun_out/${TAG}_bench.err
[ $rc -eq 0 ] && [ $brc -eq 0 ]
