#!/bin/bash
# run on the GPU box: tests, bench as the driver runs it (stdout = the contract line only), the multi-rank rehearsal's line
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py "$@" > gpurun_out/bench.stdout 2> gpurun_out/bench.stderr || { tail -30 gpurun_out/bench.stderr; exit 1; }
wc -c gpurun_out/bench.stdout
cat gpurun_out/bench.stdout
