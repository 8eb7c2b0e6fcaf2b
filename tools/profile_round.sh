#!/bin/bash
# Profiles of the bench command on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats over `bench.py --no-c2 --no-cpu-baseline` (same kernel population
#      as the roofline pass inside bench.py), raw output under gpurun_out/prof_bench
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over a 4-step eager run
# Summaries are written by tools/profile_summary.py into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- \
  python3 $R/bench.py --no-c2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
SHORT="--no-c2 --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 0 --ddim-steps 4"
timeout -s KILL 200 rocprofv3 --output-format csv --kernel-include-regex "conv_gemm" --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -- \
  python3 $R/bench.py $SHORT > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --kernel-include-regex "conv_gemm" --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -- \
  python3 $R/bench.py $SHORT > $R/gpurun_out/pmc_write.log 2>&1
grep '"metric"' $R/gpurun_out/prof_bench.log | tail -n 1 | cut -c1-400
