#!/bin/bash
# Profiles of the bench command on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats over `bench.py --no-c2 --no-cpu-baseline --no-parity` (same kernel population
#      as the roofline pass inside bench.py), raw output under gpurun_out/prof_bench
#   2. --pmc passes over a short eager run (1 step of 4 DDIM steps), one counter group per pass and only on
#      this library's kernels (--kernel-include-regex): HBM traffic (FETCH_SIZE and WRITE_SIZE do not fit one
#      pass) of the GEMM and of the normalisation kernels, and the SQ busy/issue counters of the GEMM and
#      attention kernels.
# Summaries are written by tools/profile_summary.py into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- \
  python3 $R/bench.py --no-c2 --no-cpu-baseline --no-parity > $R/gpurun_out/prof_bench.log 2>&1
SHORT="--no-c2 --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 0 --ddim-steps 4"
pass() {   # name regex counters...
  local name=$1 regex=$2; shift 2
  timeout -s KILL 200 rocprofv3 --output-format csv --kernel-include-regex "$regex" --pmc "$@" \
    -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py $SHORT > $R/gpurun_out/pmc_$name.log 2>&1
}
# the launch labels of the same short run, in launch order (profile_summary.py splits the dominant kernel's
# dispatches into 3x3 and pointwise launches by joining on the order)
export SDMI_BENCH_LABELS=$R/gpurun_out/pmc_labels.csv
timeout -k 10 200 python3 $R/bench.py $SHORT > $R/gpurun_out/pmc_labels.log 2>&1
unset SDMI_BENCH_LABELS
pass fetch "conv_gemm" FETCH_SIZE
pass write "conv_gemm" WRITE_SIZE
pass sq "conv_gemm|attn_flash|xattn_resident" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass norm_fetch "gn_|layernorm" FETCH_SIZE
pass norm_write "gn_|layernorm" WRITE_SIZE GRBM_GUI_ACTIVE
grep '"metric"' $R/gpurun_out/prof_bench.log | tail -n 1 | cut -c1-400
