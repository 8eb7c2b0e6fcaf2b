#!/bin/bash
for shape in 1 2; do
for dbg in 0 4 8 12; do
  echo "== SDMI_GEMM_SHAPE=$shape SDMI_GEMM_DEBUG_ZERO=$dbg"
  SDMI_GEMM_SHAPE=$shape SDMI_GEMM_DEBUG_ZERO=$dbg timeout -k 10 120 python -u tools/bench_shapes.py gemm16 2>&1 | grep -E "conv3x3 320->320 @64|conv3x3 640->640 @32|1280->1280 @16|ff1 320|sum"
done
done
