#!/bin/bash
# in-situ FETCH_SIZE (KiB per launch as the counter reports it; x2 on gfx950 for bytes: MI355X_MICROARCH.md) of the C2 step's kernels with the
# 144-column tiles' old raster (whole columns per XCD) and the 8-row sweeps
for r in 32 8; do
  echo "=== RGM_G144_RASTER=$r"
  RGM_G144_RASTER=$r bash tools/pmc_kernel_insitu.sh FETCH_SIZE --no-extras --no-traffic 2>&1 | grep -i "gemm144\|gemm2_kernel<256"
done
