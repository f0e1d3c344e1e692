#!/bin/bash
# 4-bit bulk (+ hand-over) against 8-bit bulk on shapes with increasing repeat density
for rep in 0.05 0.1 0.2 0.3; do
  for bits in 4 8; do
    echo "repeats_per_kb=$rep bits=$bits" >> gpurun_out/r3_width.log
    ABLATE_REPEATS=$rep FRISK_K8_BITS=$bits python tools/ablate.py FRISK_TUNE 2>&1 | grep -v "warning\|^\s*[0-9]* |\|\^\|In file\|generated" >> gpurun_out/r3_width.log
  done
done
cat gpurun_out/r3_width.log
