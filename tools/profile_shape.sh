#!/bin/bash
# Counters of the scan on ONE shard of the C5 shape with repeat content (the side-table form real assemblies take): kernel trace +
# stats, then FETCH_SIZE / WRITE_SIZE / one SQ set in SEPARATE --pmc passes (GPU box, repo root).
# Usage: bash tools/profile_shape.sh <tag> [repeats per kb, default 0.35]
tag=${1:-shape}
rep=${2:-0.35}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --workload shard --repeats $rep --steps 2 --warmup 1 --cpu-windows 0 --no-upload --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/bench_line_under_rocprof.json 2> $out/trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > /dev/null 2> $out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > /dev/null 2> $out/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- $B > /dev/null 2> $out/pmc_sq.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq2 -- $B > /dev/null 2> $out/pmc_sq2.err
python3 tools/pmc_summary.py $out > $out/pmc_summary.json
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
ls $out
