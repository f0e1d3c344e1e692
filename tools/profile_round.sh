#!/bin/bash
# Profiling recipe for profiles/ (run on the GPU box from the repo root): kernel trace + stats, then the PMC passes
# in separate runs (never combined with tracing domains; FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# Usage: bash tools/profile_round.sh <tag> [skip_trace]
set -e
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --cpu-windows 0"
if [ -z "$2" ]; then
  timeout -k 10 400 python bench.py > $out/bench_line.json 2> $out/bench.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $out/bench_line_under_rocprof.json 2> $out/trace.err
fi
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > /dev/null 2> $out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > /dev/null 2> $out/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- $B > /dev/null 2> $out/pmc_sq.err
python tools/pmc_summary.py $out > $out/pmc_summary.json
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
ls $out
