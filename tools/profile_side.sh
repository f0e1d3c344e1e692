#!/bin/bash
# The side-table form under the profiler (GPU box, repo root): bench.py on the `realistic` unmasked shape (--repeats 0.35: the sample picks
# 4-bit counters + side table) - kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the two SQ sets in separate --pmc passes, as
# tools/profile_round.sh.  Usage: bash tools/profile_side.sh <tag>
tag=${1:-side}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-upload --no-extra --repeats 0.35"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-upload --no-extra --repeats 0.35 > $out/bench_line_under_rocprof.json 2> $out/trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > /dev/null 2> $out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > /dev/null 2> $out/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- $B > /dev/null 2> $out/pmc_sq.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq2 -- $B > /dev/null 2> $out/pmc_sq2.err
python3 tools/pmc_summary.py $out > $out/pmc_summary.json
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/trace -name "*kernel_trace.csv" -exec cp {} $out/kernel_trace.csv \;
python3 tools/pmc_bench_json.py $out > $out/pmc_bench.json
ls $out
