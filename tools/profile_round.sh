#!/bin/bash
# Profiling recipe for profiles/ (GPU box, repo root).  Kernel trace + stats, a kernel + memory-copy trace of the staged
# upload, then the PMC passes in SEPARATE runs (never combined with tracing domains; FETCH_SIZE and WRITE_SIZE do not fit
# one pass on gfx950).  Usage: bash tools/profile_round.sh <tag>
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-upload --no-extra"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-upload --no-extra > $out/bench_line_under_rocprof.json 2> $out/trace.err
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/upload_trace -- python3 tools/stream_job.py 3 > $out/stream_job.json 2> $out/upload_trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > /dev/null 2> $out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > /dev/null 2> $out/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- $B > /dev/null 2> $out/pmc_sq.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq2 -- $B > /dev/null 2> $out/pmc_sq2.err
python3 tools/pmc_summary.py $out > $out/pmc_summary.json
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/trace -name "*kernel_trace.csv" -exec cp {} $out/kernel_trace.csv \;
find $out/upload_trace -name "*kernel_trace.csv" -exec cp {} $out/upload_kernel_trace.csv \;
find $out/upload_trace -name "*memory_copy_trace.csv" -exec cp {} $out/upload_memory_copy_trace.csv \;
python3 tools/pmc_bench_json.py $out > $out/pmc_bench.json
ls $out
