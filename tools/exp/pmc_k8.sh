#!/bin/bash
# usage: pmc_k8.sh <tag> <lib.so> <bits>   (GPU box, repo root) - kernel trace + SQ counters of the scan kernel on the bench shard
tag=$1; lib=$2; bits=$3
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp FRISK_HIP_LIB=$lib FRISK_K8_BITS=$bits
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/exp/k8_one.py 1.0 3 > $out/trace.out 2> $out/trace.err
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- python3 tools/exp/k8_one.py 1.0 2 > $out/sq.out 2> $out/sq.err
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq2 -- python3 tools/exp/k8_one.py 1.0 2 > $out/sq2.out 2> $out/sq2.err
python3 tools/pmc_summary.py $out > $out/pmc_summary.json
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
tail -3 $out/sq2.err
