#!/bin/bash
# (GPU box) parking-store cache policy A/B: scan times of prebuilt variants (build/ab/{base,sc1,nt}.so) and their FETCH_SIZE / WRITE_SIZE
# on the bench shard, separate --pmc passes.  usage: bash tools/exp/park_policy.sh <outdir>
out=${1:-gpurun_out/park}
mkdir -p $out
export TMPDIR=/tmp
python3 tools/exp/ab_run.py 2 build/ab/base.so build/ab/sc1.so build/ab/nt.so > $out/ab.log 2>&1
B="python3 bench.py --workload shard --steps 2 --warmup 1 --cpu-windows 0 --no-upload --no-extra"
for v in base sc1 nt; do
  export FRISK_HIP_LIB=$PWD/build/ab/$v.so
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$v/pmc_fetch -- $B > /dev/null 2> $out/$v.fetch.err
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$v/pmc_write -- $B > /dev/null 2> $out/$v.write.err
  python3 tools/pmc_summary.py $out/$v > $out/$v.pmc.json
done
cat $out/ab.log
