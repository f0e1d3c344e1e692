#!/bin/bash
# usage (GPU box): tools/exp/ab.sh OUT "D1,D2" "D3" ...   - one experiment build per argument (ablate.py), whole bench shard, pinned rows
export ABLATE_SCALE=${ABLATE_SCALE:-1.0}
out=$1; shift
for spec in "$@"; do
    python tools/ablate.py "$spec" 2>&1 | grep -v "warning\|^\s*[0-9]* |\|\^\|In file\|generated" >> $out
done
cat $out
