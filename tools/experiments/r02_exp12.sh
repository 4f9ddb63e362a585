#!/bin/bash
# per-kernel times of the fused layer's backward (8 x 1500 x 1000 and 32 x 256 x 256)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/layer_bwd_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
V=$R/curl_amd/lib/variants
B=8 ROUNDS=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -- python3 $R/tools/ab.py $V/libcurlhip_base.so $V/libcurlhip_base.so layer_bwd > $OUT/full.log 2>&1
H=256 W=256 ROUNDS=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/crop -- python3 $R/tools/ab.py $V/libcurlhip_base.so $V/libcurlhip_base.so layer_bwd > $OUT/crop.log 2>&1
for d in full crop; do echo "## $d"; f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); grep -i "bwd\|prep" "$f" | cut -d, -f1-4; done
