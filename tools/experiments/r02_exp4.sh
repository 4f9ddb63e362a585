#!/bin/bash
# Round-2 experiment 4 (GPU box): the new default (scalar helpers + transcendental runs at raised priority) --
# full GPU test suite, bench line, 2-rank rehearsal, A/B against the round-1 code, in-kernel clock.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp4
mkdir -p $OUT
cd $R
V=curl_amd/lib/variants
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout -k 10 400 python3 bench.py > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err &&
timeout -k 10 300 python3 bench.py --gpus 2 --steps 200 --warmup 20 --no-extras > $OUT/bench_2rank_rehearsal.json 2> $OUT/bench_2rank.err
for pair in "r1 base layer" "r1 base lab_stage" "r1 base trispace" "base nopk_t3 layer" "base nofence layer"; do
  set -- $pair
  timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_$1.so $V/libcurlhip_$2.so $3 > $OUT/ab_$1_vs_$2_$3.log 2>&1 || exit 1
done
FLAGS_B=0x200 timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_base.so $V/libcurlhip_base.so layer > $OUT/ab_base_U1_vs_U2_layer.log 2>&1
timeout -k 10 300 python3 tools/stamp.py layer layer_nomem lab_stage rgb_only > $OUT/stamp_base.log 2>&1 &&
VARIANT=r1_stamp timeout -k 10 300 python3 tools/stamp.py layer layer_nomem > $OUT/stamp_r1.log 2>&1
echo "exit $?" > $OUT/done.txt
