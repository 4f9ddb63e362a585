#!/bin/bash
# Round-2 experiment 1 (GPU box): how the VALU issues mixed streams, the in-kernel clock, A/B of experiment builds.
# (Taken on the round-1 code, which was the default then; to repeat it on that code: VARIANT=r1_stamp for stamp.py,
# CURL_HIP_LIB=curl_amd/lib/variants/libcurlhip_r1.so for valu_only.py and bench.py.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp1
mkdir -p $OUT
cd $R
python3 -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
timeout -k 10 400 ./tools/ubench/issue_mix 8 4 2 1 > $OUT/issue_mix.log 2>&1 &&
timeout -k 10 300 python3 tools/stamp.py > $OUT/stamp.log 2>&1 &&
timeout -k 10 400 python3 tools/valu_only.py layer lab_stage > $OUT/valu_only_vs_full.log 2>&1 &&
timeout -k 10 300 python3 tools/ab.py curl_amd/lib/variants/libcurlhip_base.so curl_amd/lib/variants/libcurlhip_nofence.so layer > $OUT/ab_base_vs_nofence.log 2>&1 &&
timeout -k 10 300 python3 tools/ab.py curl_amd/lib/variants/libcurlhip_base.so curl_amd/lib/variants/libcurlhip_nodep.so layer > $OUT/ab_base_vs_nodep.log 2>&1 &&
(cd /tmp && export TMPDIR=/tmp &&
 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_layer -- python3 $R/bench.py --workload layer --steps 200 --warmup 50 --no-extras > $OUT/pmc_layer.json 2> $OUT/pmc_layer.err &&
 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mix -- $R/tools/ubench/issue_mix 8 > $OUT/pmc_mix.log 2> $OUT/pmc_mix.err)
echo "exit $?" > $OUT/done.txt
cd /tmp && timeout 60 rocprofv3 -L > $OUT/counters.txt 2>&1
ls $OUT
