#!/bin/bash
# round 3: hypothesis searches of tests/test_gpu_properties.py at 10x the examples (a soak, once), then the plain run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 -m pytest tests/test_gpu_properties.py -q -m gpu > $O/hyp_plain.log 2>&1; tail -2 $O/hyp_plain.log
CURL_HYP_SCALE=10 python3 -m pytest tests/test_gpu_properties.py -q -m gpu -p no:cacheprovider > $O/hypothesis_soak.log 2>&1; tail -3 $O/hypothesis_soak.log
