#!/bin/bash
# Round-2 experiment 6 (GPU box): workgroup size of the streaming kernels; backward kernels under the new default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp6
mkdir -p $OUT
cd $R
V=curl_amd/lib/variants
python3 - > $OUT/block_size_check.log 2>&1 <<'PY'
import torch
from curl_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for shape in ((2, 40, 60), (1, 33, 65), (2, 7, 9)):
    B, H, W = shape
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
    ref, _ = ops.curl_layer_forward(img, mask, L, R, Hk)
    for fl in (0x800, 0x1000, 0x1000 | 0x200, 0x800 | 0x400):
        out, _ = ops.curl_layer_forward(img, mask, L, R, Hk, flags=fl)
        print(shape, hex(fl), "bit-exact" if torch.equal(out, ref) else "DIFFERS")
        ls, _ = ops.lab_stage(img, mask, L, flags=fl)
        ls0, _ = ops.lab_stage(img, mask, L)
        assert torch.equal(ls, ls0)
PY
cat $OUT/block_size_check.log
for fl in 0x800 0x1000 0x1200; do
  FLAGS_B=$fl timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_base.so $V/libcurlhip_base.so layer > $OUT/ab_layer_block_$fl.log 2>&1 || exit 1
  tail -4 $OUT/ab_layer_block_$fl.log
done
FLAGS_B=0x1000 timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_base.so $V/libcurlhip_base.so lab_stage > $OUT/ab_lab_block_0x1000.log 2>&1
tail -4 $OUT/ab_lab_block_0x1000.log
H=256 W=256 timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_r1.so $V/libcurlhip_base.so layer_bwd > $OUT/ab_r1_vs_base_layer_bwd_256.log 2>&1
tail -2 $OUT/ab_r1_vs_base_layer_bwd_256.log
B=8 timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_r1.so $V/libcurlhip_base.so layer_bwd > $OUT/ab_r1_vs_base_layer_bwd_full.log 2>&1
tail -2 $OUT/ab_r1_vs_base_layer_bwd_full.log
echo "exit $?" > $OUT/done.txt
