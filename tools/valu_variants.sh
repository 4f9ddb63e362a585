#!/bin/bash
# compare library builds (scheduler strategies): tools/valu_variants.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "=== $lib"
  CURL_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/valu_only.py 2>/dev/null | grep -E "^(layer|lab_stage)"
done
