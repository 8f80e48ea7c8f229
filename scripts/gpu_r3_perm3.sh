#!/bin/bash
# binomial x4 on smaller resident volumes: LDS form (index 0 / 9: default, multi_px 0) against pixels on lanes (4: multi_px 32)
set -o pipefail
for shape in "1920 1080 256" "1920 1080 64" "1920 1080 16" "1280 720 128" "3840 2160 64" "640 480 64"; do
echo "== $shape"
timeout -k 10 200 python -u scripts/bench_binom.py $shape 0,4 2>&1 | grep -v amdgpu.ids || exit 1
done
