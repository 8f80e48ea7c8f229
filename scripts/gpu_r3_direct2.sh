#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for t in 128 192 256 384 512 768; do
echo "== all 512 levels tseg $t"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 0 512 "{\"iter_direct\":1,\"tseg\":$t}" 2>&1 | grep "binomial x4\|diffusion x6" | cut -c1-90 || exit 1
done
for t in 64 96 128 192 256 384; do
echo "== levels 256..319 tseg $t"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 256 64 "{\"iter_direct\":1,\"tseg\":$t}" 2>&1 | grep "binomial x4\|diffusion x6" | cut -c1-90 || exit 1
done
for t in 128 256 512; do
echo "== 1920x1080x256 all levels tseg $t"
timeout -k 10 300 python -u scripts/bench_iter.py 1920 1080 256 0 256 "{\"iter_direct\":1,\"tseg\":$t}" 2>&1 | grep "binomial x4\|diffusion x6" | cut -c1-90 || exit 1
done
echo "== 1920x1080x256 all levels, not direct"
timeout -k 10 300 python -u scripts/bench_iter.py 1920 1080 256 0 256 "{}" 2>&1 | grep "binomial x4\|diffusion x6" | cut -c1-90
