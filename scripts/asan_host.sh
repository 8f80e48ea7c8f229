#!/bin/bash
# Host-side AddressSanitizer / UBSan run of the C++ driver (host code only: the HIP library and the GPU
# code are not instrumented).  Builds an instrumented StereoMatch here (CPU) into build/asan/ (git-ignored, travels with gpurun) and,
# on the GPU box, runs the driver tests against it:
#   bash scripts/asan_host.sh build          (here)
#   bash scripts/asan_host.sh run            (GPU box)
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/asan
case "${1:-build}" in
build)
    mkdir -p "$out"
    cd "$root/stereomatch_amd/csrc/host"
    g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
        ImageIO.cpp ParameterIO.cpp StereoParameters.cpp StereoMatcher.cpp StcEvaluate.cpp StereoIO.cpp main.cpp \
        -L"$root/stereomatch_amd/lib" -lsmhip -Wl,-rpath,"$root/stereomatch_amd/lib" -Wl,-rpath-link,/opt/rocm/lib \
        -o "$out/StereoMatch"
    ls -la "$out/StereoMatch"
    ;;
run)
    export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:exitcode=66
    export SMHIP_DRIVER_BIN="$out/StereoMatch"
    cd "$root"
    timeout -k 10 900 python -u -m pytest tests/test_gpu_host_driver.py -x -q --timeout 300 --timeout-method thread -m gpu
    ;;
esac
