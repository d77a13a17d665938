#!/bin/bash
# AddressSanitizer + UBSan over the host-side sources (no GPU code involved).
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -fno-omit-frame-pointer -o /tmp/rtx_host_sanitize tools/host_sanitize_main.cpp \
    raytracing_amd/csrc/host/scene_parser.cpp raytracing_amd/csrc/host/image_io.cpp raytracing_amd/csrc/host/checkpoint.cpp
/tmp/rtx_host_sanitize
