#!/bin/bash
# Debug build with s_memtime stage stamps: cooperative-search_amd/csrc/libcoopsearch_tl.so (use with COOPSEARCH_LIB=...)
cd "$(dirname "$0")/../cooperative-search_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
  -DCS_TIMELINE -DCS_ONLY_N=${1:-3} -I ../../include coopsearch.hip policy.hip episodes.hip -o libcoopsearch_tl.so && echo built libcoopsearch_tl.so
