#!/bin/bash
# Experimental build of the library for ONE team size: build/var/<name>.so (use with COOPSEARCH_LIB=...).
#   usage: tools/build_var.sh <name> <n_agents> [extra -D flags...]
R="$(cd "$(dirname "$0")/.." && pwd)"
NAME="${1:?name}"; N="${2:?n_agents}"; shift 2
mkdir -p "$R/build/var"
cd "$R/cooperative-search_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DCS_ONLY_N=$N "$@" \
  -I ../../include coopsearch.hip policy.hip episodes.hip -o "$R/build/var/$NAME.so" 2>&1 | grep -E "error|warning: var" -A3
exit ${PIPESTATUS[0]}
