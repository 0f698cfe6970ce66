#!/bin/bash
# round 3 measurement set: bench lines + kernel traces, PMC traffic passes, batch sweep
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
bash tools/final_measure.sh 2>&1 | tail -25
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; echo "pmc traffic rc=$?"; tail -3 gpurun_out/pmc_traffic.log
python tools/batch_sweep.py > gpurun_out/batch_sweep.md 2> gpurun_out/batch_sweep.err; echo "sweep rc=$?"; tail -60 gpurun_out/batch_sweep.md
