#!/bin/bash
# round 3 end-of-round set: GPU tests, bench lines + kernel traces, PMC traffic, SQ counters of the headline kernel, batch sweep
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/final_gpu_tests.log
bash tools/final_measure.sh 2>&1 | tail -16
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; echo "pmc traffic rc=$?"
bash tools/pmc.sh od3 k_rollout_od tools/exp_workload.py flight_easy 3 ode 4096 rollout 4 100 > gpurun_out/pmc_od3.log 2>&1; echo "pmc od3 rc=$?"; tail -40 gpurun_out/pmc_od3.log
bash tools/pmc.sh oct3 k_rollout_oct tools/exp_workload.py flight_easy 3 oct 32768 rollout 4 100 > gpurun_out/pmc_oct3.log 2>&1; echo "pmc oct3 rc=$?"
bash tools/pmc.sh lane3 k_rollout_lane tools/exp_workload.py flight_easy 3 lane 262144 rollout 3 100 > gpurun_out/pmc_lane3.log 2>&1; echo "pmc lane3 rc=$?"
python tools/batch_sweep.py > gpurun_out/batch_sweep.md 2> gpurun_out/batch_sweep.err; echo "sweep rc=$?"
