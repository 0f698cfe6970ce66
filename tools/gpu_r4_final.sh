#!/bin/bash
# round 4 end-of-round set: GPU tests, bench lines + kernel traces, PMC traffic, SQ counters of the headline and the 5-agent kernels,
# batch sweep.  Everything lands under gpurun_out/ (final/, traffic/, pmc_*/); the summaries are copied into profiles/ afterwards.
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
rm -f gpurun_out/lockstep_fractions.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final_smoke.log
python -m pytest tests -m gpu -q --durations=10 > gpurun_out/final_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/final_gpu_tests.log
bash tools/final_measure.sh 2>&1 | tail -16
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; echo "pmc traffic rc=$?"
bash tools/pmc.sh od3 k_rollout_od tools/exp_workload.py flight_easy 3 ode 4096 rollout 4 100 > gpurun_out/pmc_od3.log 2>&1; echo "pmc od3 rc=$?"
bash tools/pmc.sh lanev5 k_rollout_lanev tools/exp_workload.py flight_easy 5 lanev 262144 rollout 3 100 > gpurun_out/pmc_lanev5.log 2>&1; echo "pmc lanev5 rc=$?"; tail -30 gpurun_out/pmc_lanev5.log
bash tools/pmc.sh lanev3 k_rollout_lanev tools/exp_workload.py flight_easy 3 lanev 262144 rollout 3 100 > gpurun_out/pmc_lanev3.log 2>&1; echo "pmc lanev3 rc=$?"
bash tools/pmc.sh od5 k_rollout_od tools/exp_workload.py flight_easy 5 od 16384 rollout 4 100 > gpurun_out/pmc_od5.log 2>&1; echo "pmc od5 rc=$?"
python tools/batch_sweep.py > gpurun_out/batch_sweep.md 2> gpurun_out/batch_sweep.err; echo "sweep rc=$?"
for B in 4096 65536; do B=$B COOPSEARCH_LIB=$R/build/var/tlp_n3.so timeout 300 python tools/exp_policy_loop_timeline.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/policy_loop_timeline.txt; echo "timeline rc=$?"
