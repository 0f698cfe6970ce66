#!/bin/bash
# round 5 measurement set: smoke, GPU tests, bench lines + kernel traces (tools/final_measure.sh), PMC traffic of every bench kernel
# (tools/pmc_traffic.sh), SQ counters of the 5-agent kernels, executed SGPR spills of the lane kernel (tools/spill_exec.py), batch
# sweep, flight sweep.  Everything lands under gpurun_out/; the summaries are copied into profiles/ afterwards.
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final_smoke.log
python -m pytest tests -m gpu -q --durations=10 > gpurun_out/final_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/final_gpu_tests.log
bash tools/final_measure.sh 2>&1 | tail -16
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; echo "pmc traffic rc=$?"
bash tools/pmc.sh od3 k_rollout_od tools/exp_workload.py flight_easy 3 ode 4096 rollout 4 100 > gpurun_out/pmc_od3.log 2>&1; echo "pmc od3 rc=$?"
bash tools/pmc.sh ode5 k_rollout_od tools/exp_workload.py flight_easy 5 ode 8192 rollout 4 100 > gpurun_out/pmc_ode5.log 2>&1; echo "pmc ode5 rc=$?"
bash tools/pmc.sh od5 k_rollout_od tools/exp_workload.py flight_easy 5 od 16384 rollout 4 100 > gpurun_out/pmc_od5.log 2>&1; echo "pmc od5 rc=$?"
bash tools/pmc.sh lanev5 k_rollout_lanev tools/exp_workload.py flight_easy 5 lanev 262144 rollout 3 100 > gpurun_out/pmc_lanev5.log 2>&1; echo "pmc lanev5 rc=$?"
bash tools/pmc.sh lanev3 k_rollout_lanev tools/exp_workload.py flight_easy 3 lanev 262144 rollout 3 100 > gpurun_out/pmc_lanev3.log 2>&1; echo "pmc lanev3 rc=$?"
for n in 3 5; do python tools/spill_exec.py run $n 262144 2>&1 | grep -v amdgpu.ids > gpurun_out/spill_exec_n$n.txt; tail -3 gpurun_out/spill_exec_n$n.txt; done
python tools/batch_sweep.py > gpurun_out/batch_sweep.md 2> gpurun_out/batch_sweep.err; echo "sweep rc=$?"
python tools/flight_sweep.py batch 2>&1 | grep "^|" > gpurun_out/flight_batch.md; python tools/flight_sweep.py teams 2>&1 | grep "^|" > gpurun_out/flight_teams.md; echo "flight sweep rc=$?"
