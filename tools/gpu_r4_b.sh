#!/bin/bash
# round 4, call B: instruction-cache counters of the big rollout kernels (is the hot loop larger than the instruction cache?)
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
G="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES;SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY"
LIB=$R/build/var/base_n5.so bash tools/pmc_groups.sh ic_lane5 k_rollout_lane "$G" tools/exp_workload.py flight_easy 5 lane 262144 rollout 3 100 2>&1 | tail -26
LIB=$R/build/var/occ1_n5.so bash tools/pmc_groups.sh ic_lane5_occ1 k_rollout_lane "$G" tools/exp_workload.py flight_easy 5 lane 262144 rollout 3 100 2>&1 | tail -26
LIB=$R/build/var/base_n3.so bash tools/pmc_groups.sh ic_lane3 k_rollout_lane "$G" tools/exp_workload.py flight_easy 3 lane 262144 rollout 3 100 2>&1 | tail -26
LIB=$R/build/var/base_n5.so bash tools/pmc_groups.sh ic_oct5 k_rollout_oct "$G" tools/exp_workload.py flight_easy 5 oct 262144 rollout 3 100 2>&1 | tail -26
LIB=$R/build/var/base_n5.so bash tools/pmc_groups.sh ic_od5 k_rollout_od "$G" tools/exp_workload.py flight_easy 5 od 16384 rollout 3 100 2>&1 | tail -26
LIB=$R/build/var/base_n3.so bash tools/pmc_groups.sh ic_ode3 k_rollout_od "$G" tools/exp_workload.py flight_easy 3 ode 4096 rollout 3 100 2>&1 | tail -26
