#!/bin/bash
# Copies the summaries of a `bash tools/gpu_r6_final.sh` pass from gpurun_out/ into profiles/r06_* (what gets committed) and regenerates
# profiles/traffic.json.   usage (in the build container, after gpurun merged its output): bash tools/install_profiles.sh [bench-only]
set -eu
cd "$(dirname "$0")/.."
F=gpurun_out/final
cp $F/bench_default.json profiles/r06_bench_default.json
cp $F/bench_default_detail.json profiles/r06_bench_default_detail.json
cp $F/bench_driver_args.json profiles/r06_bench_driver_args.json
cp $F/bench_driver_args_detail.json profiles/r06_bench_driver_args_detail.json
cp $F/bench_default_traced.json profiles/r06_bench_default_traced.json
cp $F/bench_driver_args_traced.json profiles/r06_bench_driver_args_traced.json
cp $F/bench_default_traced_detail.json profiles/r06_bench_default_traced_detail.json
cp $F/trace_default_kernel_stats.txt profiles/r06_bench_default_traced_kernel_stats.txt
cp $F/trace_driver_kernel_stats.txt profiles/r06_bench_driver_args_traced_kernel_stats.txt
[ "${1:-}" = "bench-only" ] && exit 0
cp gpurun_out/traffic/traffic_raw.json profiles/r06_traffic_raw.json
for k in od3 ode5 od5; do cp gpurun_out/pmc_$k/summary.json profiles/r06_${k}_pmc.json; done
cp gpurun_out/batch_sweep.md profiles/r06_batch_sweep.md
(echo "# flight (probability map) batch sweep and team sweep, round 6 (tools/flight_sweep.py)"; echo; cat gpurun_out/flight_batch.md; echo; cat gpurun_out/flight_teams.md) > profiles/r06_flight_sweep.md
tail -16 gpurun_out/final_gpu_tests.log > profiles/r06_gpu_tests.log
python tools/make_traffic_json.py > /dev/null
