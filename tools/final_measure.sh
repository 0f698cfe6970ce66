#!/bin/bash
# End-of-round measurement set (run on the GPU box through gpurun); everything lands under gpurun_out/final/.
# Every bench line that is committed next to a kernel trace comes out of the SAME process as that trace: the traced
# runs keep their own stdout line (*_traced.json); the untraced lines are what the driver's run looks like.
R="${GRAFT_REPO_ROOT:?}"; O="$R/gpurun_out/final"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench default rc=$?"; cp gpurun_out/bench_detail.json "$O/bench_default_detail.json"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_args.json" 2> "$O/bench_driver_args.err"; echo "bench driver rc=$?"; cp gpurun_out/bench_detail.json "$O/bench_driver_args_detail.json"
wc -c "$O/bench_default.json" "$O/bench_driver_args.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$O/trace_default" -o p -- python3 "$R/bench.py" --no-cpu-baseline > "$O/trace_default.log" 2>&1; echo "trace default rc=$?"; cp "$R/gpurun_out/bench_detail.json" "$O/bench_default_traced_detail.json"
rocprofv3 --kernel-trace --stats -d "$O/trace_driver" -o p -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/trace_driver.log" 2>&1; echo "trace driver rc=$?"; cp "$R/gpurun_out/bench_detail.json" "$O/bench_driver_args_traced_detail.json"
cd "$R"
grep -h '^{"metric"' "$O/trace_default.log" | tail -1 > "$O/bench_default_traced.json"
grep -h '^{"metric"' "$O/trace_driver.log" | tail -1 > "$O/bench_driver_args_traced.json"
python tools/prof_summary.py "$O/trace_default/p_results.db" > "$O/trace_default_kernel_stats.txt" 2>&1
python tools/prof_summary.py "$O/trace_driver/p_results.db" > "$O/trace_driver_kernel_stats.txt" 2>&1
rm -f "$O"/trace_*/p_results.db   # keep gpurun_out small: the summaries are what gets committed
head -14 "$O/trace_default_kernel_stats.txt"
