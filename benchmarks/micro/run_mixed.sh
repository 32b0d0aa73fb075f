set -uo pipefail
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 "$root/benchmarks/bench_mixed.py" --images 125000 > "$out/r3_mixed.jsonl" 2>/dev/null
python3 "$root/benchmarks/bench_mixed.py" --images 125000 --dhash >> "$out/r3_mixed.jsonl" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r3_stats_mixed" -- python3 "$root/benchmarks/bench_mixed.py" --images 125000 --dhash --reps 2 > "$out/r3_stats_mixed.log" 2>&1
python3 "$root/benchmarks/shape_grid.py" > "$out/r3_grid_p.txt" 2>/dev/null
python3 "$root/benchmarks/shape_grid.py" dhash > "$out/r3_grid_pd.txt" 2>/dev/null
cat "$out/r3_mixed.jsonl"; cat "$out/r3_grid_p.txt" "$out/r3_grid_pd.txt"
f=$(find "$out/r3_stats_mixed" -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-200
