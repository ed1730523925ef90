#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/run_profile.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of bench.py; keeps only the small CSV summaries under
# gpurun_out/profile_<tag>/ (copy them to profiles/ to commit).
set -u
exec </dev/null
tag=$1; shift
out=gpurun_out/profile_$tag
raw=/tmp/rocprof_raw_$tag
mkdir -p "$out" "$raw"
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$raw" -o run -- \
    python bench.py --no-cpu-baseline --no-exact --no-configs "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "rocprofv3 rc=$?" >> "$out/bench.err"
for f in $(find "$raw" -name "*stats*.csv" 2>/dev/null); do cp "$f" "$out/"; done
ls -la "$raw" $(find "$raw" -type d | head -3) > "$out/files.txt" 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -25 "$f"; else echo "no kernel_stats.csv"; cat "$out/files.txt"; fi
rm -rf "$raw"
