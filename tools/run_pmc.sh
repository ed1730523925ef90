#!/bin/bash
# PMC passes (own runs, no other tracing domains): FETCH_SIZE and WRITE_SIZE for the bench command.
set -u
exec </dev/null
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  raw=/tmp/pmc_raw_${tag}_$c
  rm -rf "$raw"; mkdir -p "$raw"
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$raw" -o run -- \
      python bench.py --no-cpu-baseline --no-profile --no-exact --no-configs "$@" > "$out/bench_$c.json" 2> "$out/bench_$c.err"
  echo "rc=$?" >> "$out/bench_$c.err"
  f=$(find "$raw" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|k_conv_|k_split_rows|k_win_build|k_attn_" "$f" > "$out/conv_$c.csv"; head -2 "$out/conv_$c.csv"; else echo "no counter csv for $c"; find "$raw" | head; fi
  rm -rf "$raw"
done
python tools/pmc_summary.py "$out/conv_FETCH_SIZE.csv" "$out/conv_WRITE_SIZE.csv" "$out/pmc_conv.json" "${PASCO_COMMIT:-unknown}"
