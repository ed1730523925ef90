#!/bin/bash
# SQ counter pass for the conv microbench: tools/run_sq.sh <tag> <level> [split]
set -u
exec </dev/null
tag=$1; level=$2; mode=${3:-}
out=gpurun_out/sq_$tag; mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1)); raw=/tmp/sq_raw_${tag}_$i; rm -rf "$raw"; mkdir -p "$raw"
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$raw" -o run -- python tools/conv_only.py $level 3 $mode > "$out/log_$i.txt" 2>&1
  f=$(find "$raw" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|k_conv_" "$f" > "$out/set_$i.csv"; else echo "no csv set $i"; tail -3 "$out/log_$i.txt"; fi
  rm -rf "$raw"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$out/set_*.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k.ljust(34), "%.4g" % (sum(v) / len(v)), "(n=%d)" % len(v))
PY
