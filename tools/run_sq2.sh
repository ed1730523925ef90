#!/bin/bash
# Second SQ counter pass (LDS / TA FIFO pressure, issue levels) for the conv microbench: tools/run_sq2.sh <tag> <level> [split]
set -u
exec </dev/null
tag=$1; level=$2; mode=${3:-}
out=gpurun_out/sq2_$tag; mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_INSTS_FLAT" \
           "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_IFETCH"; do
  i=$((i+1)); raw=/tmp/sq2_raw_${tag}_$i; rm -rf "$raw"; mkdir -p "$raw"
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
