#!/bin/bash
# PMC passes over one replayed convolution launch: tools/run_pmc_layer.sh <tag> <kernel id> <min rows> <kernel name pattern>
set -u
exec </dev/null
tag=$1; kid=$2; rows=$3; pat=$4
out=gpurun_out/pmc_$tag; mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1)); raw=/tmp/pmc_raw_${tag}_$i; rm -rf "$raw"; mkdir -p "$raw"
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$raw" -o run -- python tools/layer_only.py $kid $rows 3 > "$out/log_$i.txt" 2>&1
  f=$(find "$raw" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > "$out/set_$i.csv"; grep -E "$pat" "$f" | tail -n 4000 >> "$out/set_$i.csv"; else echo "no csv set $i"; tail -3 "$out/log_$i.txt"; fi
  rm -rf "$raw"
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/set_*.csv")):
    rows = list(csv.DictReader(open(f)))
    # the replayed launches are the last ones: keep the dispatches with the largest grid
    if not rows: continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    agg = collections.defaultdict(list)
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        v = v[-3:]
        print(k.ljust(62), c.ljust(34), "%.5g" % (sum(v) / len(v)), "(n=%d)" % len(v))
PY
