#!/bin/bash
# Memory-path counters of the k = 1 stream kernels (k_conv_lin / k_conv_dma through tools/lin_time.py) next to the pure
# read / write mixes of the same shape (tools/ubench/store_bench 1): tools/run_pmc_lin.sh <tag>
set -u
exec </dev/null
tag=$1
out=gpurun_out/pmc_lin_$tag; mkdir -p "$out"
export TMPDIR=/tmp
i=0
# (round 4: the sets "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ..." and "TCP_PENDING_STALL_CYCLES_sum ... TA_BUSY_avr ..." abort
# rocprofv3 on this image (signal 6 after its own 300 s): they are not requested any more)
for set in "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_TAG_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for what in lin rw; do
    raw=/tmp/pmclin_raw_${tag}_${i}_$what; rm -rf "$raw"; mkdir -p "$raw"
    if [ $what = lin ]; then cmd="python tools/lin_time.py first"; else cmd="tools/ubench/store_bench 1"; fi
    timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$raw" -o run -- $cmd > "$out/log_${i}_$what.txt" 2>&1
    f=$(find "$raw" -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then grep -E "Counter_Name|k_conv_lin|k_conv_dma|k_rw" "$f" > "$out/set_${i}_$what.csv"; else echo "no csv set $i $what"; tail -3 "$out/log_${i}_$what.txt"; fi
    rm -rf "$raw"
  done
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$out/set_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[:k.index("(")] if "(" in k else k
        if k.startswith("void "): k = k[5:]
        key = (k, r.get("Grid_Size", ""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/summary.txt", "w") as fo:
    for key in sorted(agg):
        line = f"{key[0]} grid={key[1]}"
        print(line); fo.write(line + "\n")
        for c, v in agg[key].items():
            line = f"    {c:44s} {sum(v) / len(v):14.5g}  (n={len(v)})"
            print(line); fo.write(line + "\n")
PY
