mkdir -p gpurun_out/r4x
for i in 1 2 3; do
  for v in 1 0; do
    PASCO_CONV_LIN=$v timeout 200 python bench.py --steps 48 --no-cpu-baseline --no-exact --no-configs --no-profile > gpurun_out/r4x/abab_lin${v}_$i.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("gpurun_out/r4x/abab_lin${v}_$i.json").read().strip().splitlines()[-1])
print("PASCO_CONV_LIN=$v run $i:", d["value"], "scenes/s", d["ms_per_step"], "ms; rounds", d["step_ms_by_round"]["rounds"], "one at a time", d.get("in_flight_1", {}).get("ms_per_step"))
PY
  done
done
