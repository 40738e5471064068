#!/bin/bash
# one wave per SIMD (512 registers) against two waves per SIMD (256 registers, StepCoef in LDS) at batches that give every SIMD >= 2 waves
for B in 131072 262144 524288; do for occ in 1 2; do
  GLGYM_OCC=$occ python bench.py --batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-alt-scheme 2>/dev/null | grep "^{" > /tmp/occ_line.json
  python - "$occ" <<'PY'
import json, sys
d = json.load(open("/tmp/occ_line.json"))
print("B", d["config"]["batch_per_gpu"], "GLGYM_OCC", sys.argv[1], "%.3e env-steps/s" % d["value"], "%.3f ms/step" % d["ms_per_step"],
      "kernel %.3f ms" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"], "failed", d["ode_failures"])
PY
done; done
