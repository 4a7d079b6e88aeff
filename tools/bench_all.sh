#!/bin/bash
# Round results beyond the default bench line: the other BASELINE configurations, the selectivity sweep of
# config 2 and the kernel-per-node mode.  One JSON line each into gpurun_out/bench_all.jsonl.
OUT=gpurun_out/bench_all.jsonl; : > $OUT
run() { timeout -k 10 240 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" >> $OUT 2>> gpurun_out/bench_all.err || echo "{\"failed\": \"$*\"}" >> $OUT; }
run --workload config2
run --workload config2_swapped
run --workload config2 --null-pct 1
run --workload config2 --selectivity 0.01
run --workload config2 --selectivity 0.10
run --workload config2 --selectivity 0.50
run --workload config2 --selectivity 1.00
run --workload config3
run --workload config4
run --workload config1 --rows 1000000
run --workload config2 --exec-mode per_node --steps 3 --warmup 1
python - <<'PY'
import json
for line in open("gpurun_out/bench_all.jsonl"):
    j = json.loads(line)
    if "failed" in j: print("FAILED", j); continue
    r = j["roofline"]
    print(f'{j["config"]["workload"][:70]:70s} {j["config"]["exec_mode"]:8s} rows {j["config"]["rows_per_gpu"]:>11d} sel {j["config"]["selected_rows_total"]/j["config"]["rows_total"]:.4f} '
          f'step {j["ms_per_step"]:.3f} ms kernel {r["kernel_ms"]:.3f} ms {r["achieved"]:.0f} GB/s frac {r["frac"]:.3f} frac_moved {r.get("frac_moved") or 0:.3f} '
          f'rows/s {j["value"]:.3e} {r["kernel"]}')
PY
timeout -k 10 120 python tools/bench_q6.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_q6.txt
