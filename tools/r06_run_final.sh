#!/bin/bash
# final state: the whole GPU suite, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06_final_tests.txt
grep -E "passed|failed" gpurun_out/r06_final_tests.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" | tail -3
python bench.py 2>/dev/null | grep '^{"metric' > gpurun_out/r06_final_bench_line.json
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r06_final_bench_line.json").read().strip().splitlines()[-1])
print("value %.2f ms/step %.2f frac %.3f frac_serial %.3f cpu %.4f fp32x3 %.2f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("frac_serial", 0), r["cpu_baseline"]["value"], r["fp32_mode"]["ms_per_step"]))
PY
