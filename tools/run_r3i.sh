mkdir -p gpurun_out/r3i
python -m pytest tests -m gpu -x -q > gpurun_out/r3i/pytest.log 2>&1; tail -3 gpurun_out/r3i/pytest.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3i/bench.json") if l.startswith('{"metric"')][-1])
print("headline", round(d["value"]), "Mrays/s", "step median", round(d["step_ms_median"],2), "min", round(d["step_ms_min"],2), "sclk", d["roofline"]["sclk_mhz"], "identical", d["jobs_bit_identical"])
for k,v in d["configs"].items(): print(k, round(v.get("value",0)), v.get("seconds"), v.get("error"))
PY
