mkdir -p gpurun_out/r3s
python -m pytest tests -m gpu -x -q > gpurun_out/r3s/pytest.log 2>&1; tail -3 gpurun_out/r3s/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3s/bench.json 2> gpurun_out/r3s/bench.err; python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3s/bench.json") if l.startswith('{"metric"')][-1])
rl=d["roofline"]
print("headline", round(d["value"]), "Mrays/s", "step median", round(d["step_ms_median"],2), "min", round(d["step_ms_min"],2), "ms_per_step", round(d["ms_per_step"],2), "sclk", round(rl["sclk_mhz"]), "identical", d["jobs_bit_identical"], "frac", round(rl["frac"],3), "useful", round(rl["useful_lane_frac"],3), "stale", rl["pmc_stale"], "launch_ms", round(rl["launch_ms"],2))
for k,v in d["configs"].items(): print(k, round(v.get("value",0)), v.get("seconds"), (v.get("valu") or {}).get("frac"), (v.get("valu") or {}).get("useful_lane_frac"), (v.get("hbm") or {}).get("frac"), v.get("error"))
print(d["cpu_baseline"])
PY
