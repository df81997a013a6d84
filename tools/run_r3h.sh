mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_scenes.py tests/test_gpu_t2.py -x -q > gpurun_out/r3h/pytest.log 2>&1; tail -3 gpurun_out/r3h/pytest.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err; tail -c 6000 gpurun_out/r3h/bench.json; tail -3 gpurun_out/r3h/bench.err
bash tools/hang_probe.sh 6 > gpurun_out/r3h/hang.log 2>&1; cat gpurun_out/r3h/hang.log
