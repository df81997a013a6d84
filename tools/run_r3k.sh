mkdir -p gpurun_out/r3k
timeout -k 10 600 python -m pytest tests/test_gpu_volpath.py -x -q > gpurun_out/r3k/pytest.log 2>&1; tail -15 gpurun_out/r3k/pytest.log
timeout -k 10 300 python3 tools/dev.py rate dragon-fog dragon-fog-ww dragon-class dragon-ww --launches 2 --no-tune > gpurun_out/r3k/rate.log 2>&1; grep -v amdgpu.ids gpurun_out/r3k/rate.log
