cd /root/repo
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_tolerance.py -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|FAILED|assert" gpurun_out/t.log | tail -8
