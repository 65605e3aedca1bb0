set -x
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r20_suite.log 2>&1
tail -n 6 gpurun_out/r20_suite.log
