set -x
mkdir -p gpurun_out
bash tools/_run14.sh > gpurun_out/r27_evidence.log 2>&1
tail -n 3 gpurun_out/r27_evidence.log
timeout 2000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r27_suite.log 2>&1
tail -n 4 gpurun_out/r27_suite.log
