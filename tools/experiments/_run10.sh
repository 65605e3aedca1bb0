python tools/ab/geglu_fused.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r10_geglu_micro.log
python -m pytest tests/test_gpu_ops.py -x -q -k "linear_geglu" 2>&1 | tail -n 3
