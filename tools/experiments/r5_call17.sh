#!/bin/bash
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu -k "deferred or segments or stale or old_flat or test_gemm or iteration_matches" 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-700
