#!/bin/bash
python tools/ab/ff_chunked.py 2>&1 | grep -v amdgpu.ids
