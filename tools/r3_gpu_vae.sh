#!/bin/bash
O=$PWD/gpurun_out/r3n; mkdir -p $O
timeout 300 python tools/ff_trace.py --shipped 1 2 2>&1 | grep -v amdgpu.ids > $O/ff_shipped.txt
cat $O/ff_shipped.txt
timeout 600 python tools/vae_bound.py --frames 8 2>&1 | grep -v amdgpu.ids > $O/vae_bound.txt
cat $O/vae_bound.txt
