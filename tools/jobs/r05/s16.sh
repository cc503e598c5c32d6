#!/bin/bash
# r05 session 16: the gather + VALU kernel's shader clock alone and beside the writer
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 400 python tools/ubench/gather_beside_writer.py 2>&1 | grep -v amdgpu.ids | tee $O/gather_beside_writer.log
