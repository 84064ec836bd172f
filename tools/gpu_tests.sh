#!/bin/bash
# run the GPU test tier the way the driver does; log to gpurun_out/
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/gpu_tests.log
exit $rc
