#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4p
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4p/test.txt 2>&1
tail -5 gpurun_out/r4p/test.txt
