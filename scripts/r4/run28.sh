#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
bash scripts/r4/ab.sh r4ab default unif default unif
