#!/bin/bash
# quick parity gate + timing of the tree's library: a subset of the GPU suite, then the kernel times
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
bash scripts/experiments/ab_bench.sh
