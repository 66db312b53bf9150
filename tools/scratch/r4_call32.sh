#!/bin/bash
# round-4 call 32: the order-4/5 tests on other shapes (exact window, two species)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c32
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "orders_4 or order_6" > gpurun_out/c32/tests.log 2>&1; rc=$?; tail -15 gpurun_out/c32/tests.log; exit $rc
