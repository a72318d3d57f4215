#!/bin/bash
mkdir -p gpurun_out/s9
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "skinny or time" > gpurun_out/s9/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s9/pytest.log
timeout 600 python tools/b1_decompose.py > gpurun_out/s9/decomp.log 2>&1
tail -3 gpurun_out/s9/pytest.log
cat gpurun_out/s9/decomp.log
