#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3y
python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -m gpu -x -q -k "yield or reserved or bench or reducer or comm" > gpurun_out/r3y/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3y/pytest.log
