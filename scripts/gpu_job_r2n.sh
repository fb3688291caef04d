#!/bin/bash
mkdir -p gpurun_out/r2n
O=gpurun_out/r2n
(timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "beyond_2gib or conv" > $O/t.log 2>&1; echo rc=$? >> $O/t.log); tail -5 $O/t.log
