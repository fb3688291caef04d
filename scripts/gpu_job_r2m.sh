#!/bin/bash
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
(timeout 1500 python -m pytest tests -q -m gpu -x > $O/t.log 2>&1; echo rc=$? >> $O/t.log); tail -4 $O/t.log
