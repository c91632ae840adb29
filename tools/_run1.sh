#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -5 > gpurun_out/r06_t9.txt
