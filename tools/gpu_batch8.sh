#!/bin/bash
mkdir -p gpurun_out/b8
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/b8/pytest.log 2>&1; tail -14 gpurun_out/b8/pytest.log
