#!/bin/bash
# GPU box: the whole GPU test suite on the in-tree library, then bench.py A/B of the given variants (interleaved, twice)
# tools/ab_tests_bench.sh a.so b.so ...
set -e
mkdir -p gpurun_out/abt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/abt/pytest.log 2>&1 || { tail -30 gpurun_out/abt/pytest.log; exit 1; }
tail -2 gpurun_out/abt/pytest.log
tools/ab_bench.sh "$@"
