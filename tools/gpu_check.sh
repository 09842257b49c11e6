#!/bin/bash
# One GPU-box session: micro-test, the GPU test suite, short bench lines of every configuration.  Output under gpurun_out/.
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
{
  echo "== reduce_test"; timeout 60 tools/_bin/reduce_test
  echo "== pytest -m gpu"; timeout ${PYTEST_TIMEOUT:-1200} python -m pytest tests -q -m gpu -x --timeout=600 2>&1 | tail -40
  for cfg in K3 K3-256 K3-2048 K5; do
    echo "== bench $cfg"; timeout 300 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -3
  done
} > gpurun_out/check.log 2>&1
tail -60 gpurun_out/check.log
