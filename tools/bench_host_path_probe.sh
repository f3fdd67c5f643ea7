#!/bin/bash
# Developer tool (GPU box): bench.py's host_path_ms under different OpenMP environments.
for env in "" "OMP_WAIT_POLICY=passive" "OMP_NUM_THREADS=1" "OPTY_HIP_HOST_THREADS=32"; do
  echo "### env: [$env]"
  env $env timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['config']['host_path_ms'])"
done
