#!/bin/bash
# Same-box A/B of library builds on the secondary configurations: tools/ab_configs.sh ab/libldpc_X.so ...
set -e
cp libldpc_amd/libldpc.so /tmp/libldpc_orig.so
for v in "$@"; do
  cp "$v" libldpc_amd/libldpc.so
  echo "== $v"
  timeout -k 10 600 python3 tools/bench_configs.py 2>&1 | grep config | python3 -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l); print('  ', j['config'][:58], round(j['kernel_ms'],2))"
done
cp /tmp/libldpc_orig.so libldpc_amd/libldpc.so
