#!/bin/bash
# VN block dealing weights (plan.cpp, LDPC_AMD_VN_COST=a,b: cost of a block = a*degree + b) on the headline workload
for r in 1 2; do for c in 2,2 2,4 1,4 1,5 2,9; do
  echo -n "VN_COST=$c (round $r): "
  LDPC_AMD_VN_COST=$c timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pmc --steps 60 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('ms/step %.3f  kernel %.3f'%(j['ms_per_step'], j['roofline'].get('kernel_ms_avg') or 0))"
done; done
