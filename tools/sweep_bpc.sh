#!/bin/bash
# GPU box: occupancy sweep of the trace kernel (blocks of 256 threads per CU) on config M
for b in 1 2 3 4 5; do
  MIPT_BLOCKS_PER_CU=$b python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('bpc', $b, 'ms', r['roofline']['kernel_ms'], 'Mray/s', r['roofline']['mray_s_kernel'])"
done
