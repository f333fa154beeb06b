#!/bin/bash
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', 'ms', r['roofline']['kernel_ms'], 'Mray/s', r['roofline']['mray_s_kernel'], r.get('parity'))"; }
MIPT_LDS_TOP=0 run "lds_top=0"
MIPT_LDS_TOP=1 run "lds_top=1"
MIPT_LDS_TOP=0 run "lds_top=0"
MIPT_LDS_TOP=1 run "lds_top=1"
