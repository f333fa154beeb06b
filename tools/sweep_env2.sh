#!/bin/bash
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', 'ms', r['roofline']['kernel_ms'], 'Mray/s', r['roofline']['mray_s_kernel'], r.get('parity'))"; }
MIPT_XCD_QUEUES=0 run "xcd_queues=0"
MIPT_XCD_QUEUES=1 run "xcd_queues=1"
MIPT_XCD_QUEUES=1 MIPT_REVERSE_TILES=1 run "xcd_queues=1,reverse"
for nd in "1 6" "1 4" "1 3" "1 2"; do set -- $nd; MIPT_SERVICE_NUM=$1 MIPT_SERVICE_DEN=$2 run "service=$1/$2"; done
for b in 4 5; do MIPT_BLOCKS_PER_CU=$b run "bpc=$b"; done
