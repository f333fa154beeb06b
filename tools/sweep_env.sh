#!/bin/bash
# GPU box: sweep kernel scheduling knobs (env vars read by mipt_render_device) on config M
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', 'ms', r['roofline']['kernel_ms'], 'Mray/s', r['roofline']['mray_s_kernel'])"; }
for t in 1 4 8 12 16 24 32; do MIPT_LEAF_THRESH=$t run "leaf_thresh=$t"; done
for nd in "1 8" "1 6" "1 4" "1 3" "1 2"; do set -- $nd; MIPT_SERVICE_NUM=$1 MIPT_SERVICE_DEN=$2 run "service=$1/$2"; done
