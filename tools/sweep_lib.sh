#!/bin/bash
# GPU box: A/B kernel build variants (rust_ray_tracing_amd/variants/*.so) on config M
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', 'ms', r['roofline']['kernel_ms'], 'Mray/s', r['roofline']['mray_s_kernel'])"; }
run base
for f in rust_ray_tracing_amd/variants/*.so; do MIPT_LIB=$PWD/$f run $(basename $f); done
