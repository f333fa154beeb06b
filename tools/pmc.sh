#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes for the trace kernel on config M; writes gpurun_out/pmc_<tag>/ with
#   kernel_stats.csv  (--kernel-trace --stats of the default bench command)
#   pmc_summary.csv   (mean per launch of every counter over the launches AFTER the first, warm-up one; first line = provenance:
#                      kernel source hash, date, command)
# Counters are collected in their own runs, one group per pass (TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2; SQ 8).
#   tools/pmc.sh <tag> ["extra bench.py args"] [kernel-name filter, default the culled CPU-shading instantiation]
tag=${1:-run}
extra=${2:-}            # e.g. "--mode samples" or "--traversal reference"; recorded in the provenance line
filt=${3:-"pt_trace_kernel<false, true, 0>"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-render-multi $extra"
out=gpurun_out/pmc_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-render-multi $extra > $out/bench.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p1 -- $B > /dev/null 2> $out/p1.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p2 -- $B > /dev/null 2> $out/p2.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_REQ_sum --output-format csv -d $out/p5 -- $B > /dev/null 2> $out/p5.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/p3 -- $B > /dev/null 2> $out/p3.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/p4 -- $B > /dev/null 2> $out/p4.err
python3 tools/pmc_summary.py $out "$filt" "bench.py--steps3--warmup1${extra// /}"
