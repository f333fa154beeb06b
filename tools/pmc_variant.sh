#!/bin/bash
# GPU box: the same kernel trace + PMC passes as tools/pmc.sh, for a configuration bench.py does not time -- driven by tools/ab.py
# (4 launches of the product library, the first one dropped as warm-up):
#   tools/pmc_variant.sh mode1     "--shading 1"   "pt_trace_kernel<false, true, 1>"
#   tools/pmc_variant.sh unculled  "--traversal 0" "pt_trace_kernel<false, false, 0>"
tag=$1; abargs=$2; filt=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 tools/ab.py --reps 4 $abargs rust_ray_tracing_amd/libmipt.so"
out=gpurun_out/pmc_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/ab.txt 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p1 -- $B > /dev/null 2> $out/p1.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p2 -- $B > /dev/null 2> $out/p2.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_REQ_sum --output-format csv -d $out/p5 -- $B > /dev/null 2> $out/p5.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/p3 -- $B > /dev/null 2> $out/p3.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/p4 -- $B > /dev/null 2> $out/p4.err
cat $out/ab.txt
python3 tools/pmc_summary.py $out "$filt" "tools/ab.py--reps4${abargs// /}"
