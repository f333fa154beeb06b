cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/absq2; mkdir -p $out
L=rust_ray_tracing_amd/libmipt.so; Q=rust_ray_tracing_amd/variants/q1.so
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR --output-format csv -d $out/p1 -- python3 tools/ab.py --reps 2 $L $Q > $out/r1.txt 2> $out/p1.err
rocprofv3 --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM --output-format csv -d $out/p2 -- python3 tools/ab.py --reps 2 $L $Q > $out/r2.txt 2> $out/p2.err
python3 - $out <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
for d in sorted(glob.glob(out+"/p*/")):
    per=collections.defaultdict(dict)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pt_trace_kernel" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]]=per[int(r["Dispatch_Id"])].get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
    disp=sorted(per)
    for k in sorted(per[disp[1]]): print(f"{k:34s} old {per[disp[1]][k]:16.0f}  q {per[disp[3]][k]:16.0f}")
PY
