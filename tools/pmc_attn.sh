ROOT=$(pwd); O=$ROOT/gpurun_out/r2_pmc_attn; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/a -o run -- python3 $ROOT/tools/attn_only.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/b -o run -- python3 $ROOT/tools/attn_only.py > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/c -o run -- python3 $ROOT/tools/attn_only.py > $O/c.log 2>&1
cd $ROOT
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob
for d in ("a","b","c"):
    agg={}
    for f in glob.glob(f"gpurun_out/r2_pmc_attn/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_shared" not in r["Kernel_Name"]: continue
            e=agg.setdefault(r["Counter_Name"],[0,0.0,0.0]); e[0]+=1; e[1]+=float(r["Counter_Value"]); e[2]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for k,(n,v,t) in sorted(agg.items()): print(d, k, n, f"{v/n:.4g}", f"dur_us {t/n/1e3:.1f}")
PY
tail -3 $O/a.log $O/b.log $O/c.log
