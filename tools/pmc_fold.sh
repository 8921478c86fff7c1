# PMC passes of the folded cross-attention kernel alone (tools/fold_bench.py): separate --pmc passes, --kernel-trace only (the pool's rule)
ROOT=$(pwd); O=$ROOT/gpurun_out/pmc_fold; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/a -o run -- python3 $ROOT/tools/fold_bench.py 6720 f16 quick > $O/a.log 2>&1
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/b -o run -- python3 $ROOT/tools/fold_bench.py 6720 f16 quick > $O/b.log 2>&1
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/c -o run -- python3 $ROOT/tools/fold_bench.py 6720 f16 quick > $O/c.log 2>&1
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/d -o run -- python3 $ROOT/tools/fold_bench.py 6720 f16 quick > $O/d.log 2>&1
cd $ROOT
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob
for d in ("a","b","c","d"):
    agg={}
    for f in glob.glob(f"gpurun_out/pmc_fold/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "xattn_fold" not in r["Kernel_Name"]: continue
            e=agg.setdefault(r["Counter_Name"],[0,0.0,0.0]); e[0]+=1; e[1]+=float(r["Counter_Value"]); e[2]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for k,(n,v,t) in sorted(agg.items()): print(d, k, n, f"{v/n:.5g}", f"dur_us {t/n/1e3:.1f}")
PY
tail -2 $O/a.log $O/d.log
