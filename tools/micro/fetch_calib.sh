# FETCH_SIZE per dispatch of a command, in dispatch order:  bash tools/micro/fetch_calib.sh <filter> <cmd...>   (GPU box)
R=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $R
F=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/lw_pmc -- "$@" > $R/lw_out.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/lw_pmc/**/*counter_collection.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if r["Counter_Name"]=="FETCH_SIZE" and "$F" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Dispatch_Id"]))
for r in rows: print("%6s %-70s FETCH_SIZE x 1024 = %8.1f MB" % (r["Dispatch_Id"], r["Kernel_Name"].replace("void eks::","")[:70], float(r["Counter_Value"])*1024/1e6))
PY
cp $(find $R/lw_pmc -name "*counter_collection.csv" | head -1) $R/lw_counters.csv
rm -rf $R/lw_pmc
