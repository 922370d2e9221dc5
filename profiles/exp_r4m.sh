cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; rm -rf $O; mkdir -p $O; cd $R
export GPU_MAX_HW_QUEUES=24
echo "--- sc1 loads (product)"; timeout 200 python profiles/seq_probe.py 24 8 0 epzs 2>&1 | grep -E "depth"
echo "--- plain loads (timing experiment only)"; JMHIP_LIB=$R/jm_amd/libjmhip_plain.so timeout 200 python profiles/seq_probe.py 24 8 0 epzs 2>&1 | grep -E "depth"
cd /tmp; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/wc -o t -- $R/profiles/microbench/write_calib > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/wc/t_counter_collection.csv")))
rows.sort(key=lambda r:int(r["Dispatch_Id"]))
for r in rows: print(r["Kernel_Name"][:40], r["Counter_Value"])
PY
