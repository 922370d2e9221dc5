# measurement aid: kernel trace of profiles/seq_probe.py (who runs when)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; rm -rf $O; mkdir -p $O; cd $R
export GPU_MAX_HW_QUEUES=24
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python profiles/seq_probe.py ${NPIC:-16} ${DEPTH:-8} 0 ${MODE:-epzs} > $O/probe.txt 2>&1
cat $O/probe.txt | grep -E "depth|launches"
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/tr/t_kernel_trace.csv")) if r["Kernel_Name"].startswith("k_mb_pipe")]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for i,r in enumerate(rows):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(i, r["Kernel_Name"][:18], "queue", r.get("Queue_Id"), "grid", r.get("Grid_Size"), "start %.1f end %.1f dur %.1f ms"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6))
PY
