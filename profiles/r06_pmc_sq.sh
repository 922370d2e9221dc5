# round 6: what bounds the waves of k_mb_pipe -- instruction cache, issue, LDS?  SQ / SQC counters over the timed launch (eight-wave form unless JMHIP_FS_WAVES says otherwise)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
B="python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0"
export JMHIP_FS_WAVES=${2:-8}
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $O/p1 -o t -- $B > /dev/null 2> $O/p1.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p2 -o t -- $B > /dev/null 2> $O/p2.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/p3 -o t -- $B > /dev/null 2> $O/p3.err
timeout 600 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_TC_INST_REQ SQC_TC_STALL SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $O/p4 -o t -- $B > /dev/null 2> $O/p4.err
python - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in agg.items():
            if "mb_pipe" in k:
                print(p, k, {a: "%.4g" % b for a, b in d.items()})
PY
