# measurement aid: parity of the sequence entry points, the bench line, and the two PMC passes of the one-launch bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_seq.py tests/test_gpu_mbenc.py -x -q 2>&1 | tail -3
for s in 20 40; do python bench.py --steps $s --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 2>/dev/null | tail -1 > $O/bench_$s.json; python -c "
import json,sys
d=json.load(open('$O/bench_$s.json')); print(d['steps'], d['value'], d['ms_per_step'], d['config']['records_equal_jm'], d['config']['records_equal_picture_after_picture'], d.get('configs2',{}))"; done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 > /dev/null 2> $O/pmc1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 > /dev/null 2> $O/pmc2.err
python - <<PY
import csv, collections
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    rs = list(csv.DictReader(open("$O/%s/t_counter_collection.csv" % d)))
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    v = [float(r["Counter_Value"]) for r in rs if r["Kernel_Name"].startswith("k_mb_pipe(")]
    print(c, "KB: I picture %.0f, warm-up launch (5 pictures) %.0f, timed launch (20 pictures) %.0f = %.0f per picture; one-picture launches %.0f" % (v[0], v[1], v[2], v[2] / 20, sum(v[4:29]) / 25))
PY
