# round 6: the last library as built by __graft_entry__.build(): smoke, the driver's bench command, the drop-in regression case, a subset of the suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_last; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_20_full.json 2> $O/bench_20_full.err; python - <<P
import json
d=json.load(open("$O/bench_20_full.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["traffic_over_algorithmic"], r["valu_frac_issued"], d["config"]["records_equal_jm"], d["config"]["records_equal_picture_after_picture"], d["cpu_baseline"]["value"], d["end_to_end"]["p_frame_ms"], d["end_to_end"]["md5_ok"], d["configs2"]["end_to_end"]["p_frame_ms_hip"])
P
timeout 600 python -m pytest tests/test_lencod_dropin.py -q -m gpu -k "many_times or teardown or m3h" 2>&1 | tail -2
