#!/usr/bin/env python3
"""Turn one gpurun_out/<dir> collection of profiles/collect4.sh into profiles/<tag>_{bench.json,kernel_stats.csv,kernel_stats.md,timeline.txt}.
usage: python profiles/make_summary4.py gpurun_out/<dir> <tag> "title" """
import collections, csv, json, shutil, sys
O, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(f"{O}/stats/t_kernel_stats.csv")))
out = [f"# {title} -- rocprofv3 --kernel-trace --stats", "",
       "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --streams 0`, 1x MI355X, configs[1].  "
       "k_mb_pipe launches: 24 of the timed sequence (I + 3 warm-up + 20 timed P pictures, eight in flight, the loop filter and the interpolation inside each), then 24 of the same "
       "sequence picture after picture (the check), and k_mb_pipe_epzs_t8's for the configs[2] figure", "",
       "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
for r in rows:
    out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.3f} | {float(r['Percentage']):.2f} |")
b, bb = json.load(open(f"{O}/bench_prof.json")), json.load(open(f"{O}/bench.json"))
# ---- the timed sequence's launches from the kernel trace: who ran when
tr = [r for r in csv.DictReader(open(f"{O}/stats/t_kernel_trace.csv")) if r["Kernel_Name"].startswith("k_mb_pipe(")]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
nseq = b["steps"] + b["warmup"] + 1
seq, cls = tr[:nseq], tr[nseq:2 * nseq]
t0 = int(seq[0]["Start_Timestamp"])
tl = ["# the timed sequence's k_mb_pipe launches (rocprofv3 --kernel-trace), microseconds from the first launch's start: picture, stream (queue), start, end, duration, launches running at its start"]
for i, r in enumerate(seq):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    live = sum(1 for q in seq if int(q["Start_Timestamp"]) <= s < int(q["End_Timestamp"]))
    tl.append(f"{i:3d}  queue {r.get('Queue_Id', '?'):>3}  {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  {live}")
open(f"profiles/{tag}_timeline.txt", "w").write("\n".join(tl) + "\n")
timed = seq[1 + b["warmup"]:]
dur_seq = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed) / len(timed) / 1e6
dur_cls = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in cls[1:]) / max(1, len(cls) - 1) / 1e6
span = (max(int(r["End_Timestamp"]) for r in timed) - min(int(r["Start_Timestamp"]) for r in timed)) / 1e6
out += ["", f"the timed region's {len(timed)} launches in the trace: {dur_seq:.3f} ms each on average, {span:.1f} ms from the first one's start to the last one's end = {span / len(timed):.3f} ms per picture "
            f"(profiles/{tag}_timeline.txt: up to eight running at a time); the same pictures one launch at a time: {dur_cls:.3f} ms each",
        f"bench line of the profiled run: ms_per_step {b['ms_per_step']}, k_mb_pipe by HIP events {b['roofline']['avg_kernel_ms']} ms per launch in flight, {b['roofline']['per_launch']['avg_kernel_ms_alone']} ms alone",
        f"bench line without the profiler (profiles/{tag}_bench.json): {bb['value']} MB/s, ms_per_step {bb['ms_per_step']}, k_mb_pipe {bb['roofline']['avg_kernel_ms']} ms in flight / {bb['roofline']['per_launch']['avg_kernel_ms_alone']} alone, "
        f"records_equal_jm {bb['config']['records_equal_jm']} ({bb['config']['pictures_checked_against_jm']} pictures), equal to picture after picture {bb['config']['records_equal_picture_after_picture']} ({bb['config']['pictures_checked_against_picture_after_picture']} pictures)",
        f"end to end (lencod_hip.exe): {json.dumps({k: bb.get('end_to_end', {}).get(k) for k in ('p_frame_ms', 'macroblocks_per_s', 'md5_ok', 'speedup_vs_cpu_jm_p_frame')})}; CPU JM P picture {bb.get('cpu_baseline', {}).get('p_frame_ms')} ms",
        "", "## HBM traffic from PMC counters (separate passes, `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, bench.py --steps 3 --warmup 1; the profiler runs one kernel at a time)", "",
        "Unit KB per launch; FETCH_SIZE x 2 on gfx950 (calibration: profiles/r01_v3_kernel_stats.md, profiles/microbench/fetch_calib.hip).  `k_mb_pipe [in flight]`: the P pictures of the timed "
        "sequence (loop filter and interpolation inside); `k_mb_pipe [alone]`: the P pictures of the picture-after-picture check (followed by k_deblock_* and k_subplanes).", "",
        "| kernel | launches | FETCH_SIZE KB | x2 = read MB | WRITE_SIZE KB | traffic MB |", "|---|---|---|---|---|---|"]
acc = collections.defaultdict(dict)
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    t = collections.defaultdict(list)
    rs = list(csv.DictReader(open(f"{O}/{d}/t_counter_collection.csv")))
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rs:
        t[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    for k, v in t.items():
        if k == "k_mb_pipe":                    # 5 launches of the sequence (I, P x 4), then 5 of the check
            acc["k_mb_pipe [in flight]"][c] = sum(v[1:5]) / 4; acc["k_mb_pipe [in flight]"]["n"] = 4
            acc["k_mb_pipe [alone]"][c] = sum(v[6:10]) / 4; acc["k_mb_pipe [alone]"]["n"] = 4
            continue
        acc[k][c] = sum(v) / len(v); acc[k]["n"] = len(v)
for k in sorted(acc):
    f, w = acc[k].get("FETCH_SIZE", 0), acc[k].get("WRITE_SIZE", 0)
    out.append(f"| `{k}` | {acc[k]['n']} | {f:.0f} | {2*f*1024/1e6:.2f} | {w:.0f} | {(2*f+w)*1024/1e6:.2f} |")
k = "k_mb_pipe [in flight]"
print("k_mb_pipe traffic bytes per launch (in flight) =", round((2 * acc[k].get("FETCH_SIZE", 0) + acc[k].get("WRITE_SIZE", 0)) * 1024))
open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(out) + "\n")
shutil.copy(f"{O}/bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"{O}/stats/t_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
print("\n".join(out))
