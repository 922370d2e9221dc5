#!/usr/bin/env python3
"""Turn one gpurun_out/<dir> collection of profiles/collect5.sh into profiles/<tag>_{bench.json,bench_20.json,kernel_stats.csv,kernel_stats.md}.
usage: python profiles/make_summary5.py gpurun_out/<dir> <tag> "title" """
import collections, csv, json, shutil, sys
O, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(f"{O}/stats/t_kernel_stats.csv")))
b, bb, b20 = json.load(open(f"{O}/bench_prof.json")), json.load(open(f"{O}/bench.json")), json.load(open(f"{O}/bench_20.json"))
K, Wm = b["steps"], b["warmup"]
nseq = 1 + Wm + K
out = [f"# {title} -- rocprofv3 --kernel-trace --stats", "",
       f"command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps {K} --warmup {Wm} --no-cpu-baseline --no-end-to-end --streams 0`, 1x MI355X, configs[1].  "
       f"k_mb_pipe launches: the I picture, ONE launch of the {Wm} warm-up P pictures, ONE launch of the {K} timed P pictures (jmhip_seq_batch: loop filter and interpolation inside), then "
       f"{nseq} launches of the same sequence picture after picture (the check: each followed by k_deblock_* and k_subplanes); k_mb_pipe_epzs*: the configs[2] figures", "",
       "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
for r in rows:
    out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.3f} | {float(r['Percentage']):.2f} |")
tr = [r for r in csv.DictReader(open(f"{O}/stats/t_kernel_trace.csv")) if r["Kernel_Name"].startswith("k_mb_pipe(")]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
i_pic, warm, timed, cls = tr[0], tr[1], tr[2], tr[3:3 + nseq]
out += ["", f"k_mb_pipe in the trace: the I picture {dur(i_pic):.3f} ms; the warm-up launch ({Wm} pictures) {dur(warm):.3f} ms; **the timed launch ({K} pictures) {dur(timed):.3f} ms = {dur(timed) / K:.3f} ms per picture**; "
            f"the same P pictures one launch each: {sum(dur(r) for r in cls[1:]) / max(1, len(cls) - 1):.3f} ms on average (+ the loop filter and the interpolation behind each)",
        f"bench line of the profiled run: ms_per_step {b['ms_per_step']}, the timed launch by HIP events {b['roofline']['avg_kernel_ms']} ms",
        f"bench line without the profiler, --steps {b20['steps']} --warmup {b20['warmup']} as the driver runs it (profiles/{tag}_bench_20.json): {b20['value']} MB/s, ms_per_step {b20['ms_per_step']}, the timed launch {b20['roofline']['avg_kernel_ms']} ms",
        f"bench line without the profiler, defaults (--steps {bb['steps']} --warmup {bb['warmup']}; profiles/{tag}_bench.json): {bb['value']} MB/s, ms_per_step {bb['ms_per_step']}, the timed launch {bb['roofline']['avg_kernel_ms']} ms, "
        f"records_equal_jm {bb['config']['records_equal_jm']} ({bb['config']['pictures_checked_against_jm']} pictures), equal to picture after picture {bb['config']['records_equal_picture_after_picture']} ({bb['config']['pictures_checked_against_picture_after_picture']} pictures)",
        f"end to end (lencod_hip.exe): {json.dumps({k: bb.get('end_to_end', {}).get(k) for k in ('p_frame_ms', 'macroblocks_per_s', 'md5_ok', 'speedup_vs_cpu_jm_p_frame')})}; CPU JM P picture {bb.get('cpu_baseline', {}).get('p_frame_ms')} ms",
        "", f"## HBM traffic from PMC counters (separate passes, `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over the same command)", "",
        "Unit KB per launch; FETCH_SIZE x 2 on gfx950 (calibration: profiles/r01_v3_kernel_stats.md, profiles/microbench/fetch_calib.hip); WRITE_SIZE counts 32-byte sectors: a 16-byte store counts twice "
        f"(profiles/r04_write_calib.txt).  `k_mb_pipe [timed launch]`: the {K} P pictures of the timed region, one launch, per PICTURE; `k_mb_pipe [alone]`: the P pictures of the picture-after-picture check, per launch = per picture.", "",
        "| kernel | launches | FETCH_SIZE KB | x2 = read MB | WRITE_SIZE KB | traffic MB |", "|---|---|---|---|---|---|"]
acc = collections.defaultdict(dict)
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    t = collections.defaultdict(list)
    rs = list(csv.DictReader(open(f"{O}/{d}/t_counter_collection.csv")))
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rs:
        t[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    for k, v in t.items():
        if k == "k_mb_pipe":                    # I, warm-up launch, timed launch, then the check's launches
            acc["k_mb_pipe [timed launch, per picture]"][c] = v[2] / K; acc["k_mb_pipe [timed launch, per picture]"]["n"] = 1
            acc["k_mb_pipe [alone]"][c] = sum(v[4:3 + nseq]) / (nseq - 1); acc["k_mb_pipe [alone]"]["n"] = nseq - 1
            continue
        acc[k][c] = sum(v) / len(v); acc[k]["n"] = len(v)
for k in sorted(acc):
    f, w = acc[k].get("FETCH_SIZE", 0), acc[k].get("WRITE_SIZE", 0)
    out.append(f"| `{k}` | {acc[k]['n']} | {f:.0f} | {2*f*1024/1e6:.2f} | {w:.0f} | {(2*f+w)*1024/1e6:.2f} |")
k = "k_mb_pipe [timed launch, per picture]"
print("k_mb_pipe traffic bytes per picture of the timed launch =", round((2 * acc[k].get("FETCH_SIZE", 0) + acc[k].get("WRITE_SIZE", 0)) * 1024))
open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(out) + "\n")
shutil.copy(f"{O}/bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"{O}/bench_20.json", f"profiles/{tag}_bench_20.json")
shutil.copy(f"{O}/stats/t_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
