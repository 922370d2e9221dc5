#!/usr/bin/env python3
"""Turn one gpurun_out/<dir> collection of profiles/collect2.sh into profiles/<tag>_{bench.json,kernel_stats.csv,kernel_stats.md}.
usage: python profiles/make_summary2.py gpurun_out/<dir> <tag> "title" """
import collections, csv, json, shutil, sys
O, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(f"{O}/stats/t_kernel_stats.csv")))
out = [f"# {title} -- rocprofv3 --kernel-trace --stats", "",
       "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --streams 0` (the I picture that makes the "
       "reference, 12 steps incl. warm-up, 5 + 1 further launches for the per-launch time and the record check), 1x MI355X, configs[1]", "",
       "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
for r in rows:
    out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.3f} | {float(r['Percentage']):.2f} |")
b, bb = json.load(open(f"{O}/bench_prof.json")), json.load(open(f"{O}/bench.json"))
out += ["", f"bench line of the profiled run: ms_per_step {b['ms_per_step']}, k_mb_pipe by HIP events {b['roofline']['avg_kernel_ms']} ms (its average above includes the one I-picture launch)",
        f"bench line without the profiler (profiles/{tag}_bench.json): {bb['value']} MB/s, ms_per_step {bb['ms_per_step']}, k_mb_pipe {bb['roofline']['avg_kernel_ms']} ms, records_equal_jm {bb['config']['records_equal_jm']}",
        f"end to end (lencod_hip.exe): {json.dumps({k: bb.get('end_to_end', {}).get(k) for k in ('p_frame_ms', 'macroblocks_per_s', 'md5_ok', 'speedup_vs_cpu_jm_p_frame')})}; CPU JM P picture {bb.get('cpu_baseline', {}).get('p_frame_ms')} ms",
        "", "## HBM traffic from PMC counters (separate passes, `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, bench.py --steps 3 --warmup 1)", "",
        "Unit KB per launch; FETCH_SIZE x 2 on gfx950 (calibration: profiles/r01_v3_kernel_stats.md, profiles/microbench/fetch_calib.hip).  k_mb_pipe: P-picture launches only "
        "(the largest values; the I-picture launch reads no reference).", "",
        "| kernel | launches | FETCH_SIZE KB | x2 = read MB | WRITE_SIZE KB | traffic MB |", "|---|---|---|---|---|---|"]
acc = collections.defaultdict(dict)
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    t = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{O}/{d}/t_counter_collection.csv")):
        t[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    for k, v in t.items():
        if k == "k_mb_pipe":
            v = sorted(v)[len(v) // 2:]          # the P-picture launches
        acc[k][c] = sum(v) / len(v); acc[k]["n"] = len(v)
for k in sorted(acc):
    f, w = acc[k].get("FETCH_SIZE", 0), acc[k].get("WRITE_SIZE", 0)
    out.append(f"| `{k}` | {acc[k]['n']} | {f:.0f} | {2*f*1024/1e6:.2f} | {w:.0f} | {(2*f+w)*1024/1e6:.2f} |")
k = "k_mb_pipe"
print("k_mb_pipe traffic bytes per launch =", round((2 * acc[k].get("FETCH_SIZE", 0) + acc[k].get("WRITE_SIZE", 0)) * 1024))
open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(out) + "\n")
shutil.copy(f"{O}/bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"{O}/stats/t_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
print("\n".join(out))
