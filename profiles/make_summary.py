#!/usr/bin/env python3
"""Turn one gpurun_out/<dir> collection (bench.json, bench_prof.json, stats/, pmc_fetch/, pmc_write/) into profiles/<tag>_*.
usage: python profiles/make_summary.py gpurun_out/r01d r01_v5 "title"""
import collections, csv, json, shutil, sys
O, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(f"{O}/stats/t_kernel_stats.csv")))
out = [f"# {title} -- rocprofv3 --kernel-trace --stats", "",
       "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline` (12 steps incl. warm-up), 1x MI355X, 1080p SR=32", "",
       "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
for r in rows:
    out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.3f} | {float(r['Percentage']):.2f} |")
b, bb = json.load(open(f"{O}/bench_prof.json")), json.load(open(f"{O}/bench.json"))
out += ["", f"bench line of the profiled run: ms_per_step {b['ms_per_step']}, stages by HIP events (ms): {[k['ms'] for k in b['kernels']]}",
        f"bench line without the profiler (profiles/{tag}_bench.json): {bb['value']} MB/s, ms_per_step {bb['ms_per_step']}, stages (ms): {[k['ms'] for k in bb['kernels']]}",
        "", "## HBM traffic from PMC counters (separate passes, `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, bench.py --steps 3 --warmup 1)", "",
        "Unit KB per launch; FETCH_SIZE x 2 on gfx950 (calibration: profiles/r01_v3_kernel_stats.md, profiles/microbench/fetch_calib.hip).", "",
        "| kernel | FETCH_SIZE KB | x2 = read MB | WRITE_SIZE KB | traffic MB | algorithmic MB |", "|---|---|---|---|---|---|"]
acc = collections.defaultdict(dict)
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    t = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{O}/{d}/t_counter_collection.csv")):
        t[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    for k, v in t.items():
        acc[k][c] = sum(v) / len(v)
alg = {"k_mb16_recon_luma": 8160 * (256 + 256 + 256 + 16 * 104) / 1e6, "k_me_fs_fast": 56.99, "k_subplanes": 2.09 + 35.81, "k_tq_luma4x4": 130560 * 136 / 1e6,
       "k_deblock_rows": 2 * 1.5 * 1920 * 1088 / 1e6 + 8160 * 192 / 1e6, "k_me_refine_mb": 278.0}
for k in ("k_me_fs_fast", "k_me_refine_mb", "k_subplanes", "k_copy_chroma_planes", "k_mb16_recon_luma", "k_mc_mb16", "k_tq_luma4x4", "k_tq_rec_to_plane", "k_mc_mb16_chroma", "k_tq_chroma", "k_tqc_rec_to_planes",
          "k_deblock_prep", "k_deblock_tasks", "k_deblock_sparse", "k_deblock_rows"):
    f, w = acc[k].get("FETCH_SIZE", 0), acc[k].get("WRITE_SIZE", 0)
    out.append(f"| `{k}` | {f:.0f} | {2*f*1024/1e6:.2f} | {w:.0f} | {(2*f+w)*1024/1e6:.2f} | {round(alg[k],2) if k in alg else ''} |")
tb = lambda k: (2 * acc[k].get("FETCH_SIZE", 0) + acc[k].get("WRITE_SIZE", 0)) * 1024
print("TRAFFIC_BYTES =", [round(tb("k_subplanes") + tb("k_copy_chroma_planes")), round(tb("k_me_fs_fast") + tb("k_me_fullsearch")), round(tb("k_me_refine_mb")),
                          round(sum(tb(k) for k in ("k_mb16_recon_luma", "k_mc_mb16", "k_tq_luma4x4", "k_tq_rec_to_plane", "k_mc_mb16_chroma", "k_tq_chroma", "k_tqc_rec_to_planes"))),
                          round(tb("k_deblock_prep") + tb("k_deblock_tasks") + tb("k_deblock_sparse") + tb("k_deblock_rows"))])
open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(out) + "\n")
shutil.copy(f"{O}/bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"{O}/stats/t_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
print("\n".join(out[4:14]))
