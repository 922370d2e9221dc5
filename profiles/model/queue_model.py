"""Discrete-event model of the one-launch queue (jmhip_seq_batch): how `value` moves with the number of workgroup slots, the per-macroblock times and the
lag between pictures.  A planning aid (round 6): numbers are calibrated against profiles/r05_final_bench_20.json (78.5 ms for 20 pictures)."""
import heapq, sys

def run(npic=20, wmb=120, hmb=68, slots=256, lag=16, reach=(5, 5), t_stage=6.4, t_chain=85.0, t_tail=10.0, t_post=10.0, t_draw=2.0):
    nkeys = wmb + 2 * (hmb - 1)
    # tickets in key order (equal keys: earlier picture first)
    tickets = []
    for p in range(npic):
        for y in range(hmb):
            for x in range(wmb):
                tickets.append((p * lag + x + 2 * y, p, y, x))
    tickets.sort()
    vec = {}
    post = {}
    free = [0.0] * slots
    heapq.heapify(free)
    tmax = 0.0
    for key, p, y, x in tickets:
        d = heapq.heappop(free) + t_draw
        ref_ready = 0.0
        if p > 0:
            ref_ready = post[(p - 1, min(hmb - 1, y + reach[1]), min(wmb - 1, x + reach[0]))]
        st_end = max(d, ref_ready) + t_stage
        nb = 0.0
        if x > 0: nb = max(nb, vec[(p, y, x - 1)])
        if y > 0:
            nb = max(nb, vec[(p, y - 1, x)])
            if x < wmb - 1: nb = max(nb, vec[(p, y - 1, x + 1)])
        s = max(st_end, nb)
        v = s + t_chain
        vec[(p, y, x)] = v
        end = v + t_tail
        pr = end
        if x > 0: pr = max(pr, post[(p, y, x - 1)])
        if y > 0 and x < wmb - 1: pr = max(pr, post[(p, y - 1, x + 1)])
        post[(p, y, x)] = pr + t_post
        heapq.heappush(free, end)
        tmax = max(tmax, pr + t_post)
    return tmax

if __name__ == "__main__":
    n = wmb = 0
    base = run()
    print("baseline model: %.1f ms  -> %.2f M MB/s" % (base / 1e3, 8160 * 20 / base))
    for name, kw in [
        ("chain 70", dict(t_chain=70)),
        ("chain 60", dict(t_chain=60)),
        ("slots 512 chain 95", dict(slots=512, t_chain=95)),
        ("slots 512 chain 85", dict(slots=512, t_chain=85)),
        ("slots 512 chain 70", dict(slots=512, t_chain=70)),
        ("lag 10", dict(lag=10, reach=(3, 3))),
        ("lag 10 chain 70", dict(lag=10, reach=(3, 3), t_chain=70)),
        ("lag 10 slots 512 chain 95", dict(lag=10, reach=(3, 3), slots=512, t_chain=95)),
        ("lag 10 slots 512 chain 85", dict(lag=10, reach=(3, 3), slots=512, t_chain=85)),
        ("lag 10 slots 512 chain 70", dict(lag=10, reach=(3, 3), slots=512, t_chain=70)),
        ("lag 13 slots 512 chain 85", dict(lag=13, reach=(4, 4), slots=512, t_chain=85)),
        ("slots 384 chain 85", dict(slots=384, t_chain=85)),
        ("slots 256 chain 85 tail 2 (two-MB pipeline)", dict(t_tail=2.0, t_stage=0.5)),
        ("40 pictures baseline", dict(npic=40)),
    ]:
        t = run(**kw)
        npic = kw.get("npic", 20)
        print("%-45s %.1f ms -> %.2f M MB/s" % (name, t / 1e3, 8160 * npic / t))
