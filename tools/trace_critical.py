"""Where the wall time of bench.py's timed region goes, from a rocprofv3 --kernel-trace CSV (between the two
hiast::confusion_kernel markers): per-queue busy time, time with NO kernel running anywhere (launch / host gaps), time
with exactly one / more than one kernel running, and the kernels that run alone (= the critical path when streams
overlap).
    python tools/trace_critical.py <kernel_trace.csv> <steps>"""
import csv
import sys
from collections import defaultdict


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "confusion_kernel" in r["Kernel_Name"]]
    assert len(marks) >= 2, "markers not found"
    sel = rows[marks[0] + 1:marks[1]]
    qkey = "Queue_Id" if "Queue_Id" in sel[0] else None
    t0 = int(sel[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in sel)
    span = t1 - t0
    per_q = defaultdict(lambda: [0, 0])
    ev = []
    for i, r in enumerate(sel):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r[qkey] if qkey else "0"
        per_q[q][0] += 1
        per_q[q][1] += e - s
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    live = set()
    last = t0
    idle = one = multi = 0
    alone = defaultdict(int)         # kernel name -> ns during which it was the ONLY kernel running
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            if not live:
                idle += dt
            elif len(live) == 1:
                one += dt
                alone[sel[next(iter(live))]["Kernel_Name"]] += dt
            else:
                multi += dt
        last = t
        if d == 1:
            live.add(i)
        else:
            live.discard(i)
    ms = lambda x: x / steps / 1e6
    print("timed region: %.3f ms/step over %d steps, %d dispatches/step" % (ms(span), steps, len(sel) / steps))
    print("no kernel running: %.3f ms/step (%.1f %%) | exactly one: %.3f | two or more: %.3f"
          % (ms(idle), 100.0 * idle / span, ms(one), ms(multi)))
    for q, (n, d) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
        print("queue %-6s %7.1f dispatches/step  busy %.3f ms/step" % (q, n / steps, ms(d)))
    print("kernels running ALONE (ms/step):")
    for k, d in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
        print("  %8.3f  %s" % (ms(d), k[:140]))
    # the longest gaps with nothing running
    gaps = []
    live_n = 0
    last = t0
    prev = None
    for t, d, i in ev:
        if live_n == 0 and t - last > 0 and prev is not None:
            gaps.append((t - last, sel[prev]["Kernel_Name"][:60], sel[i]["Kernel_Name"][:60]))
        live_n += d
        last = t
        prev = i
    gaps.sort(reverse=True)
    print("longest idle gaps (us): after -> before")
    for g, a, b in gaps[:15]:
        print("  %8.1f  %s -> %s" % (g / 1e3, a, b))
    hist = defaultdict(int)
    for g, _, _ in gaps:
        hist[min(int(g / 1e3) // 5 * 5, 100)] += g
    print("idle time by gap length (us bucket: ms/step):", {k: round(ms(v), 3) for k, v in sorted(hist.items())})


if __name__ == "__main__":
    main()
