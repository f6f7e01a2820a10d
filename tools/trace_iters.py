"""Device-side view of the last N training iterations in a rocprofv3 --kernel-trace CSV of a trainer run (iterations are
delimited by hiast::adam_kernel launches): span per iteration, time with no / one / several kernels running, the largest
kernels.   python tools/trace_iters.py <kernel_trace.csv> [N]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    assert len(marks) > n
    sel = rows[marks[-n - 1] + 1:marks[-1] + 1]
    t0 = int(sel[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in sel)
    ev = []
    busy = defaultdict(lambda: [0, 0])
    for i, r in enumerate(sel):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, 1)); ev.append((e, -1))
        k = r["Kernel_Name"].split("(")[0][:80]
        busy[k][0] += 1
        busy[k][1] += e - s
    ev.sort()
    live = 0; last = t0; idle = one = multi = 0
    gaps = []
    for t, d in ev:
        dt = t - last
        if dt > 0:
            if live == 0:
                idle += dt
                gaps.append(dt)
            elif live == 1:
                one += dt
            else:
                multi += dt
        last = t
        live += d
    ms = lambda x: x / n / 1e6
    print("last %d iterations: %.2f ms/iteration on the device clock; no kernel running %.2f ms, one %.2f, several %.2f; %d dispatches/iteration"
          % (n, ms(t1 - t0), ms(idle), ms(one), ms(multi), len(sel) / n))
    gaps.sort(reverse=True)
    print("largest gaps (us):", [round(g / 1e3) for g in gaps[:12]], "| gaps > 100 us per iteration: %.1f, their sum %.2f ms/iteration"
          % (sum(1 for g in gaps if g > 1e5) / n, sum(g for g in gaps if g > 1e5) / n / 1e6))
    tot = sum(v[1] for v in busy.values())
    print("kernel time %.2f ms/iteration" % ms(tot))
    for k, (c, d) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %7.3f ms  %6.1f calls  avg %7.1f us  %s" % (ms(d), c / n, d / c / 1e3, k))


if __name__ == "__main__":
    main()
