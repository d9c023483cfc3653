"""reads the 'pipe thread:' lines of a DCS_PIPE_TRACE=2 run (stdin) and prints, for the run's last 300 ms (or argv[1] ms), how
busy every pipeline thread was and with what"""
import sys, collections
ev = []
for line in sys.stdin:
    if not line.startswith("pipe thread:"):
        continue
    _, _, who, tid, what, t0, t1, q1, q2 = line.split()
    ev.append((who + tid, what, float(t0), float(t1), int(q1), int(q2)))
if not ev:
    sys.exit("no trace lines")
WIN = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
tend = max(e[3] for e in ev); tbeg = tend - WIN
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0])
for who, what, t0, t1, q1, q2 in ev:
    if t1 < tbeg: continue
    a = agg[(who, what)]; a[0] += 1; a[1] += t1 - max(t0, tbeg); a[2] += q1; a[3] += q2
for (who, what), (n, busy, q1, q2) in sorted(agg.items()):
    print("%-10s %-12s n=%4d busy %6.1f ms = %4.1f %%  avg %.2f ms %s" % (who, what, n, busy, busy / (WIN / 100.0), busy / n, ("lists/round %.1f streams/round %.0f" % (q1 / n, q2 / n)) if q1 else ""))
