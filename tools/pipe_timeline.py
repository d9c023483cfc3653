"""reads a rocprofv3 --kernel-trace --memory-copy-trace output directory (csv) of tools/pipe_trace.py and says how busy the
link and the kernels were over the run's last second: argv[1] = directory"""
import csv, glob, sys, collections
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        kind = "index" if "dcsIndex" in name else "decode" if "dcsDecode" in name else "pack" if "dcsPack" in name else "other-kernel"
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, 0))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dirn = r.get("Direction", "")
        kind = "D2H" if "DEVICE_TO_HOST" in dirn else "H2D" if "HOST_TO_DEVICE" in dirn else "copy-" + dirn
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, int(r.get("Bytes", 0) or 0)))
ev.sort()
t1 = max(e[1] for e in ev); t0 = t1 - int(0.4e9)
win = [e for e in ev if e[0] >= t0]
span = (t1 - t0) / 1e6
print("window %.1f ms, %d events" % (span, len(win)))
by = collections.defaultdict(list)
for s, e, k, b in win:
    by[k].append((s, e, b))
for k, lst in sorted(by.items()):
    busy = 0; cur_s, cur_e = None, None
    for s, e, b in sorted(lst):
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, b in lst); nb = sum(b for s, e, b in lst)
    print("  %-14s n=%5d  busy (union) %6.1f ms = %4.1f %%   sum %7.1f ms  avg %7.3f ms  %s" %
          (k, len(lst), busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6, tot / 1e6 / len(lst), ("%.1f MB, %.1f GB/s while busy" % (nb / 1e6, nb / max(busy, 1))) if nb else ""))
