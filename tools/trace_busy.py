"""reads a rocprofv3 --kernel-trace csv directory (argv[1]) of a pipeline run and prints, for the middle half of the run, how busy
each kind of kernel was (union of its intervals), its average duration and concurrency -- the PCM's copies down are the runtime's
__amd_rocclr_copyBuffer kernels of more than 100 us"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r["Kernel_Name"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    kind = ("index" if "dcsIndex" in n else "decode" if "dcsDecode" in n else "pack" if "dcsPack" in n else "plan" if "dcsPlan" in n else
            "clear" if "dcsClear" in n else "copy-kernel" if "dcsCopy" in n else "fill" if "fillBuffer" in n else
            ("D2H-pcm" if e - s > 100000 else "D2H-small") if "copyBuffer" in n else "other")
    ev.append((s, e, kind))
t0 = min(e[0] for e in ev); t1 = max(e[1] for e in ev)
lo, hi = t0 + (t1 - t0) // 4, t1 - (t1 - t0) // 4
by = collections.defaultdict(list)
for s, e, k in ev:
    if s >= lo and e <= hi:
        by[k].append((s, e))
span = hi - lo
npcm = len(by.get("D2H-pcm", []))
print("window %.1f ms, %d lists -> %.3f ms per list" % (span / 1e6, npcm, span / 1e6 / max(npcm, 1)))
for k, lst in sorted(by.items()):
    lst.sort(); busy = 0; cs, ce = lst[0]
    for s, e in lst[1:]:
        if s > ce: busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    tot = sum(e - s for s, e in lst)
    print("  %-11s n=%5d  busy %5.1f %%  avg %8.1f us  concurrency while busy %.2f" % (k, len(lst), 100.0 * busy / span, tot / len(lst) / 1e3, tot / busy))
