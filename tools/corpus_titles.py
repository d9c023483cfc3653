"""BASELINE configs[4] at SURVEY's size (29 titles x 600 streams) through dcs_pipeline, one title per list: seconds per pass for several
numbers of titles in flight, and where a title's time goes (DCS_PIPE_TRACE=1 on the last pass, or on pass TRACE_PASS: the pipeline's "pipe life" lines, averaged).
argv[1:]: titles in flight to try (default 4 8)"""
import sys, os, time, re, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import dcsexplorer_amd as D
    from dcsexplorer_amd import workloads as W
    depth = int(sys.argv[2])
    man = W.corpus_manifest(29, 600, 2000, 5)
    streams = W.corpus_streams(man)
    titles, start = [], 0
    for k in range(1, len(streams) + 1):
        if k == len(streams) or man[k]["title"] != man[start]["title"]:
            titles.append((start, k)); start = k
    lists = [D.make_refs(streams[a:b]) for a, b in titles]
    frames = sum(((s[1][0] << 8) | s[1][1]) for s in streams)
    ctx = D.Context(0)
    pipe = ctx.pipeline(depth, index_on_device=True, pack_on_device=True, plan_on_device=True)
    for p in range(3):
        if p == int(os.environ.get("TRACE_PASS", "2")):
            os.environ["DCS_PIPE_TRACE"] = "1"
        else:
            os.environ.pop("DCS_PIPE_TRACE", None)
        done = 0; t0 = time.perf_counter(); lat = []; sub = []
        for i, (refs, keep) in enumerate(lists):
            sub.append(time.perf_counter()); pipe.submit_refs(refs, titles[i][1] - titles[i][0])
            if i >= depth - 1:
                pipe.collect(); lat.append(time.perf_counter() - sub[done]); done += 1
        while done < len(lists):
            pipe.collect(); lat.append(time.perf_counter() - sub[done]); done += 1
        dt = time.perf_counter() - t0
        print("depth %d pass %d: %.3f s = %.3g samples/s; submit->collected per title: mean %.1f ms, max %.1f" % (
            depth, p, dt, frames * 240 / dt, sum(lat) / len(lat) * 1e3, max(lat) * 1e3), flush=True)
    pipe.close(); ctx.close()
    sys.exit(0)
for depth in (sys.argv[1:] or ["4", "8"]):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", depth], capture_output=True, text=True)
    print(r.stdout, end="")
    life = [list(map(float, re.findall(r"-?\d+\.\d+", l.split(":", 1)[1]))) for l in r.stderr.splitlines() if l.startswith("pipe life:")]
    up = [list(map(float, re.findall(r"-?\d+\.\d+", l.split(":", 1)[1]))) for l in r.stderr.splitlines() if l.startswith("pipe upload:")]
    if life:
        names = ["submit->taken", "upload", "wait for indexer", "index round", "wait for worker", "stage B"]
        print("  per title, ms (mean / max): " + ", ".join("%s %.1f / %.1f" % (n, sum(x[i] for x in life) / len(life), max(x[i] for x in life)) for i, n in enumerate(names)))
    if up:
        print("  upload, ms (mean): allocs %.2f, memcpy %.2f, hip calls %.2f" % tuple(sum(x[i] for x in up) / len(up) for i in range(3)))
    if r.returncode != 0:
        print(r.stderr[-2000:]); sys.exit(1)
