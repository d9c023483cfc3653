#!/usr/bin/env python3
"""Samples per second through the decoder CLASS (GetNextSample in a bare loop, as every caller of the reference pulls it:
DCSDecoder.cpp:1632-1638, DCSEncoder.cpp:565-567, EncoderTester.cpp:94-100, DCSExplorer.cpp:1709-1711), for the three
builds of tests/cpp/dcs_pump_bench.cpp:

  hip-mirror    dcsexplorer_amd/dcs_pump_bench          DCSDecoderHIP, stand-alone build of the class
  hip-refbase   oracle/_ref/dcs_pump_bench_refbase      DCSDecoderHIP behind the reference's real ::DCSDecoder
  native        oracle/_ref/dcs_pump_bench_native       the reference's DCSDecoderNative (the CPU pump; checker and baseline)

and three scenarios: the ROM-less recipe (one 2 000-frame stream per layout), the --extract-streams loop (64 streams on one
decoder), a ROM-mode multi-channel script.  bench.py's class_surface section calls run_all(); run by hand it prints a table.
The native build is the reference: its PCM hash is what the HIP builds' hashes are compared with."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BUILDS = {
    "hip-mirror": os.path.join(ROOT, "dcsexplorer_amd", "dcs_pump_bench"),
    "hip-refbase": os.path.join(ROOT, "oracle", "_ref", "dcs_pump_bench_refbase"),
    "native": os.path.join(ROOT, "oracle", "_ref", "dcs_pump_bench_native"),
}
RECIPE_FRAMES = 2000
# (layout, os) of the six recipe streams; volume 255, level 0x64 (the levels track programs use, DCSDecoderNative.h:88-93)
RECIPES = [("93-T0", 0, 1), ("93b-T1", 1, 1), ("93a-T1", 2, 0), ("94-T0", 3, 2), ("94-T1s0", 4, 2), ("94-T1s3", 5, 3)]


def _run(exe, args, timeout):
    p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    if p.returncode != 0:
        return {"error": "exit %d: %s" % (p.returncode, p.stderr.strip()[-300:])}
    return json.loads(p.stdout.strip().splitlines()[-1])


def _rate(res):
    """samples/s of the median repetition, the first one (which creates the context) left out when there are several"""
    if "error" in res:
        return res
    play = sorted(res["play_ms"][1:] or res["play_ms"])
    med = play[len(play) // 2]
    return {"samples_per_s": res["samples"] / (med * 1e-3), "play_ms": med, "first_play_ms": res["play_ms"][0],
            "boot_ms": sorted(res["boot_ms"])[len(res["boot_ms"]) // 2], "first_boot_ms": res["boot_ms"][0],
            "frames": res["frames"], "us_per_frame": med * 1e3 / res["frames"], "fnv1a64": res["fnv1a64"], "decoder": res["decoder"]}


def make_inputs(tmp):
    import dcsexplorer_amd as D
    from util import make_stream, splitmix
    import romkit
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_seq_golden as G
    inp = {"recipe": [], "extract": {}, "script": {}}
    for name, fmt, os_ in RECIPES:
        path = os.path.join(tmp, "recipe_%s.bin" % name)
        open(path, "wb").write(make_stream(fmt, RECIPE_FRAMES, seed=0x600 + fmt, profile=5 if fmt >= 3 else 6 if fmt < 2 else 0,
                                           nbands=12 if fmt != D.FMT_93A_T1 else None))
        inp["recipe"].append((name, os_, path))
    # the extract loop: 64 streams of one OS (1994+, the three layouts), 20..600 frames
    g = splitmix(0x6E)
    paths = []
    for i in range(64):
        n = 20 + next(g) % 581
        path = os.path.join(tmp, "ex_%02d.bin" % i)
        open(path, "wb").write(make_stream(3 + i % 3, n, seed=0x6E00 + i, profile=5, nbands=12))
        paths.append(path)
    inp["extract"] = {"os": 3, "paths": paths}
    # ROM mode: the sequencer ROM set of the tests (OS95), tracks on six channels, looping streams; commands through the data
    # port in the first 60 ticks, a volume change and two more tracks later, quiet in between
    rs = G.build(G.CASES[3])
    roms = []
    for chip, image in sorted(rs.images.items()):
        path = os.path.join(tmp, "u%d.rom" % chip)
        open(path, "wb").write(image)
        roms.append("%d=%s" % (chip, path))
    pc = romkit.port_cmd
    ev = pc(0, 1) + pc(3, 2) + pc(12, 5) + pc(30, 6) + pc(40, 7) + pc(60, 16) \
        + [(500, 0, 0x55), (500, 0, 0xAA), (500, 0, 0xB0), (500, 0, 0x4F)] + pc(900, 12) + pc(1400, 2) + [(1700, 2, 0xE0)]
    evf = os.path.join(tmp, "events.txt")
    open(evf, "w").write("".join("%d %d %d\n" % e for e in sorted(ev, key=lambda x: x[0])))
    inp["script"] = {"roms": roms, "events": evf, "ticks": 2000, "volume": G.VOLUME}
    return inp


def run_all(builds=("hip-mirror", "hip-refbase", "native"), lookaheads=(-1, 1), reps=5, budget_s=200.0, log=None):
    """{scenario: {build[@lookahead]: rate}}; lookahead -1 = the decoder's default, 1 = tick by tick (HIP builds only)"""
    import time
    t_end = time.time() + budget_s
    out = {"recipe": {}, "extract": {}, "script": {}, "one_shot": None, "notes": []}
    with tempfile.TemporaryDirectory(prefix="dcs_pump_") as tmp:
        inp = make_inputs(tmp)
        for b in builds:
            exe = BUILDS[b]
            if not os.path.exists(exe):
                out["notes"].append("%s not built (%s)" % (b, os.path.relpath(exe, ROOT)))
                continue
            for la in (lookaheads if b != "native" else (-1,)):
                key = b if la < 0 else "%s@lookahead%d" % (b, la)
                r = reps if la != 1 else 2                 # (tick by tick is slow: two repetitions)
                left = lambda: max(5.0, t_end - time.time())
                if time.time() > t_end:
                    out["notes"].append("time budget spent before %s" % key)
                    continue
                for name, os_, path in inp["recipe"]:
                    res = _rate(_run(exe, ["recipe", os_, 255, 0x64, la, r, "-", path], left()))
                    out["recipe"].setdefault(name, {})[key] = res
                    if log: log("recipe %s %s: %s" % (name, key, res.get("samples_per_s", res)))
                res = _rate(_run(exe, ["extract", inp["extract"]["os"], 255, 0x64, la, r, "-"] + inp["extract"]["paths"], left()))
                out["extract"][key] = res
                if log: log("extract %s: %s" % (key, res.get("samples_per_s", res)))
                s = inp["script"]
                res = _rate(_run(exe, ["script", s["volume"], la, r, "-", s["ticks"], s["events"]] + s["roms"], left()))
                out["script"][key] = res
                if log: log("script %s: %s" % (key, res.get("samples_per_s", res)))
            if b == "hip-mirror" and time.time() < t_end:
                # the one-shot C ABI underneath the class: microseconds per dcs_decode_batch call, next to the box's floor
                name, os_, path = inp["recipe"][-1]
                res = _run(exe, ["oneshot", os_, 255, 0x64, 300, path, 1, 8, 64, 512], max(5.0, t_end - time.time()))
                out["one_shot"] = res.get("calls", res)
                if log: log("one_shot: %s" % (out["one_shot"],))
    # the reference's PCM is the native build's: every other build and look-ahead must hash the same
    def check(d):
        want = d.get("native", {}).get("fnv1a64")
        hashes = {k: v.get("fnv1a64") for k, v in d.items() if isinstance(v, dict) and "fnv1a64" in v}
        return {"reference_hash": want, "all_equal": len(set(hashes.values())) == 1 and bool(hashes), "compared": sorted(hashes)}
    out["bit_exact"] = {"recipe": {n: check(d) for n, d in out["recipe"].items()}, "extract": check(out["extract"]), "script": check(out["script"])}
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--builds", default="hip-mirror,hip-refbase,native")
    ap.add_argument("--lookaheads", default="-1,1")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = run_all(tuple(a.builds.split(",")), tuple(int(x) for x in a.lookaheads.split(",")), a.reps, budget_s=900,
                  log=lambda m: sys.stderr.write(m + "\n"))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
    def row(name, d):
        for k, v in sorted(d.items()):
            if "error" in v:
                print("%-28s %-28s %s" % (name, k, v["error"]))
            else:
                print("%-28s %-28s %10.3e samples/s  %8.3f us/frame  play %9.3f ms (first %9.3f)  boot %8.3f ms  %s"
                      % (name, k, v["samples_per_s"], v["us_per_frame"], v["play_ms"], v["first_play_ms"], v["boot_ms"], v["fnv1a64"]))
    for n, d in res["recipe"].items():
        row("recipe " + n, d)
    row("extract (64 streams)", res["extract"])
    row("script (2000 ticks, ROM mode)", res["script"])
    for c in (res.get("one_shot") or []) if isinstance(res.get("one_shot"), list) else []:
        print("one-shot dcs_decode_batch, %4d frames: %7.2f us per call (min %.2f); floor on this box: launch + wait %.2f us, with the copy down %.2f us"
              % (c["frames"], c["us_per_call"], c["us_min"], c["floor_launch_wait_us"], c["floor_launch_copy_wait_us"]))
    print(json.dumps(res["bit_exact"]))
    for n in res["notes"]:
        print("note:", n)
