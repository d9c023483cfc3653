#!/usr/bin/env python3
"""bench.py -- throughput of the batched DCS frame decode on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dcs93_4096|dcs94_65536|mixed_16384]

One "step" = one pass of the hot path (one kernel launch) over one resident batch of synthetic frames.
Metric (BASELINE.json): bit-exact int16 PCM samples/s; value = samples of all ranks / max-over-ranks time.
Multi-GPU: one process per GPU (torch.distributed.run), each rank decodes its own range of the stream
corpus (weak scaling, no data-path collective: frames of different streams are independent).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(streams, budget_s=12.0):
    """the CPU checker timed on this box's host cores on a bounded sample of the same workload: the
    compiled reference (oracle/_ref) when it travelled with the repo, else the oracle port.  The
    threads live inside the C library (no Python in the timed loop); each thread plays its share of the
    streams through one decoder object, LoadAudioStream + 240 x GetNextSample per frame, the way the
    reference's own batch decode (--extract-streams) drives the path."""
    from oracle.dcs_oracle import Oracle, Reference, reference_available
    kind = "reference" if reference_available() else "port"
    chk = Reference() if kind == "reference" else Oracle()
    cores = max(1, min(os.cpu_count() or 1, 64))
    frames_per_pass = sum((s[1][0] << 8) | s[1][1] for s in streams)
    # calibrate one pass, then size the repeat count for the budget
    t0 = time.perf_counter()
    chk.decode_many(streams, 1, cores)
    one = max(time.perf_counter() - t0, 1e-4)
    repeat = max(1, int(budget_s / one))
    t0 = time.perf_counter()
    frames = chk.decode_many(streams, repeat, cores)
    dt = time.perf_counter() - t0
    return dict(value=frames * 240 / dt, unit="samples/s", cores=cores, kind=kind,
                sample="the %d streams (%d frames) of the workload decoded %d times in %.1f s on %d threads, "
                       "one decoder object per thread" % (len(streams), frames_per_pass, repeat, dt, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="dcs93_4096")
    ap.add_argument("--fpw", type=int, default=0, help="frames per wavefront override (8/16/32/64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=1, help="batches in flight: the K steps are dealt to this many batch objects "
                    "(same workload), each on a stream of its own, so that launches of different batches overlap on the GPU "
                    "(1 = the contract's back-to-back steps; more is reported as config.inflight, never the default)")
    ap.add_argument("--scale", type=int, default=1, help="decode SCALE times the workload's streams per step (further "
                    "seeds of the same recipe): the large-batch rate; not a BASELINE config")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import dcsexplorer_amd as D
    from dcsexplorer_amd import sharding, workloads

    rank, local_rank, world = sharding.rank_info()

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    # this rank's range of the corpus: same shape on every rank, different streams (seeds)
    streams = sharding.rank_streams(args.workload, rank)
    if args.scale > 1:
        import inspect
        fn = workloads.WORKLOADS[args.workload]
        n = inspect.signature(fn).parameters["n_streams"].default
        streams = fn(n_streams=n * args.scale * world)[rank * n * args.scale:(rank + 1) * n * args.scale]
    b = D.build_stream_batch(streams)
    if args.workload == "mixed_16384":
        b, _ = workloads.interleave(b)
    n_frames = int(b["jobs"].size)

    ctx = D.Context(local_rank)
    if args.fpw:
        ctx.set_frames_per_wave(args.fpw)
    batch = ctx.batch(b["blob"], b["srcs"], b["jobs"])
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()

    # further batches of the same workload on streams of their own (--inflight): a step is still one launch over one batch
    extra = [(ctx.batch(b["blob"], b["srcs"], b["jobs"]), torch.cuda.Stream()) for _ in range(max(0, args.inflight - 1))]
    lanes = [(batch, stream)] + [(bt, st.cuda_stream) for bt, st in extra]
    share = [args.steps // len(lanes) + (1 if k < args.steps % len(lanes) else 0) for k in range(len(lanes))]

    for _ in range(args.warmup):
        for bt, st in lanes:
            bt.run(st)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (bt, st), k in zip(lanes, share):
        if k:
            bt.run_many(k, st)              # K launches back to back (one step = one launch), issued from C
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    dt = sharding.max_over_ranks(dt, device="cuda")

    # kernel-only average duration by HIP events on the launch stream (roofline denominator)
    kern_ms = batch.time(max(10, args.steps), stream)
    algo_bytes = batch.algorithmic_bytes

    # bit-exactness of what was just timed (rank 0, default corpus range): per-stream hashes vs the
    # reference's committed hashes
    bit_exact = None
    if rank == 0:
        from oracle.dcs_oracle import fnv1a64
        pcm, err = batch.download()
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "dcs_golden_hashes.json")))["workloads"]
        if args.workload in gold and args.scale == 1:
            if args.workload == "mixed_16384":
                _, perm = workloads.interleave(D.build_stream_batch(streams))
                inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
                pcm = pcm[inv]
            first = D.build_stream_batch(streams)["first_job"] if args.workload == "mixed_16384" else b["first_job"]
            got = ["%016x" % fnv1a64(pcm[first[k]:first[k + 1]].tobytes()) for k in range(len(first) - 1)]
            bit_exact = bool(got == gold[args.workload]["stream_hashes"]) and not bool(err.any())

    if rank == 0:
        samples = n_frames * 240 * world * args.steps
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        # HBM traffic per launch: FETCH_SIZE + WRITE_SIZE of the rocprofv3 --pmc passes of this same command
        # (tools/prof.sh), committed under profiles/; null when no profile of this workload exists
        traffic, traffic_note, valu = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath) and args.scale == 1 and not args.fpw:       # (the committed counters are those of the plain workload)
            t = json.load(open(tpath))
            # the kernel's reads are 16-byte-per-lane loads of the chunk packages, for which FETCH_SIZE reports half
            # the bytes on gfx950 (MI355X_MICROARCH.md, HBM section): doubled here; WRITE_SIZE is exact
            traffic = t["traffic_bytes_fetch_x2"]
            traffic_note = ("2 x FETCH_SIZE + WRITE_SIZE from %s (separate --pmc passes; kilobytes; the gfx950 wide-read "
                            "correction applied to FETCH_SIZE; uncorrected sum: %d)"
                            % (os.path.relpath(tpath, ROOT), t["traffic_bytes_fetch_raw"]))
            if "SQ_INSTS_VALU" in t and "GRBM_GUI_ACTIVE" in t:
                # what actually bounds this integer kernel: wave64 VALU issue, 4 cycles per instruction per SIMD.
                # Counters are from the committed profile of this workload; the cycle count is GRBM_GUI_ACTIVE
                # (summed over the 8 XCDs) of the same profile.
                simds = 256 * 4
                cycles = t["GRBM_GUI_ACTIVE"] / 8.0
                valu = {"valu_insts_per_launch": t["SQ_INSTS_VALU"], "simds": simds, "issue_cycles_per_inst": 4,
                        "kernel_cycles": cycles, "frac_of_valu_issue_peak": t["SQ_INSTS_VALU"] * 4 / (simds * cycles),
                        "lds_bank_conflict_cycles": t.get("SQ_LDS_BANK_CONFLICT"),
                        "source": os.path.relpath(tpath, ROOT)}
        out = {
            "metric": "bit_exact_int16_pcm_samples_per_sec",
            "value": samples / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {"workload": args.workload, "frames_per_gpu_per_step": n_frames,
                       "samples_per_frame": 240, "frames_per_wave": args.fpw or "auto",
                       "arithmetic": "ADSP-2105 1.15 fixed point, 32-bit integer intermediates",
                       "partition": "range over streams, no collective", "scale": args.scale, "inflight": args.inflight},
            "bit_exact": bit_exact,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "dcsDecodeKernel", "kernel_avg_ms": kern_ms,
                         "algorithmic_bytes_per_launch": algo_bytes, "valu_issue": valu},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(streams)
        print(json.dumps(out))

    for bt, _ in extra:
        bt.close()
    batch.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
