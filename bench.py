#!/usr/bin/env python3
"""bench.py -- LFA strips analysed per second (detect + profile + peaks) on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs here.  N > 1 without a launcher's WORLD_SIZE: starts `python -m torch.distributed.run --nnodes=1
        --nproc-per-node N --master-addr 127.0.0.1 ...` on itself as a child -- before anything touches the GPU -- and
        relays its output.  Under a launcher (RANK / LOCAL_RANK / WORLD_SIZE set) WORLD_SIZE must equal --gpus.

Headline workload (BASELINE.json configs[1]): a batch of 1024 synthetic 2048x1536 colour images, 8 strips each, per GPU,
resident in HBM before the timed region.  A step = one pass of lfa_analyze_batch over that batch (in chunks of --chunk
images alternating over --streams handles).  The batch is a pool of --pool distinct seeded images
(image_analysis_amd.synth.synth_rgb) tiled on the device.  Multi-GPU: every rank owns its own batch (weak scaling, no
data-path collective); after the timed region the per-strip result records of all ranks are gathered once over the
collective backend (RCCL; `shard.gather_strip_records`), as north_star's "final result gather".

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel of the step (found in warm-up with every
kernel timed, then timed alone with HIP events inside the timed region); its algorithmic bytes are the image bytes one
launch reads plus what it must write.  `cpu_baseline` is the oracle (numpy/scipy restatement of the reference
arithmetic) on a bounded sample of the same images on the host.  `configs` holds the other BASELINE.json configurations:
at N = 1 configs[3] (3072x4096 images, with its own record gather), configs[2], the reference's own 16-bit frame,
configs[4] and the Gaussian stage, measured by `benchmarks_configs.py all` in a child process that finishes before this
one initialises the GPU; at N > 1 configs[3] on every rank (weak scaling) with its record gather over RCCL.
`sustained_hbm` prices the whole step in HBM bytes actually moved (PMC counters per kernel, profiles/traffic.json).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per step")
    ap.add_argument("--chunk", type=int, default=256, help="images per lfa_analyze_batch call")
    ap.add_argument("--pool", type=int, default=16, help="distinct synthetic images tiled to the batch")
    ap.add_argument("--height", type=int, default=1536)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=64, help="images for the one-core CPU baseline (0 = skip); the pool run takes 8x as many")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the CPU pool baseline (0 = min(16, cpu_count))")
    ap.add_argument("--streams", type=int, default=3, help="lanes (one LfaHandle + one HIP stream each) the chunks take in turn")
    ap.add_argument("--stagger", type=int, default=2, help="0: lanes run free; 1: a chunk's kernels start when the previous chunk's histogram pass is done (lfa_stream_wait_passes); 2: only its image passes wait (lfa_chain_passes)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-every", type=int, default=4,
                    help="in the timed region every N-th launch of the dominant kernel (per lane) is bracketed by HIP events; "
                         "1 = every launch (the records between the kernels then cost about 2 %% of the step)")
    ap.add_argument("--free-running", action="store_true", help="after the timed region, time the same steps again with no ordering between the lanes (reported as `free_running`); off by default so that a profile of the default command holds one schedule only")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary configurations (the `configs` object)")
    ap.add_argument("--config-steps", type=int, default=10, help="timed steps of each secondary configuration")
    ap.add_argument("--config-batch", type=int, default=128, help="N > 1: images per GPU of configs[3] (3072x4096x3)")
    ap.add_argument("--config-chunk", type=int, default=64, help="N > 1: images per call of configs[3]")
    return ap.parse_args()


def merge_times(dicts):
    out = {}
    for d in dicts:
        for k, (ms, n) in d.items():
            a = out.get(k, (0.0, 0))
            out[k] = (a[0] + ms, a[1] + n)
    return out


def cpu_baseline(args, pool_run=True):
    """The oracle on a bounded sample of the workload, in a child process of its own (oracle/cpu_baseline.py), run to
    completion BEFORE this process initialises the GPU: one core, then (pool_run) a pool of processes."""
    procs = args.cpu_procs or min(16, os.cpu_count() or 1)
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--height", str(args.height), "--width", str(args.width),
           "--pool", str(args.pool), "--seed0", "1234", "--single", str(args.cpu_sample),
           "--parallel", str(args.cpu_sample * 8 if pool_run else 0), "--procs", str(procs)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-2000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    one, par = d["single"], d.get("parallel")
    best = par if par else one
    cpu = {"value": round(best["strips"] / best["seconds"], 2), "unit": "strips/s", "cores": par["procs"] if par else 1, "kind": "port",
           "sample": f"oracle/lfa_oracle.analyze_image on {best['images']} images of the same workload ({args.pool} distinct) in "
                     f"{best['seconds']:.1f} s" + (f" on a pool of {par['procs']} processes" if par else " on one core") +
                     "; the reference additionally runs canny+Hough (dead compute), so this is a lower bound on its time",
           "images_per_s": round(best["images"] / best["seconds"], 3),
           "one_core": {"value": round(one["strips"] / one["seconds"], 2), "images": one["images"], "seconds": round(one["seconds"], 2)}}
    return cpu


def self_launch(args):
    """--gpus N > 1 without a launcher: become the launcher.  Nothing in this process has touched the GPU yet."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def secondary_child(args):
    """The other configurations, each in a process of its own (`benchmarks_configs.py <name> --json`) that has the GPU to
    itself and has finished before this process initialises the device: measured in one process after another
    configuration, configs[2] and configs[3] came out 10-15 % lower (placement of their buffers among what the earlier
    run had held) -- alone they repeat within 1 %."""
    out = {}
    for name in ("big", "u16", "u16real", "photo", "rois", "gauss", "lines"):
        cmd = [sys.executable, os.path.join(ROOT, "benchmarks_configs.py"), name, "--json", "--steps", str(args.config_steps)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(f"secondary configuration {name} failed: " + r.stderr[-3000:])
        out.update(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    return out


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        return self_launch(args)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree\n")
        return 2
    # the oracle on the host, before this process initialises the GPU: whole sample at N = 1; at N > 1 rank 0 times the
    # one-core sample only (a 16-process pool would compete with the other ranks' host threads)
    cpu = None
    if args.cpu_sample > 0 and rank == 0:
        cpu = cpu_baseline(args, pool_run=(world == 1))
    configs = None
    if world == 1 and not args.no_configs:
        configs = secondary_child(args)

    import torch
    import torch.distributed as dist
    # LFA_BENCH_REHEARSAL=1 (tests only): ranks share the GPUs there are and talk over gloo, so that the multi-rank code
    # path can be exercised on a one-GPU box; the real run is one rank per GPU over RCCL ("nccl" IS RCCL on ROCm)
    rehearsal = os.environ.get("LFA_BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = torch.device("cpu") if rehearsal else dev          # where the collective operands live

    capi = importlib.import_module("image_analysis_amd._capi")
    synth = importlib.import_module("image_analysis_amd.synth")
    shard = importlib.import_module("image_analysis_amd.shard")

    H, W, C = args.height, args.width, 3
    B, CH = args.batch, min(args.chunk, args.batch)
    mm_c = int(W / 40)
    mm_r = int(mm_c * 20)
    max_strips = 16
    stride = int(H * .6) + 1
    img_bytes = H * W * C

    # ---- inputs: pool of distinct images (rank-dependent seeds), tiled to the batch on the device ----
    pool_np = np.stack([synth.synth_rgb(H, W, 8, 1234 + rank * args.pool + i) for i in range(args.pool)])
    pool = torch.from_numpy(pool_np.reshape(args.pool, -1)).to(dev)
    idx = torch.arange(B, device=dev) % args.pool
    batch = pool[idx].contiguous()                       # (B, H*W*3) uint8, resident in HBM
    del pool
    assert batch.data_ptr() % 16 == 0

    # one handle + one stream per lane of the chunk pipeline: the latency-bound tail kernels of one chunk overlap with
    # the streaming kernels of the next (per-device streams, no cross-stream dependency: chunks are independent)
    NS = max(1, min(args.streams, (B + CH - 1) // CH))
    engs = [capi.Engine(device=dev_index, max_h=H, max_w=W, max_c=3, max_batch=CH) for _ in range(NS)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    if NS > 1 and args.stagger == 2:                     # standing rule: lane i's image passes follow lane i-1's
        for i in range(NS):
            capi._check(capi.lib.lfa_chain_passes(engs[i]._h, engs[i - 1]._h))
    strips = capi.dev_bytes(B * max_strips * capi.STRIP_DT.itemsize, dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    info = capi.dev_bytes(B * capi.INFO_DT.itemsize, dev)
    rois = capi.dev_bytes(B * max_strips * capi.ROI_DT.itemsize, dev)
    profiles = torch.empty(B * max_strips * stride, dtype=torch.float64, device=dev)
    peaks = capi.dev_bytes(B * max_strips * capi.PEAKS_DT.itemsize, dev)
    import ctypes

    lane = [0]                                           # chunks take the lanes in turn ACROSS steps too (a lane = handle + stream)

    def step(ns=NS):
        for b0 in range(0, B, CH):
            ci = lane[0]
            lane[0] += 1
            nb = min(CH, B - b0)
            e = engs[ci % ns]
            stream = ctypes.c_void_p(streams[ci % ns].cuda_stream)
            if ns > 1 and args.stagger == 1:
                capi._check(capi.lib.lfa_stream_wait_passes(engs[(ci - 1) % ns]._h, stream))
            capi._check(capi.lib.lfa_analyze_batch(
                e._h, batch.data_ptr() + b0 * img_bytes, capi.LFA_U8, nb, H, W, C, mm_r, mm_c, 10,
                strips.data_ptr() + b0 * max_strips * capi.STRIP_DT.itemsize, max_strips,
                counts.data_ptr() + b0 * 4, info.data_ptr() + b0 * capi.INFO_DT.itemsize,
                rois.data_ptr() + b0 * max_strips * capi.ROI_DT.itemsize,
                profiles.data_ptr() + b0 * max_strips * stride * 8, stride,
                peaks.data_ptr() + b0 * max_strips * capi.PEAKS_DT.itemsize, stream))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def sum_over_ranks(value):
        if world == 1:
            return int(value)
        tt = torch.tensor([int(value)], dtype=torch.int64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        return int(tt.item())

    def gather_records(strips_t, counts_t, peaks_t, ms, n_local):
        """every rank ends with the records of all world*n_local images in image order; returns (records, seconds)"""
        torch.cuda.synchronize()
        cn = counts_t.cpu().numpy()
        rec = shard.pack_strip_records(capi.from_device(strips_t, capi.STRIP_DT, n_local * ms).reshape(n_local, ms),
                                       capi.from_device(peaks_t, capi.PEAKS_DT, n_local * ms).reshape(n_local, ms), cn, rank * n_local)
        barrier()
        t0 = time.perf_counter()
        all_counts, all_rec = shard.gather_strip_records(cn, rec, world * n_local, device=cdev)
        barrier()
        return all_counts, all_rec, max_over_ranks(time.perf_counter() - t0)

    torch.cuda.synchronize()                             # inputs are resident before any chunk stream starts

    # ---- warm-up; with every kernel timed, to find the dominant one ----
    dominant, warm_times = None, {}
    if not args.no_kernel_timing:
        for e in engs:
            e.timing("*")
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    if not args.no_kernel_timing:
        warm_times = merge_times([e.timing_read() for e in engs])
        dominant = max(warm_times, key=lambda k: warm_times[k][0])
        for e in engs:
            e.timing(dominant if args.time_every <= 1 else "%s/%d" % (dominant, args.time_every))

    # ---- timed region ----
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)

    ktimes = merge_times([e.timing_read() for e in engs]) if dominant else {}
    step_times = {}

    # ---- the same steps with the lanes running free (no ordering between their passes): reported beside the timed region ----
    free = None
    if NS > 1 and args.stagger == 2 and args.free_running:
        for e in engs:
            capi._check(capi.lib.lfa_chain_passes(e._h, None))
        for _ in range(max(1, args.warmup)):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_free = max_over_ranks(time.perf_counter() - t1)
        kfree = merge_times([e.timing_read() for e in engs]) if dominant else {}
        free = {"ms_per_step": round(dt_free / args.steps * 1e3, 3), "steps": args.steps}
        if dominant and dominant in kfree and kfree[dominant][1] > 0:
            free["dominant_avg_launch_ms"] = round(kfree[dominant][0] / kfree[dominant][1], 4)
        for i in range(NS):
            capi._check(capi.lib.lfa_chain_passes(engs[i]._h, engs[i - 1]._h))
    if dominant:                                   # one more, untimed, step with every kernel bracketed: the per-kernel table
        for e in engs:
            e.timing("*")
        step(1)                                    # on ONE stream: kernel times without the other lane's kernels beside them
        torch.cuda.synchronize()
        step_times = merge_times([e.timing_read() for e in engs])
    for e in engs:
        e.timing(None)

    pred_mode_seen = bool(engs[0].debug_pred_path(0))      # the last call's first image took its mask from the predicted-threshold bit plane
    # ---- results of one step; the final gather of the per-strip records (the path's only collective) ----
    inf = capi.from_device(info, capi.INFO_DT, B)
    assert (inf["status"] == 0).all(), "strip capacity overflow"
    amb = int(inf["ambiguous"].sum())
    all_counts, all_rec, gather_s = gather_records(strips, counts, peaks, max_strips, B)
    total_strips = int(all_counts.sum())
    assert total_strips == sum_over_ranks(int(counts.sum().item())) and len(all_rec) == total_strips
    assert len(all_counts) == world * B and (np.diff(all_rec["image"]) >= 0).all()
    n_peaks = int(all_rec["peaks"]["n_peaks"].sum())
    for e in engs:
        e.close()
    del batch, profiles
    torch.cuda.empty_cache()

    # ---- N > 1: configs[3] (the configuration BASELINE.json names for the scaling curve) on every rank, with its own gather
    if world > 1 and not args.no_configs:
        bc = importlib.import_module("benchmarks_configs")
        r = bc.bench_big(B=args.config_batch, CH=min(args.config_chunk, args.config_batch), pool=min(4, args.config_batch),
                         steps=args.config_steps, seed0=77 + 4 * rank, keep_pipeline=True, barrier=barrier)
        pl = r.pop("_pipeline")
        dt3 = max_over_ranks(r["ms_per_step"] * 1e-3)
        s3 = sum_over_ranks(r["strips_per_step"])
        c3, rec3, g3 = gather_records(pl.strips, pl.counts, pl.peaks, pl.ms, pl.B)
        assert int(c3.sum()) == s3 == len(rec3)
        pl.close()
        del pl
        r.update({"n_gpus": world, "scaling": "weak", "ms_per_step": round(dt3 * 1e3, 3),
                  "images_per_s": round(world * r["images"] / dt3, 1), "strips_per_s": round(s3 / dt3, 1), "strips_per_step": s3,
                  "e2e_frac_hbm": round(r["images"] / dt3 * 3072 * 4096 * 3 / (HBM_PEAK_GBS * 1e9), 4),
                  "gather": {"records": len(rec3), "record_bytes": rec3.dtype.itemsize, "seconds": round(g3, 5),
                             "backend": "gloo (rehearsal)" if rehearsal else "rccl"}})
        configs = {"configs[3]": r}

    if rank == 0:
        value = total_strips * args.steps / dt
        images_s = world * B * args.steps / dt
        roof = None
        if dominant and dominant in ktimes and ktimes[dominant][1] > 0:
            ms, launches = ktimes[dominant]
            avg_ms = ms / launches
            # algorithmic bytes of one launch = what the stage must read + what it must write, once each:
            # the three image passes of the colour front end; every other kernel is priced at the image bytes (a lower bound
            # on nothing: they are latency-bound and never dominant)
            # SURVEY.md 8(d) prices S1+S2 (gray / gamma / histogram) at N*C*s bytes READ: that is `frac`.  What the kernel
            # additionally writes by this implementation's own choice (k_hist_rgb: the 4-bit class plane, N/2 bytes) is
            # reported beside it as `frac_with_plane_store`, never inside `frac`.  (Settled handles write ONE bit per pixel -- the
            # predicted-threshold plane, N/8 bytes -- and `plane_store_bytes_per_launch` says which plane this run wrote.)
            npx = H * W
            per_image = {"k_minmax_rgb": npx * 3,                 # image in
                         "k_hist_rgb": npx * 3,                    # image in
                         "k_mask": npx // 2 + npx // 8}.get(dominant, img_bytes)   # class plane in, bit mask out
            # plane out: one bit per pixel in the predicted-threshold mode (what the settled handles of this run write), else the 4-bit class plane
            own_store = {"k_hist_rgb": npx // 8 if pred_mode_seen else npx // 2}.get(dominant, 0)
            alg_bytes = CH * per_image
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")       # HBM bytes per launch from rocprofv3 --pmc passes
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                key = f"{dominant}:{H}x{W}x{C}:chunk{CH}"
                traffic = tj.get(key)
                if traffic is not None:
                    traffic_source = tj.get("_source", "profiles/traffic.json")
            roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                    "launches_in_timed_region": args.steps * ((B + CH - 1) // CH), "timed_every": max(1, args.time_every),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "frac_with_plane_store": round((alg_bytes + CH * own_store) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "plane_store_bytes_per_launch": CH * own_store,
                    "note": (f"timed region: {NS} lanes, the image passes of consecutive chunks chained (lfa_chain_passes), so a launch of "
                             "this kernel has the chip to itself but for the tail kernels of the chunks before; "
                             if NS > 1 and args.stagger == 2 else
                             f"timed region: {NS} lanes running free: a launch shares the CUs with the other lanes' passes, so `frac` says "
                             "how much else ran beside the kernel, not how well it uses the memory system; ") +
                            "every `timed_every`-th launch of it on each lane is bracketed by HIP events (`launches` of "
                            "`launches_in_timed_region`; --time-every 1 brackets all of them, at about 2 % of the step for the "
                            "records between the kernels); "
                            "`isolated` is the same kernel in one extra untimed step on a single stream; `traffic` is "
                            "not measured in this run: it is the PMC figure (2*FETCH_SIZE + WRITE_SIZE, KiB units) of the "
                            "rocprofv3 passes named in `traffic_source`"}
            if dominant in step_times and step_times[dominant][1] > 0:
                iso_ms = step_times[dominant][0] / step_times[dominant][1]
                roof["isolated"] = {"avg_launch_ms": round(iso_ms, 4), "achieved": round(alg_bytes / (iso_ms * 1e-3) / 1e9, 1),
                                    "frac": round(alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # the whole step in HBM bytes actually moved: per-launch PMC traffic of every kernel of a chunk (profiles/traffic.json,
        # measured at this shape and chunk size) x chunks per step / step time
        sustained = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            suffix = f":{H}x{W}x{C}:chunk{CH}"
            per_chunk = sum(v for k, v in tj.items() if k.endswith(suffix))
            if per_chunk:
                per_step = per_chunk * ((B + CH - 1) // CH)
                gbs = per_step / (dt / args.steps) / 1e9
                sustained = {"hbm_bytes_per_step": per_step, "GBps": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                             "algorithmic_bytes_per_step": B * img_bytes, "traffic_over_algorithmic": round(per_step / (B * img_bytes), 3),
                             "source": tj.get("_source", "profiles/traffic.json")}
        out = {
            "metric": "LFA strips analysed/sec (detect+profile+peaks)", "value": round(value, 1), "unit": "strips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: {B} synthetic {W}x{H} RGB uint8 images per GPU, 8 strips each, "
                                   f"end-to-end detect+profile+peaks", "images_per_gpu": B, "chunk": CH, "streams": NS,
                       "schedule": {0: "lanes run free", 1: "a chunk starts after the previous chunk's histogram pass",
                                    2: "image passes of consecutive chunks chained, tails beside them"}[args.stagger if NS > 1 else 0],
                       "pool": args.pool,
                       "min_maxr": mm_r, "min_maxc": mm_c, "parallelism": f"images sharded over {world} GPU(s), no collective on the data path; "
                                                                           "one gather of the per-strip records after the timed region"},
            "images_per_s": round(images_s, 1), "strips_per_step": total_strips,
            "e2e_fraction_of_hbm_roofline": round(images_s / world * img_bytes / (HBM_PEAK_GBS * 1e9), 4),
            "ambiguous_pixels": amb, "peaks_in_gathered_records": n_peaks,
            "gather": {"records": len(all_rec), "record_bytes": all_rec.dtype.itemsize, "seconds": round(gather_s, 5),
                       "backend": "gloo (rehearsal)" if rehearsal else ("rccl" if world > 1 else "none (one rank)")},
            "roofline": roof, "sustained_hbm": sustained, "cpu_baseline": cpu,
            "free_running": (dict(free, value=round(total_strips * args.steps / (free["ms_per_step"] * 1e-3 * args.steps), 1),
                                  note="the same steps with no ordering between the lanes' passes: passes of different chunks "
                                       "share the chip (their launches last 2-3x as long), the step is ~3 % shorter") if free else None),
            "kernel_ms_per_step": {k: round(v[0], 3) for k, v in sorted(step_times.items(), key=lambda kv: -kv[1][0])},
            "configs": configs,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
