#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hipcanny hot path (see BASELINE.json / SURVEY.md §8d).

A "step" is one pass of the full Canny pipeline (Mode R, thresholds 10/40, final stage HYSTER)
over one batch of device-resident synthetic 1080p grayscale frames, through the C ABI
(hc_run_device).  `value` = frames/s of the whole job (all ranks), inputs already in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Frames shard over ranks with no data-path collective (independent frames): weak
scaling, one context + one stream per device; torch.distributed only carries the barrier and the
MAX-reduce of the timing.  torch is plumbing here: device memory, stream, rendezvous.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cudacam_amd import api, shard, synth  # noqa: E402

W, H = 1920, 1080          # configs[1]; --width/--height select another BASELINE config (e.g. 3840x2160)
LOW, HIGH = 10, 40
HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBPS = 6290.0


# hc_last_run_info's front form -> (config name, kernel name)
CLOCK_WARMUP_S = 0.6   # untimed steps until the GPU has been busy this long (clock ramp-up of a process that starts on an idle GPU)
FORM_NAME = {5: ("front-mx", "k_front_mx"), 2: ("front8", "k_front8"), 4: ("front8-half", "k_front8 (half-strip form)"), 1: ("split", "k_blur+k_nms"), 0: ("fused4", "k_front"), 3: ("k_front8o", "k_front8o"), -1: ("k_front_o", "k_front_o")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (0.3 s of GPU time at the default batch; the pipelined mode exposes one hysteresis tail per run of steps)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU (2 GiB in + 2 GiB out at 1080p; hysteresis is latency-bound, larger batches amortise it)")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic frames (tiled to the batch; about 0.7 s of host time each at 1080p)")
    ap.add_argument("--kind", default="natural", choices=["natural", "noise"], help="content of the single batch of --rotate 1")
    ap.add_argument("--rotate", type=int, default=4, help="content rotation: the steps cycle through N DIFFERENT batches -- natural (seed set A), natural (seed set B), iid noise, "
                    "three natural frames blended as BGR -> grey would -- so that the library's adaptive choices (hysteresis launches queued, tile height, worklists, grids) "
                    "meet content they did not just see; 1 = the same batch every step (rounds 1-2)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--hyst-launches", type=int, default=0, help="hysteresis launches queued per run (0 = auto)")
    ap.add_argument("--no-pipeline", action="store_true", help="disable HC_OPT_PIPELINE (default on: run i+1's VALU-bound front kernel overlaps run i's latency-bound hysteresis on a second stream)")
    ap.add_argument("--out-buffers", type=int, default=0, help="pipelined mode: output buffers used in turn (a run into memory that an earlier, still unfinished run writes waits for that run); 0 = as many as the context keeps runs in flight: 2, or 4 for small batches")
    ap.add_argument("--front", default=None, choices=["front8", "split", "fused4"], help="front path (HC_OPT_FRONT_SPLIT): front8 = one kernel, 8 px per lane (default; Mode O: k_front8o); split = k_blur + k_nms; fused4 = the 4-px fused kernel (Mode O: both = k_front_o)")
    ap.add_argument("--mx", default="auto", choices=["auto", "never", "always"], help="k_front_mx (HC_OPT_FRONT_MX): the front path with blur and Sobel sums on the matrix pipe; auto = the library's default (never: opt-in)")
    ap.add_argument("--slots", default="auto", choices=["auto", "2", "3"], help="pipelined big batches: output sets in flight (HC_OPT_PIPELINE_SLOTS): auto = two, a third on trial while the hysteresis chain bounds the step")
    ap.add_argument("--wpb", default="auto", choices=["auto", "1", "4"], help="waves per workgroup of the front kernel (HC_OPT_FRONT_WPB): auto = the library's run-time rule")
    ap.add_argument("--dense", default="auto", choices=["auto", "never", "always"], help="k_front8's dense path (HC_OPT_FRONT_DENSE): wave-wide NMS for windows full of candidates")
    ap.add_argument("--mode", default="R", choices=["R", "O"], help="R: reference-exact pipeline (default, the headline); O: cv::Canny semantics")
    ap.add_argument("--channels", type=int, default=1, choices=[1, 3], help="3: interleaved BGR input (grey conversion fused into the load)")
    ap.add_argument("--per-channel", action="store_true", help="with --channels 3: one edge map per channel (BASELINE configs[4])")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the CPU baseline sample (0 = auto)")
    ap.add_argument("--sweep", action="store_true", help="one JSON line per batch size in {1, 8, 64, 256, 1024} (SURVEY 8d C2) instead of the single headline line")
    ap.add_argument("--sweep-batches", default="1,8,64,256,1024", help="the batch sizes of --sweep")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the bounded host-fed (PCIe-inclusive) end-to-end leg")
    return ap.parse_args()


def main():
    a = parse()
    global W, H, LOW, HIGH
    W, H = a.width, a.height
    if a.mode == "O":
        LOW, HIGH = 50, 150
    api.preload_hip_runtime()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "HC_BENCH_LOCAL_DEVICE" in os.environ:   # tests: several ranks on ONE device (the 1-GPU box exercising the N > 1 launch path)
        local = int(os.environ["HC_BENCH_LOCAL_DEVICE"])
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HC_BENCH_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" IS RCCL on ROCm
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        # the launcher's world must be the one asked for: one rank per GPU, frames sharded over exactly --gpus devices
        assert dist.get_world_size() == a.gpus == world, f"--gpus {a.gpus} but the process group has {dist.get_world_size()} ranks"
    elif a.gpus != 1:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus} (WORLD_SIZE is 1)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if a.sweep:
        if world != 1:
            raise SystemExit("--sweep is a single-GPU measurement")
        for b in [int(x) for x in a.sweep_batches.split(",")]:
            a.batch = b
            a.steps_eff = max(a.steps, min(2000, 20480 // b))   # small batches: enough steps for a stable mean
            run_config(a, torch, dist, world, rank, local, backend, brief=True)
        return
    run_config(a, torch, dist, world, rank, local, backend, brief=False)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_config(a, torch, dist, world, rank, local, backend, brief):
    steps = getattr(a, "steps_eff", a.steps)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B = a.batch
    f0, f1 = shard.frame_range(B * world, rank, world)   # this rank's block of every step's frame stream
    assert f1 - f0 == B
    C = a.channels
    if C == 3 and W % 4:
        raise SystemExit("--channels 3 needs a width that is a multiple of 4 (whole 12-byte pixel groups in a tight row)")
    if a.per_channel and a.mode != "R":
        raise SystemExit("--per-channel is a mode R option")
    # ---- content: `rot` different batches, the steps cycle through them (every step of rounds 1-2 processed the
    # same batch, so "what the last run needed" always predicted the next run perfectly) ------------------------------------
    rot = max(1, a.rotate)
    kinds = [a.kind] if rot == 1 else [("natural", "natural-B", "noise", "blend")[k % 4] for k in range(rot)]
    nat_per = max(2, min(a.unique, B) // (3 if rot > 1 else 1))   # distinct natural frames per natural batch (0.7 s of host time each at 1080p)
    seed_r = synth.SEED0 + 1000 * rank
    _nat_cache = {}

    def nat(seed_off, n):   # (n, H, W) natural frames, cached by seed offset
        key = (seed_off, n)
        if key not in _nat_cache:
            _nat_cache[key] = synth.frames("natural", W, H, n, seed=seed_r + seed_off)
        return _nat_cache[key]

    def planes(kind, chan, bi):   # the distinct frames of one batch, one channel
        off = 77 * chan + 100003 * (bi // 4)
        if kind == "natural":
            return nat(off, nat_per)
        if kind == "natural-B":
            return nat(off + 500, nat_per)
        if kind == "noise":
            return synth.frames("noise", W, H, 8, seed=seed_r + 9000 + off)
        # "blend": what BGR -> grey makes of three unrelated natural planes (rgb2mono's weights, cannyEdgeD.cu:14-19) -- edges
        # of three images overlaid, chains that wind through the tiles: the stream that needed 25 hysteresis launches in round 2
        pa, pb = nat(off, nat_per).astype(np.uint32), nat(off + 500, nat_per).astype(np.uint32)
        pc = np.roll(pa, nat_per // 2, axis=0)[:, ::-1, :]
        return ((7 * pa + 38 * pb + 19 * pc) >> 6).astype(np.uint8)

    d_ins = []
    for bi, kind in enumerate(kinds):
        if C == 1:
            uniq = planes(kind, 0, bi)
        else:   # three differently seeded planes interleaved as B, G, R
            uniq = np.stack([planes(kind, c, bi) for c in range(3)], axis=-1)
        d_u = torch.from_numpy(np.ascontiguousarray(uniq)).to(dev)
        reps = (B + d_u.shape[0] - 1) // d_u.shape[0]
        d_ins.append(d_u.repeat(*([reps] + [1] * (d_u.dim() - 1)))[:B].contiguous())   # (B, H, W[, 3]) u8, tight pitch
        del d_u
    _nat_cache.clear()
    d_in = d_ins[0]
    n_out = 3 * B if a.per_channel else B
    # output batches used in turn, as a pipelined consumer would (run i's maps are read while run i+1 computes): as many
    # as the context keeps runs in flight (hc_pipeline_depth: 3 for big batches -- two slots, a third while the hysteresis chain bounds the step -- 4 for small ones); with a single one the library
    # falls back to the non-provisional expand to keep run i+1's map intact
    # (--front split / fused4 in mode R: the round-1 kernels live in the test library, cudacam_amd/libhipcanny_legacy.so)
    ctx = api.Context(W, H, C, B, api.MODE_R if a.mode == "R" else api.MODE_O, device=local,
                      front_split=None if a.front is None else {"front8": 2, "split": 1, "fused4": 0}[a.front])
    ctx.set_option(api.OPT_PIPELINE, 0 if a.no_pipeline else 1)
    if a.per_channel:
        ctx.set_option(api.OPT_PER_CHANNEL, 1)
    d_outs = [torch.empty((n_out, H, W), dtype=torch.uint8, device=dev) for _ in range(1 if a.no_pipeline else (a.out_buffers or ctx.pipeline_depth(B)))]
    d_out = d_outs[0]

    ctx.set_thresholds(LOW, HIGH)
    ctx.set_tuning(a.chunk, a.hyst_launches)
    if a.mx != "auto":
        ctx.set_option(api.OPT_FRONT_MX, {"never": 0, "always": 1}[a.mx])
    if a.slots != "auto":
        ctx.set_option(api.OPT_PIPELINE_SLOTS, int(a.slots))
    if a.wpb != "auto":
        ctx.set_option(api.OPT_FRONT_WPB, int(a.wpb))
    if a.dense != "auto":
        ctx.set_option(api.OPT_FRONT_DENSE, {"never": 0, "always": 1}[a.dense])
    # the context keeps its own (non-blocking) stream: the inputs were produced before the synchronize below, and the
    # timed region is bracketed by hc_sync + torch.cuda.synchronize, so no ordering with torch's stream is needed

    nstep = [0]

    def step():
        o = d_outs[nstep[0] % len(d_outs)]
        src = d_ins[nstep[0] % len(d_ins)]
        nstep[0] += 1
        ctx.run_device(src.data_ptr(), W * C, W * C * H, o.data_ptr(), W, W * H, B, api.CannyStage.HYSTER)

    # Untimed, before the W warm-up steps: steps until the GPU has been busy for CLOCK_WARMUP_S -- a process that starts on
    # an idle MI355X runs its first ~100 ms of kernels at a fraction of the clock (a 28-step run from idle measured 5.4 ms
    # per step against 2.5 ms once the clocks are up, tools/experiments/clock_probe.sh) -- and, with a content rotation, as
    # many more as it takes for the timed region to begin with the rotation's first batch, after at least one whole rotation
    # (so that `value` does not depend on where the warm-up ended; with --steps a multiple of the rotation every content
    # is timed equally often).
    clock_steps = 0
    tw = time.perf_counter()
    while time.perf_counter() - tw < CLOCK_WARMUP_S:
        for _ in range(4):
            step()
        ctx.sync()
        clock_steps += 4
    for _ in range(a.warmup):
        step()
    align_steps = (-nstep[0]) % rot if rot > 1 else 0
    if rot > 1 and nstep[0] + align_steps < 2 * rot:
        align_steps += rot
    for _ in range(align_steps):
        step()
    ctx.sync()
    torch.cuda.synchronize(dev)
    ctx.enable_profiling(True)     # hipEvent pairs around the fused kernel, on the launch stream
    ctx.profile_get(reset=True)
    ctx.hysteresis_totals(reset=True)
    first_timed = nstep[0]

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()                      # queues what the pipelined mode still holds back (the last step's hysteresis), waits
                                    # for all of it and verifies the convergence of every step
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    ksums, kruns = ctx.profile_get_front()
    intervals = ctx.profile_intervals(steps + 8)
    front_each = ctx.profile_front_each(steps + 8)
    h_runs, h_continued, h_work, h_queued = ctx.hysteresis_totals()
    sums, nruns = ctx.profile_get(reset=True)
    work_launches, continued = ctx.hysteresis_info()
    in_staged, out_staged, front_form = ctx.last_run_info()
    front_waves = ctx.front_waves_per_workgroup() if a.mode == "R" else None   # k_front8: 4, or 1 while the hysteresis stream has the slack (HC_OPT_FRONT_WPB)
    slots_used = ctx.pipeline_slots_in_use()   # 2, or 3 once the context saw the hysteresis chain outlast the next front kernel

    # Untimed extra leg (rank 0, N = 1, rotation on): rounds 1-2 measured ONE natural batch processed every step; the same
    # here over a few steps, so that the line carries the figure that compares with theirs.  Never part of `value`.
    same_batch = None
    if rot > 1 and world == 1 and not brief:
        nsame = max(8, min(24, steps))
        for k in range(3):   # warm: the adaptive estimates settle on this content
            ctx.run_device(d_ins[0].data_ptr(), W * C, W * C * H, d_outs[k % len(d_outs)].data_ptr(), W, W * H, B, api.CannyStage.HYSTER)
        ctx.sync()
        ctx.profile_get(reset=True)
        ts0 = time.perf_counter()
        for k in range(nsame):
            ctx.run_device(d_ins[0].data_ptr(), W * C, W * C * H, d_outs[k % len(d_outs)].data_ptr(), W, W * H, B, api.CannyStage.HYSTER)
        ctx.sync()
        ts1 = time.perf_counter()
        ssums, sruns = ctx.profile_get(reset=True)
        sfront = ssums[1] / max(sruns, 1)
        same_batch = {"content": kinds[0], "steps": nsame, "value": round(B * nsame / (ts1 - ts0), 1), "unit": "frames/s", "ms_per_step": round((ts1 - ts0) / nsame * 1e3, 4),
                      "kernel_ms": round(sfront, 4),
                      "roofline_frac": round(float(W * H * C + W * H * (3 if a.per_channel else 1)) * B / (sfront * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if sfront > 0 else None,
                      "note": "the same natural batch every step (the workload of rounds 1-2; bench.py --rotate 1 times it as the headline)"}

    elapsed = shard.reduce_max_seconds(t1 - t0, dist if world > 1 else None, dev if backend == "nccl" else None)   # (gloo reduces host tensors)

    if rank == 0:
        frames_total = B * steps * world
        fps = frames_total / elapsed
        alg_bytes_per_frame = float(W * H * C + W * H * (3 if a.per_channel else 1))   # 2*W*H per mono frame
        alg_bytes_per_launch = alg_bytes_per_frame * B
        front_ms = sums[1] / max(nruns, 1)
        hyst_ms = sums[2] / max(nruns, 1)
        achieved = alg_bytes_per_launch / (front_ms * 1e-3) / 1e9 if front_ms > 0 else 0.0
        out = {
            "metric": f"frames/sec, {W}x{H} " + ("grayscale" if C == 1 else "3-channel per-channel" if a.per_channel else "BGR") + " Canny (" + ("5-stage, Mode R" if a.mode == "R" else "cv::Canny semantics, Mode O") + ", device-resident)",
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": a.warmup,
            "warmup_untimed_extra": {"clock_steps": clock_steps, "clock_warmup_s": CLOCK_WARMUP_S, "rotation_alignment_steps": align_steps,
                                     "note": "untimed steps before / after the W warm-up steps: until the clocks are up, and so that the timed region starts with the rotation's first batch"},
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "ms_per_frame": round(elapsed / (steps * B) * 1e3, 6),
            # per-step times on the device clock: end-of-step to end-of-step hipEvent intervals (rank 0), SURVEY 8d "median and p10/p90"
            "step_ms": _percentiles(intervals),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": (f"synthetic ({a.kind}, {d_ins and min(a.unique, B)} distinct frames tiled to the batch, the same batch every step)" if rot == 1 else
                     f"synthetic, {rot} different batches in rotation ({', '.join(kinds)}); {nat_per} distinct frames per natural / blend batch, 8 per noise batch, tiled to the batch"),
            "content_rotation": rot,
            "output_buffers": len(d_outs),
            "pipeline_slots": slots_used,
            "config": {"workload": (f"configs[1]: 1920x1080 grayscale, full 5-stage HIP pipeline, batch {B} frames/step/GPU" if (W, H, a.mode) == (1920, 1080, "R")
                                    else f"{W}x{H} " + ("grayscale" if C == 1 else "BGR, per-channel Canny" if a.per_channel else "BGR -> grey") + f", mode {a.mode}, batch {B} frames/step/GPU"),
                       "width": W, "height": H, "batch": B, "low": LOW, "high": HIGH, "sharding": f"frames x{world}",
                       "pipeline": not a.no_pipeline, "front": FORM_NAME.get(front_form, ("?", "?"))[0], "content_rotation": rot,
                       "world_size": world, "backend": backend},
            "e2e_alg_GBps": round(alg_bytes_per_frame * frames_total / elapsed / 1e9, 1),
            "roofline": {
                "bound": "hbm", "kernel": FORM_NAME.get(front_form, ("?", "?"))[1], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "frac_of_measured_copy": round(achieved / HBM_MEASURED_COPY_GBPS, 4),
                "traffic": None, "kernel_ms": round(front_ms, 4),
                "kernel_ms_each": ({"k_blur": round(ksums[0] / kruns, 4), "k_nms": round(ksums[1] / kruns, 4)} if kruns else None),
                "hyst_expand_ms": round(hyst_ms, 4), "launches_timed": nruns,
            },
            # `continued` = runs of the whole timed region that needed the host-side continuation (each one stalls the stream)
            "hysteresis": {"launches_with_work": work_launches, "continued": h_continued, "runs": h_runs,
                           "launches_with_work_mean": round(h_work / max(h_runs, 1), 2), "launches_queued_mean": round(h_queued / max(h_runs, 1), 2)},
            "by_content": _by_content(kinds, first_timed, intervals, front_each, B, alg_bytes_per_launch, steps),
            "by_content_note": ("step_ms = device-side interval between the completion of the previous run and of this one (front kernel + this "
                                "run's hysteresis tail - the previous run's); kernel_ms = this content's front kernel, which runs BESIDE THE "
                                "PREVIOUS content's hysteresis (noise: 3.9 ms after a natural batch, 5.0 ms after another noise batch)") if rot > 1 else None,
            "same_batch_every_step": same_batch,
            # the caller's buffers were used in place (no hidden staging copies) and the front path that actually ran
            "buffers": {"input_staged": in_staged, "output_staged": out_staged, "front_form": {5: "k_front_mx", 4: "k_front8 (half-strip form)", 3: "k_front8o", 2: "k_front8", 1: "k_blur+k_nms", 0: "k_front", -1: "k_front_o"}.get(front_form),
                        "front_waves_per_workgroup": front_waves},
        }
        # HBM bytes per launch of the front kernels from the committed PMC passes (separate rocprofv3 runs of this
        # command, tools/collect_profiles.sh); only quoted when that profile was taken on this very configuration
        import glob
        for pf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json"))):
            try:
                prof = json.load(open(pf))
                if prof["config"] == out["config"] and prof["metric"] == out["metric"]:
                    out["roofline"]["traffic"] = round(prof["front_kernels_hbm_bytes_per_launch"] / 1e9, 3)
                    out["roofline"]["traffic_unit"] = "GB per launch (2*FETCH_SIZE + WRITE_SIZE, " + os.path.relpath(pf, ROOT) + ")"
                    out["roofline"]["algorithmic_GB_per_launch"] = round(alg_bytes_per_launch / 1e9, 3)
                    # the same kernels' real HBM traffic against the peak (how busy the memory system is, not a roofline claim)
                    out["roofline"]["traffic_GBps"] = round(out["roofline"]["traffic"] / (front_ms * 1e-3), 1) if front_ms > 0 else None
                    out["roofline"]["traffic_frac_of_peak"] = round(out["roofline"]["traffic"] / (front_ms * 1e-3) / HBM_PEAK_GBPS, 4) if front_ms > 0 else None
            except (OSError, KeyError, ValueError):
                pass
        if not brief and not a.no_cpu_baseline and C == 1 and world == 1:   # the CPU baseline is an N = 1 figure
            # (untimed) one more run per content into the first output buffer: the oracle checks a sample of EVERY batch of the rotation
            ncheck = min(B, 16)
            samples = []
            for kind, src in zip(kinds, d_ins):
                ctx.run_device(src.data_ptr(), W * C, W * C * H, d_out.data_ptr(), W, W * H, B, api.CannyStage.HYSTER)
                ctx.sync()
                samples.append((kind, src[:ncheck].cpu().numpy(), d_out[:ncheck].cpu().numpy()))
            out["cpu_baseline"] = cpu_baseline(a, d_ins, samples)
            if a.mode == "R":   # the closer stand-in for north_star's "host-core OpenCV cv::Canny baseline", at top level and labelled
                om = out["cpu_baseline"]["other_mode"]
                out["cpu_baseline_cv_canny_restatement"] = {"value": om["value"], "unit": om["unit"], "cores": om["cores"], "kind": "port",
                                                            "sample": out["cpu_baseline"]["sample"],
                                                            "note": "cv::Canny(img, 50, 150, 3, false) as restated in oracle/canny_oracle.c (Mode O): never compared with a real OpenCV (none on this host)"}
        if not brief and not a.no_host_fed and world == 1:
            out["host_fed"] = host_fed(a, d_in)
        print(json.dumps(out), flush=True)
    ctx.close()


def _by_content(kinds, first_timed, intervals, front_each, B, alg_bytes_per_launch, steps):
    """Per content kind: the steps that processed it (interval j ends with timed step j + 1), their median step time and
    the front kernel's mean time on that content."""
    if len(kinds) == 1:
        return None
    if len(front_each) != steps:
        return None   # the library keeps the timestamps of the last 255 runs between two syncs: with more steps than that the samples no longer line up with the steps
    out = {}
    for ki, kind in enumerate(kinds):
        st = [intervals[j] for j in range(len(intervals)) if (first_timed + j + 1) % len(kinds) == ki]
        fr = [front_each[j] for j in range(len(front_each)) if (first_timed + j) % len(kinds) == ki]
        med = float(np.median(st)) if st else None
        fm = float(np.mean(fr)) if fr else None
        out[kind] = {"steps": len(fr), "step_ms_median": round(med, 4) if med else None, "step_ms_max": round(max(st), 4) if st else None,
                     "frames_per_s": round(B / med * 1e3, 1) if med else None,
                     "kernel_ms": round(fm, 4) if fm else None,
                     "roofline_frac": round(alg_bytes_per_launch / (fm * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if fm else None}
    return out


def _percentiles(ms):
    if not ms:
        return None
    v = np.sort(np.asarray(ms, np.float64))
    q = lambda f: float(v[min(len(v) - 1, int(f * len(v)))])
    return {"n": len(v), "median": round(q(0.5), 4), "p10": round(q(0.1), 4), "p90": round(q(0.9), 4), "min": round(float(v[0]), 4), "max": round(float(v[-1]), 4)}


def host_fed(a, d_in):
    """End-to-end with the frames in HOST memory (never `value`): page-locked staging, a ring of three contexts so that
    batch i uploads while batch i-1 computes and batch i-2 downloads -- the Python twin of cvp::io::FrameStreamer
    (include/cvp/frameIO.hpp).  Bounded: about a second of work."""
    import ctypes as C
    lib = api.load_library()
    nb = min(32, d_in.shape[0])   # 64 MiB of 1080p frames per batch: below that the hand-over between copies and kernels shows (tools/experiments/pcie_raw2.hip)
    ch = a.channels
    frame_in, frame_out = W * H * ch, W * H * (3 if a.per_channel else 1)
    host_frames = d_in[:nb].cpu().numpy()
    ring = []
    for _ in range(3):
        ctx = api.Context(W, H, ch, nb, api.MODE_R if a.mode == "R" else api.MODE_O)
        ctx.set_option(api.OPT_COPY_STREAMS, 1)
        if a.per_channel:
            ctx.set_option(api.OPT_PER_CHANNEL, 1)
        ctx.set_thresholds(LOW, HIGH)
        hin, hout = lib.hc_host_alloc(frame_in * nb), lib.hc_host_alloc(frame_out * nb)
        C.memmove(hin, host_frames.ctypes.data, frame_in * nb)
        ring.append((ctx, hin, hout))
    nbatches = max(6, min(200, int(2.0e9 // (frame_in * nb))))   # ~2 GB through PCIe each way
    busy = [False] * 3

    def finish(k):
        api._ck(lib.hc_download_end(ring[k][0].handle))
        busy[k] = False

    def stream(count):
        for i in range(count):
            k = i % 3
            if busy[k]:
                finish(k)
            ctx, hin, hout = ring[k]
            api._ck(lib.hc_upload(ctx.handle, C.c_void_p(hin), W * ch, frame_in, nb))
            ctx.run(api.CannyStage.HYSTER, nb)
            # the download is queued behind the run at once: it crosses PCIe while the next batches upload
            api._ck(lib.hc_download_begin(ctx.handle, C.c_void_p(hout), W, W * H, nb * (3 if a.per_channel else 1)))
            busy[k] = True
        for k in range(3):
            if busy[(count + k) % 3]:
                finish((count + k) % 3)

    stream(6)   # warm: first-touch of the page-locked buffers, first launches of every context
    t0 = time.perf_counter()
    stream(nbatches)
    dt = time.perf_counter() - t0
    for ctx, hin, hout in ring:
        ctx.close()
        lib.hc_host_free(C.c_void_p(hin))
        lib.hc_host_free(C.c_void_p(hout))
    frames = nbatches * nb
    out = {"value": round(frames / dt, 1), "unit": "frames/s", "batch": nb, "batches": nbatches, "contexts": 3, "staging": "page-locked (hc_host_alloc)",
           "pcie_GBps_each_way": round(frames * frame_in / dt / 1e9, 2), "note": "frames start and end in host memory; one staging thread (Python twin of cvp::io::FrameStreamer)"}
    # the C++ host pipeline itself (tools/stream_bench.cpp over cvp::io::FrameStreamer), as a child process: with the staging
    # buffers already filled (a decoder that writes straight into stage()), and with a 4-thread producer copying every batch in
    exe = os.path.join(ROOT, "tools", "bin", "stream_bench")
    if os.path.exists(exe) and a.mode == "R" and not a.per_channel:
        import subprocess
        out["cxx_streamer"] = {}
        for name, prod in (("staging_prefilled", 0), ("producer_4_threads", 4)):
            try:
                r = subprocess.run([exe, "--width", str(W), "--height", str(H), "--channels", str(ch), "--batch", str(nb), "--batches", str(max(20, nbatches)), "--producer", str(prod)],
                                   capture_output=True, text=True, timeout=120)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                out["cxx_streamer"][name] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:   # a missing runtime library, a timeout: reported, never fatal to the bench line
                out["cxx_streamer"][name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def cpu_baseline(a, d_ins, checks):
    """The oracle (a CPU port of the reference pipeline, kind "port") timed on the host cores on a bounded sample of the
    benchmark's frames -- at 1 thread and at all the threads this process may use; the GPU output of a sample of every
    batch of the rotation checked against it; the cv::Canny restatement (Mode O) beside it; and a real OpenCV if this host
    happens to have one."""
    from oracle import oracle as O   # test infrastructure: only this leg may touch it
    O.build()
    # the GPU box gives one GPU a 16-core CPU share: size the OpenMP pool to it
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    n = a.cpu_frames or max(cores * 4, 16)
    # the timed sample: frames of every batch of the rotation in equal parts (what a step processes on average)
    per = max(1, n // len(d_ins))
    sample = np.concatenate([d[:per].cpu().numpy() for d in d_ins])
    n = sample.shape[0]
    run = O.canny_r_batch if a.mode == "R" else O.canny_o_batch
    run(sample[:min(n, cores)], LOW, HIGH, threads=cores)   # warm the pages/threads

    def timed(fn, frames, threads):
        t0 = time.perf_counter()
        r = fn(frames, threads)
        return r, time.perf_counter() - t0

    ref, dt = timed(lambda f, t: run(f, LOW, HIGH, threads=t), sample, cores)
    n1 = max(2, min(n, 8))
    _, dt1 = timed(lambda f, t: run(f, LOW, HIGH, threads=t), sample[:n1], 1)
    same = {kind: bool(np.array_equal(run(src, LOW, HIGH, threads=cores), got)) for kind, src, got in checks}
    other = O.canny_o_batch if a.mode == "R" else O.canny_r_batch
    olo, ohi = (50, 150) if a.mode == "R" else (10, 40)
    _, dto = timed(lambda f, t: other(f, olo, ohi, threads=t), sample, cores)
    out = {"value": round(n / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{n} of the benchmark's {W}x{H} frames ({per} of each batch of the rotation), OpenMP over frames, oracle/canny_oracle.c ({'orc_canny_r' if a.mode == 'R' else 'orc_canny_o'})",
           "gpu_output_matches": all(same.values()), "gpu_output_matches_by_content": same, "frames_checked_per_content": int(checks[0][1].shape[0]),
           "one_thread": {"value": round(n1 / dt1, 2), "unit": "frames/s", "cores": 1, "sample": f"{n1} frames"},
           "other_mode": {"mode": "O (cv::Canny restatement, 50/150)" if a.mode == "R" else "R (reference pipeline, 10/40)", "value": round(n / dto, 2), "unit": "frames/s", "cores": cores},
           "opencv": opencv_probe(sample, cores)}
    return out


def opencv_probe(sample, cores):
    """Real cv::Canny on the host cores, if OpenCV is discoverable here (the reference's Conan dependency `opencv`,
    conanfile.py:22, is not part of this image): timed, and compared with the Mode O restatement bit for bit."""
    try:
        import cv2  # noqa: F401
    except Exception as e:   # ImportError, or a broken binary wheel
        return {"available": False, "note": f"OpenCV not available on this host ({type(e).__name__}); the Mode O figures are the CPU restatement, not cv::Canny"}
    from oracle import oracle as O
    cv2.setNumThreads(cores)
    t0 = time.perf_counter()
    maps = [cv2.Canny(f, 50, 150, apertureSize=3, L2gradient=False) for f in sample]
    dt = time.perf_counter() - t0
    want = O.canny_o_batch(sample, 50, 150, threads=cores)
    return {"available": True, "version": cv2.__version__, "value": round(len(sample) / dt, 2), "unit": "frames/s", "threads": cores,
            "restatement_matches_cv2": bool(all(np.array_equal(m, w) for m, w in zip(maps, want))),
            "note": "IPP / OpenCL builds of OpenCV may take a different code path than modules/imgproc/src/canny.cpp"}


if __name__ == "__main__":
    main()
