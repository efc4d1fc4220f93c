#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native MRI ray-marcher.

Metric (BASELINE.json): Msamples/s and achieved HBM GB/s, 512^3 fp32 volume @ 1024^2 x 512 steps.

N = 1: BASELINE config 3 ("C3") — one fp32 channel, perspective rays, central-difference gradient +
Blinn-Phong shading, early-ray termination (T <= 0.01), 1024 x 1024 px, on the synthetic scene of SURVEY.md
section 8(d) (``mrirt.synth``; intensityAlpha 16 so that termination fires).
N > 1, one rank per GPU: BASELINE config 4 ("C4") — the same kernel and
volume at a FIXED 2048 x 2048 image (strong scaling): 64 x 64 tiles dealt round-robin to the ranks, volume
replicated, one RCCL gather of the compact tile buffers to rank 0 per frame (each peer's 1/N of the frame crosses
its own xGMI link to the root once), de-tiling kernel on rank 0.

A *step* is one rendered frame: ray generation, march, composite and the framebuffer store — and, for N > 1,
the gather plus de-tiling (issued asynchronously: frame k's exchange overlaps frame k+1's march; all K frames
are fully exchanged and de-tiled before the clock stops).

``value`` = live samples of all ranks' frames / wall time.  Live samples (march-loop iterations that fetch the
volume) are counted by the kernel itself in an untimed pass, so skipping work cannot inflate the rate.  Inputs
are resident in HBM before the timed region.

``roofline`` (dominant kernel, this rank's launch): ``frac`` = HBM bytes per launch / kernel time / 8 TB/s, where
the bytes are the rocprofv3 PMC measurement committed in profiles/traffic.json — used only while the kernel
sources still have the digest they were measured on (tools/srchash.py); otherwise ``traffic`` is null and the
fraction falls back to the COMPULSORY bytes (volume once + framebuffer), a floor of the real traffic, and says
so in ``basis``.  SURVEY 8(d)'s algorithmic rate (cache-served taps, can exceed the HBM peak) is kept as
``algorithmic_GBs``; ``on_chip`` carries the two on-chip roofs that actually bind this kernel.

Launching.  ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself: the parent only counts devices, runs ``python -m torch.distributed.run
--nproc-per-node N`` on this file as a CHILD process (never an exec), relays rank 0's JSON line and exits with
the children's status; with fewer than N devices visible it exits non-zero instead of printing an N=1 line.  Under
torch.distributed.run (WORLD_SIZE set, as the driver launches it) each process is one rank.
``--backend gloo --dry-run`` walks the same launcher and the same tile sharding + exchange on host tensors without
a GPU (tests/test_bench_launcher.py).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # same guide: dense bf16 (the 5 PF headline includes 2:1 sparsity)
BYTES_PER_SAMPLE = 32         # 8 taps x 4 B, one fp32 channel          
BYTES_PER_SHADED = 192        # + 6 x 8 taps x 4 B central differences  
BYTES_PER_PIXEL = 16          # fp32 RGBA framebuffer store
DEPTH = 3                     # frames in flight in the N>1 loop (exchange slots; on the GPU each slot has its own HIP stream)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--volume", type=int, default=512, help="volume side N (N^3 fp32)")
    ap.add_argument("--image", type=int, default=0, help="image side; 0 = 1024 at one GPU (C3), 2048 at N > 1 (C4)")
    ap.add_argument("--march-steps", type=int, default=512)
    ap.add_argument("--math", default="strict", choices=["strict", "fast"])
    ap.add_argument("--layout", default="auto", choices=["auto", "vg", "vga", "quad", "brick", "linear"],
                    help="HBM layout of the volume; auto = vga (value+gradient float4 voxels in three axis-flat brick copies) when shading, quad otherwise")
    ap.add_argument("--no-shade", action="store_true", help="reference-only K1 (no gradient shading)")
    ap.add_argument("--alpha", type=float, default=16.0, help="intensityAlpha (16 = dense preset: ERT fires)")
    ap.add_argument("--variant", type=int, default=0, help="kernelVariant (experiments)")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--tile-skew", type=int, default=-1, help="tileSkew of the N>1 deal; -1 = tiles.balanced_skew (diagonals), 0 = plain row-major deal")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="rows of the frame the CPU baseline renders; 0 = skip, -1 = auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inr", action="store_true", help="skip the INR (MFMA) side measurement")
    ap.add_argument("--no-k1", action="store_true", help="skip the config-2 (reference K1 semantics) side measurement")
    ap.add_argument("--no-scaling-model", action="store_true", help="N=1: skip the one-GPU emulation of the 1/2/4/8-rank tile shares")
    ap.add_argument("--no-pipelined", action="store_true", help="N=1: skip the frames-in-flight side measurement (profiling runs: overlapped "
                                                                 "launches of the same kernel would blur its per-launch statistics)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N=1 only: still create the RCCL group (world size 1), render compact tiles and run the "
                         "asynchronous gather + de-tiling path — a single-GPU rehearsal of the N>1 code")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend: nccl (= RCCL, the product) or gloo (host tensors; only with --dry-run)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: every rank fills its round-robin tiles with a known pattern and runs the exchange step "
                         "(gather to rank 0 + de-tiling) on host tensors; checks the launcher and the N>1 plumbing")
    return ap.parse_args()


def launch_ranks(a):
    """``--gpus N`` (N > 1) outside torch.distributed.run: start the N ranks as children of this process.
    The parent only COUNTS devices (``torch.cuda.device_count()`` — on ROCm that may go through ``hipGetDeviceCount``, which does
    not create a context or touch a queue) and never renders; the ranks are a CHILD process (``subprocess.run``), not an exec of
    this one, so whatever the count initialised in the parent is irrelevant to them."""
    if not a.dry_run:
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) visible; refusing to report an N={have} run as N={a.gpus}",
                  file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def dry_run(a, world, rank):
    """The N>1 plumbing without a GPU: same tile ownership rule, same FrameExchange (pad to rank 0's tile count,
    gather to rank 0, de-tile), host tensors over gloo.  Tile t's pixels hold t + 1/16ths of their index, so a
    misplaced, missing or padded tile shows up in the assembled frame."""
    import torch
    import torch.distributed as dist
    from mrirt import tiles
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    image = a.image or 2048
    n_tiles = tiles.num_tiles(image, image, a.tile)

    def tile_pattern(t):
        px = torch.arange(a.tile * a.tile * 4, dtype=torch.float32).reshape(a.tile, a.tile, 4)
        return px / 16.0 + float(t)

    skew = tiles.balanced_skew(image, a.tile, world)
    ex = tiles.FrameExchange(image, image, a.tile, torch.float32, "cpu", depth=DEPTH, dst=0, skew=skew)
    owned = list(range(rank, n_tiles, world))
    assert len(owned) == ex.n_local
    t0 = time.perf_counter()
    frame = None
    for s in range(a.warmup + a.steps):
        slot = s % DEPTH
        if s >= DEPTH:
            frame = ex.finish(slot)
        buf = ex.local(slot)
        for lt, t in enumerate(owned):
            buf[lt] = tile_pattern(t)
        ex.submit(slot)
    for s in range(max(0, a.warmup + a.steps - DEPTH), a.warmup + a.steps):
        frame = ex.finish(s % DEPTH)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    counts = [tiles.local_tile_count(image, image, a.tile, r, world) for r in range(world)]
    if rank == 0:
        ok = frame is not None and tuple(frame.shape) == (image, image, 4)
        for t in range(n_tiles if ok else 0):
            x0, y0 = tiles.tile_origin(t, image, a.tile, skew)
            want = tile_pattern(t)[: image - y0, : image - x0]
            ok = ok and torch.equal(frame[y0:y0 + a.tile, x0:x0 + a.tile], want)
        print(json.dumps({"metric": "dry run of the N-rank launcher and exchange (no GPU, no rendering)", "value": 0.0,
                          "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(elapsed / max(1, a.warmup + a.steps) * 1e3, 4), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "backend": a.backend, "frame_ok": bool(ok), "tiles_per_rank": counts,
                          "config": {"workload": f"dry run: {image}x{image} px in {a.tile}x{a.tile} tiles over {world} rank(s)"}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def measured_traffic(key):
    """(HBM bytes per launch, on-chip utilisation figures, note) from the committed rocprofv3 PMC passes
    (profiles/traffic.json) — only when the entry was measured on the kernel sources this tree holds."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import srchash
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            entry = json.load(fh).get(key)
        if not entry:
            return None, None, "no PMC entry for this configuration"
        if entry.get("source_digest") != srchash.source_digest():
            return None, None, "PMC entry is stale: measured on other kernel sources (tools/srchash.py digest differs)"
        return int(entry["hbm_bytes_per_launch"]), entry.get("on_chip"), entry.get("source")
    except (OSError, ValueError, KeyError, ImportError) as e:
        return None, None, f"traffic.json unreadable: {type(e).__name__}"


def siren_c5(rng):
    """The 4 x 256 SIREN of BASELINE config 5 (7 -> 256 x 4 -> 4) with the notebook's initialisation
    (neumors_inr.ipynb:1150-1163: U(-r, r), r = sqrt(6 / fan_in), / w0 on the first layer)."""
    dims = [7, 256, 256, 256, 256, 4]
    params = []
    for i in range(len(dims) - 1):
        r = math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)
        params.append({"W": rng.uniform(-r, r, (dims[i], dims[i + 1])).astype(np.float32),
                       "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)})
    return dims, params


def inr_path(dev, frames=5):
    """The other half of the north star, measured in the same run: BASELINE config 5 AS A FRAME — 256^3 x 4
    modalities (BraTS-shaped synthetic scene), 512 x 512 px, 256 samples/ray, the prediction overlay's class of
    every LIVE sample queried from a 4 x 256 SIREN on the bf16 MFMA kernel (mrirt_render_brats_inr: ERT-aware
    passes — 96 steps on this scene, where no ray can terminate on intensity alone; 32 otherwise: plan + emit -> MLP -> composite).  HIP events around whole frames on the launch stream.
    ``roofline`` prices the frame: useful flops = live (composited) samples x flop/query over the frame time.
    ``mlp_kernel`` is the MLP kernel by itself on 67.1 M resident random queries (512^2 x 256 nominal)."""
    import ctypes as C
    import torch
    import mrirt
    from mrirt import _lib, inr, synth
    rng = np.random.default_rng(0)
    dims, params = siren_c5(rng)
    flop = 2 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4)

    # ---- the frame ---------------------------------------------------------------------------------------
    n, image, steps = 256, 512, 256
    vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
    lab = synth.synth_labels(n)
    zmu = [float(v[v != 0].mean()) for v in vols]
    zsg = [float(v[v != 0].std() + 1e-6) for v in vols]
    p5 = synth.brats_scene(n, image, steps, channels=4, show_seg=True, show_pred=True, intensity_alpha=0.4)
    gv = mrirt.upload_mod4(vols, (n, n, n))             # the four modalities as one float4 grid (MRIRT_LAYOUT_MOD4): 268 MB resident
    gl = mrirt.upload_grid(lab, (n, n, n), "brick")
    out = torch.empty((image, image, 4), dtype=torch.float32, device=dev)
    _, aux = inr.render_brats_inr(p5, gv, net, zmu, zsg, labels=gl, out=out, return_aux=True)     # untimed: accounting
    for _ in range(2):
        inr.render_brats_inr(p5, gv, net, zmu, zsg, labels=gl, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frames)]
    for e0, e1 in ev:
        e0.record(); inr.render_brats_inr(p5, gv, net, zmu, zsg, labels=gl, out=out); e1.record()
    torch.cuda.synchronize()
    frame_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    live, queries = aux["live_samples"], aux["queries"]
    tf_frame = flop * live / (frame_ms * 1e-3) / 1e12
    del gv, gl

    # ---- the MLP kernel alone ----------------------------------------------------------------------------
    nq = 512 * 512 * 256
    coords = torch.rand((nq, 3), device=dev) * 2 - 1
    feats = torch.randn((nq, 4), device=dev)
    cls = torch.empty(nq, dtype=torch.int16, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch(desc):
        _lib.check(_lib.lib().mrirt_inr_forward(C.byref(desc), C.c_void_p(coords.data_ptr()), C.c_void_p(feats.data_ptr()),
                                                nq, None, C.c_void_p(cls.data_ptr()), stream), "mrirt_inr_forward")
    def timed(desc):
        for _ in range(3):
            launch(desc)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for e0, e1 in ev:
            e0.record(); launch(desc); e1.record()
        torch.cuda.synchronize()
        return float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))

    # the bf16 MFMA pass with its near-tie MARKING but no second pass (MrirtInrDesc.flags = MRIRT_INR_MARK_ONLY), then the
    # class output as shipped: bf16 pass + split-bf16 re-evaluation of the marked points
    ms = timed(inr.with_flags(net, mark_only=True).desc)
    ms_refined = timed(net.desc)
    marked = float(inr.calibration(net)["rms_error"])
    tflops = flop * nq / (ms * 1e-3) / 1e12
    return {"workload": "C5 frame: 256^3 x 4 modalities (one float4 MOD4 grid) + seg, 512x512 px, 256 samples/ray, per-sample SIREN 7->4x256->4 "
                        f"(bf16 MFMA, fp32 accumulate), ERT-aware passes of {aux['chunk_steps']} steps",
            "value": round(live / (frame_ms * 1e-3) / 1e6, 1), "unit": "M live queries/s", "ms_per_frame": round(frame_ms, 3),
            "dtype": "bf16", "live_samples_per_frame": live, "mlp_queries_per_frame": queries,
            "nominal_samples_per_frame": image * image * steps,
            "roofline": {"bound": "mfma", "achieved": round(tf_frame, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf_frame / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                         "basis": "live samples x flop/query over the whole frame time (plan + emit + MLP + composite)",
                         "flop_per_query": flop},
            "mlp_kernel": {"workload": "the bf16 MLP kernel alone (inr_ws_kernel, weight-stationary; near-tie marking on, second pass off): "
                                       "67.1 M resident random queries, 5 launches back to back", "ms_per_launch": round(ms, 3),
                           "Mqueries_s": round(nq / (ms * 1e-3) / 1e6, 1),
                           "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                                        "basis": "2 x sum(in x out) flop per query over the launch time; the peak is the 2.4 GHz figure — under this kernel the "
                                                 "socket sits at its power limit (1.34-1.37 kW) and holds 1.73-2.15 GHz (in-kernel cycle counts / rocm-smi: "
                                                 "profiles/r04_power_clocks/); issued back to back on random operands this instruction mix sustains 0.59 of the peak "
                                                 "(tools/micro/mfma_power.hip), plain MFMA chains 0.685"},
                           "with_near_tie_refinement": {"ms_per_launch": round(ms_refined, 3), "Mqueries_s": round(nq / (ms_refined * 1e-3) / 1e6, 1),
                                                        "note": "classes as shipped: + inr_refine_kernel (split-bf16, three MFMAs per product) on the "
                                                                "points whose top-2 logit gap is below 3 sqrt(2) x the calibrated rms error "
                                                                f"({marked:.2e})"}}}


SIDE_TAG = 32768        # kernelVariant bit 15: the side measurements launch the benched kernel's tagged twin (same code, another symbol), so
                        # that a profiler's statistics of the default command keep the benched launches apart from them


def _tagged(ext):
    e = dict(ext)
    e["kernelVariant"] = int(e.get("kernelVariant", 0)) | SIDE_TAG
    return e


def pipelined_frames(dev, params, grid, ext, live, frames=40, streams=3):
    """A frame LOOP rather than a frame: the same config-3 frames dealt round-robin over `streams` HIP streams (each into its
    own buffer), so that a frame's fill and drain overlap its neighbours'.  Wall clock over `frames` frames.  Reported beside
    the headline, which stays one frame after the other on one stream (a single frame's latency is what it is)."""
    import torch
    import mrirt
    ext = _tagged(ext)
    ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    outs = [torch.empty((int(params["imageSize"][1]), int(params["imageSize"][0]), 4), dtype=torch.float32, device=dev) for _ in range(streams)]

    def run(k):
        for i in range(k):
            mrirt.render_brats(params, [grid], out=outs[i % streams], ext=ext, stream=ss[i % streams])
    torch.cuda.synchronize(); run(2 * streams); torch.cuda.synchronize()
    t = time.perf_counter(); run(frames); torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / frames * 1e3
    return {"what": f"{frames} frames of the same configuration dealt round-robin over {streams} HIP streams (frames in flight overlap; "
                    "same kernel, same bits per frame)", "ms_per_frame": round(ms, 4), "value": round(live / (ms * 1e-3) / 1e6, 1),
            "unit": "Msamples/s", "streams": streams,
            "note": "throughput of a frame loop, not a frame's latency: the headline `value` above is one frame after the other on one stream"}


def k1_reference_path(dev, frames=20):
    """BASELINE config 2 — the reference's own semantics, its primary parity target (inr/viewer/brats_rt.slang:117-165 as
    the viewer runs it): 256^3 x 4 modalities + seg overlay, 512 x 512 px, 256 steps/ray, perspective, no shading, STRICT
    math.  The four modalities are ONE float4 grid (MRIRT_LAYOUT_MOD4: what the shim binds for unshaded multi-modality frames)
    and the overlay a label-cell grid; the same frame from four QUAD grids is timed beside it.  HIP events around `frames`
    launches on the launch stream."""
    import torch
    import mrirt
    from mrirt import synth
    n, image, steps = 256, 512, 256
    vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
    gl = mrirt.upload_label_cells(synth.synth_labels(n), None, (n, n, n))      # the seg overlay as label cells: one 8-byte gather at the cell's own offset (what the shim binds)
    p = synth.brats_scene(n, image, steps, channels=4, show_seg=True, intensity_alpha=0.4)
    out = torch.empty((image, image, 4), dtype=torch.float32, device=dev)

    def timed(gv, layout):
        ext = dict(layout=layout, math="strict")
        _, st = mrirt.render_brats(p, gv, labels=gl, out=out, ext=ext, stats=True)
        for _ in range(3):
            mrirt.render_brats(p, gv, labels=gl, out=out, ext=ext)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frames)]
        for e0, e1 in ev:
            e0.record(); mrirt.render_brats(p, gv, labels=gl, out=out, ext=ext); e1.record()
        torch.cuda.synchronize()
        return float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])), st["live_samples"], out.clone()

    gq = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
    ms_quad, _, img_quad = timed(gq, "quad")
    del gq
    g4 = mrirt.upload_mod4(vols, (n, n, n))
    ms, live, img = timed([g4] * 4, "mod4")
    same = bool(torch.equal(img, img_quad))
    alg = live * (4 * BYTES_PER_SAMPLE + 4) + image * image * BYTES_PER_PIXEL      # 8 taps x 4 B x 4 modalities + one label word
    compulsory = g4.nbytes + gl.nbytes + image * image * BYTES_PER_PIXEL
    traffic, on_chip, note = measured_traffic(f"C2:{n}:{image}:{steps}:mod4:strict:4ch+seg")
    hbm = traffic if traffic is not None else compulsory
    return {"workload": f"C2: {n}^3 x 4 modalities + seg overlay, {image}x{image} px, {steps} steps/ray, perspective, no shading (the "
                        "reference's brats_main as the viewer runs it); the four modalities as one float4 MOD4 grid + label cells, "
                        "STRICT math, bit-identical to the oracle",
            "value": round(live / (ms * 1e-3) / 1e6, 1), "unit": "Msamples/s", "ms_per_frame": round(ms, 4), "dtype": "f32",
            "live_samples_per_frame": live, "nominal_samples_per_frame": image * image * steps,
            "four_quad_grids": {"ms_per_frame": round(ms_quad, 4), "same_bits": same,
                                "note": "the same frame from one QUAD grid per modality (1.07 GB of voxels instead of 268 MB): round 3's binding"},
            "roofline": {"bound": "hbm", "achieved": round(hbm / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "basis": ("rocprofv3 PMC HBM bytes per launch, " + str(note)) if traffic is not None
                                  else ("compulsory bytes (the two grids once + framebuffer): a floor; " + str(note)),
                         "on_chip": on_chip, "kernel": "brats_march_pipe_kernel<strict, MOD4, label cells>",
                         "algorithmic_GBs": round(alg / (ms * 1e-3) / 1e9, 1), "algorithmic_bytes_per_launch": alg,
                         "compulsory_bytes_per_launch": compulsory}}


def cpu_baseline(params, vol, ext, rows, n_image):
    """The C/OpenMP oracle (oracle/oracle_c.c) timed on the host cores, on a band of rows of the
    same frame.  A reported baseline, not the target."""
    from oracle import oracle_c
    oracle_c.lib()
    okeys = ("cameraMode", "orthoHalfHeight", "shadeMode", "ka", "kd", "ks", "specPow2", "gradEps")
    oext = {k: v for k, v in ext.items() if k in okeys}

    last = {}

    def run(r0, r1):
        t = time.perf_counter()
        img, aux = oracle_c.brats_main(params, [vol], None, None, oext, return_aux=True, rows=(r0, r1))
        dt = time.perf_counter() - t
        last.update(frame=img, rows=(r0, r1), live=aux["live_samples"])
        return aux["live_samples"], dt

    mid = n_image // 2
    if rows <= 0:      # calibrate on 16 central rows, then size the sample for ~12 s of wall time
        _, dt = run(mid - 8, mid + 8)
        rows = int(min(n_image, max(16, 16 * 12.0 / max(dt, 1e-3))))
    r0 = max(0, mid - rows // 2)
    r1 = min(n_image, r0 + rows)
    reps, live, dt = 0, 0, 0.0
    while reps < 64 and dt < 10.0:       # a many-core host finishes the whole frame in ~1 s: repeat it
        l, d = run(r0, r1)
        live, dt, reps = live + l, dt + d, reps + 1
        if r1 - r0 < n_image:
            break
    cores = int(os.environ.get("OMP_NUM_THREADS", 0)) or (os.cpu_count() or 1)
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown CPU")
    except OSError:
        cpu_model = "unknown CPU"
    return {"value": round(live / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"rows {r0}..{r1 - 1} of the same {n_image}x{n_image} frame x {reps} pass(es) "
                      f"({live} live samples in {dt:.1f} s; C/OpenMP oracle oracle/oracle_c.c, {cores} threads on {cpu_model})"}, last


def parity_block(gpu_frame, gpu_live, oracle_band, n_image):
    """The frame the timed kernel renders against the oracle rows the CPU leg has just rendered: the same
    synthetic volume, camera and transfer function; STRICT math is held to bit equality in the tests, the tolerance
    BASELINE.json states is 1e-4 max-abs.  ``live_equal`` compares the kernel's own live-sample counter with the
    oracle's when the band is the whole frame (a band has no GPU-side counter of its own)."""
    r0, r1 = oracle_band["rows"]
    ref = oracle_band["frame"]
    got = gpu_frame[r0:r1].cpu().numpy()
    err = float(np.abs(got - ref).max())
    whole = r0 == 0 and r1 == n_image
    return {"against": "oracle/oracle_c.c (CPU restatement of inr/viewer/brats_rt.slang:85-168 + the build-defined gradient shading), "
                       "rendered in this run on the host", "rows": int(r1 - r0), "row_range": [int(r0), int(r1)],
            "max_abs_err": err, "bit_identical": bool(np.array_equal(got, ref)), "tolerance": 1e-4,
            "live_equal": (int(gpu_live) == int(oracle_band["live"])) if whole else None,
            "live_samples_oracle": int(oracle_band["live"]) if whole else None}


def scaling_model(dev, grid, n, march_steps, ext, tile, alpha, reps=10, skew_auto=True):
    """What ONE GPU can say about 1 -> 8 scaling, clearly an emulation: BASELINE config 4
    (2048^2 px of the same volume) cut into `tile`^2 tiles dealt round-robin; for N = 1, 2, 4, 8 EVERY rank's share is
    rendered on this GPU (HIP events, `reps` launches each) and the slowest rank is what an N-GPU frame would wait for.
    ``pct_of_linear`` = share(1) / (N x share(N)).  The exchange is modelled, not measured: each peer's compact tiles
    cross its own xGMI link to the root once (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU; 0.75 of that assumed for a
    single RCCL gather), in parallel over the links, and overlap the next frame's march; the root's de-tiling kernel and
    local copy are measured here."""
    import torch
    import mrirt
    from mrirt import synth, tiles
    image = 2048
    ext = _tagged(ext)
    p = synth.brats_scene(n, image, march_steps, channels=1, intensity_alpha=alpha)
    share, worst_rank, mean_share, skews, flight = {}, {}, {}, {}, {}
    ss = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for world in (1, 2, 4, 8):
        times, times3 = [], []
        skews[world] = tiles.balanced_skew(image, tile, world) if skew_auto else 0
        for r in range(world):
            e = tiles.shard_ext(ext, r, world, tile, skews[world])
            out = mrirt.render_brats(p, [grid], ext=e)
            mrirt.render_brats(p, [grid], out=out, ext=e)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                mrirt.render_brats(p, [grid], out=out, ext=e)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
            # the same share as the N>1 frame loop runs it: three frames in flight, each on its own stream
            outs = [out, torch.empty_like(out), torch.empty_like(out)]
            for i in range(3):
                mrirt.render_brats(p, [grid], out=outs[i], ext=e, stream=ss[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3 * reps):
                mrirt.render_brats(p, [grid], out=outs[i % 3], ext=e, stream=ss[i % 3])
            torch.cuda.synchronize()
            times3.append((time.perf_counter() - t0) / (3 * reps) * 1e3)
            del out, outs
        flight[world] = max(times3)
        share[world] = max(times)
        mean_share[world] = float(np.mean(times))
        worst_rank[world] = int(np.argmax(times))
    # the root's part of the exchange: local copy of its own tiles + the de-tiling kernel over the gathered buffer
    world = 8
    max_local = tiles.local_tile_count(image, image, tile, 0, world)
    gathered = torch.zeros((world, max_local, tile, tile, 4), device=dev)
    frame = torch.empty((image, image, 4), device=dev)
    for _ in range(2):
        mrirt.detile(gathered, image, image, tile, world, out=frame)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mrirt.detile(gathered, image, image, tile, world, out=frame)
    e1.record()
    torch.cuda.synchronize()
    detile_ms = e0.elapsed_time(e1) / reps
    link_gbs, eff = 153.0, 0.75
    wire = {w: (0.0 if w == 1 else tiles.local_tile_count(image, image, tile, 0, w) * tile * tile * 16 / (link_gbs * eff * 1e9) * 1e3)
            for w in (1, 2, 4, 8)}
    return {"emulated_on_one_gpu": True,
            "workload": f"C4: {n}^3 volume, {image}x{image} px, {march_steps} steps/ray, {tile}x{tile} tiles dealt along diagonals (tileSkew); "
                        "every rank's share rendered on this one GPU, slowest rank per world size",
            "tile_skew": {str(w): skews[w] for w in skews},
            "share_ms": {str(w): round(share[w], 4) for w in share}, "slowest_rank": {str(w): worst_rank[w] for w in worst_rank},
            "mean_rank_share_ms": {str(w): round(mean_share[w], 4) for w in mean_share},
            "pct_of_linear": {str(w): round(100.0 * share[1] / (w * share[w]), 1) for w in share},
            "share_ms_three_frames_in_flight": {str(w): round(flight[w], 4) for w in flight},
            "pct_of_linear_three_frames_in_flight": {str(w): round(100.0 * flight[1] / (w * flight[w]), 1) for w in flight},
            "in_flight_note": "per-frame time of a rank's share when the frame loop keeps three frames in flight on three HIP streams (what bench.py --gpus N "
                              "does): a share of one or two rounds of packets is mostly fill and drain, which consecutive frames hide for each other",
            "exchange_wire_ms_model": {str(w): round(wire[w], 4) for w in wire},
            "exchange_model": f"per peer: its compact tiles over one xGMI link at {link_gbs} GB/s x {eff} (assumed), peers in parallel; "
                              "asynchronous: overlaps the next frame's march (FrameExchange)",
            "root_detile_ms": round(detile_ms, 4),
            "modelled_frame_ms": {str(w): round(max(share[w], wire[w] + (detile_ms if w > 1 else 0.0)), 4) for w in share},
            "note": "NOT a multi-GPU measurement: bench.py --gpus N on an N-GPU node measures the real thing (RCCL gather, all ranks)"}


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if (a.backend == "gloo") != a.dry_run:
        raise SystemExit("--backend gloo and --dry-run go together: the renderer has no CPU path")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))                      # parent of the N ranks: no GPU call was made here

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry_run:
        sys.exit(dry_run(a, world, rank))

    import torch
    import torch.distributed as dist
    import mrirt
    from mrirt import synth, tiles
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ray-marcher has no CPU path")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grouped = world > 1 or a.force_exchange
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = a.volume
    image = a.image or (1024 if world == 1 else 2048)            # C3 at one GPU, C4 (fixed image: strong scaling) beyond
    cfg_name = {1024: "C3", 2048: "C4"}.get(image, "C3-like") if not a.no_shade else "K1-reference"
    vol = synth.synth_volume(n)                                   # host, deterministic
    params = synth.brats_scene(n, image, a.march_steps, channels=1, intensity_alpha=a.alpha)
    if a.layout == "auto":
        a.layout = "quad" if a.no_shade else "vga"
    ext = {} if a.no_shade else dict(synth.SHADE_EXT)
    ext.update(math=a.math, layout=a.layout, kernelVariant=a.variant)
    grid = mrirt.upload_grid(vol, (n, n, n), a.layout)            # resident in HBM (bricked on device)
    torch.cuda.synchronize()

    # DEPTH frames in flight in the N>1 loop: each on its own HIP stream, so that frame k+1's march starts while frame k drains (a
    # launch of one or two rounds of packets is mostly fill and drain: profiles/r04_tile_share) and frame k's gather overlaps both
    if grouped:
        skew = tiles.balanced_skew(image, a.tile, world) if a.tile_skew < 0 else a.tile_skew
        my_ext = tiles.shard_ext(ext, rank, world, a.tile, skew)
        # asynchronous exchange, DEPTH slots: frame k's gather to rank 0 overlaps the marches of frames k+1, k+2
        ex = tiles.FrameExchange(image, image, a.tile, torch.float32, dev, depth=DEPTH, dst=0, skew=skew)
        local = ex.local(0)
        slot_streams = [ex.stream(i) for i in range(DEPTH)]
    else:
        my_ext, ex = ext, None
        local = torch.empty((image, image, 4), dtype=torch.float32, device=dev)
        slot_streams = None

    def run_frames(count, events=None):
        """`count` frames back to back; every frame is fully exchanged and de-tiled on rank 0
        before this returns (the caller synchronises).  N = 1: one stream, one frame after the other (the headline number is
        a single frame's time).  N > 1: frame s runs on slot s % DEPTH's stream — its wait for the slot's previous exchange, its
        march and its gather are ordered on that stream; frames of different slots overlap."""
        import contextlib
        for s in range(count):
            slot = s % DEPTH
            with (torch.cuda.stream(slot_streams[slot]) if ex is not None else contextlib.nullcontext()):
                if ex is not None and s >= DEPTH:
                    ex.finish(slot)                   # frame s-DEPTH used this slot: wait + de-tile it
                buf = ex.local(slot) if ex is not None else local
                if events is not None:
                    events[s][0].record()
                mrirt.render_brats(params, [grid], out=buf, ext=my_ext)
                if events is not None:
                    events[s][1].record()
                if ex is not None:
                    ex.submit(slot)
        if ex is not None:
            for s in range(max(0, count - DEPTH), count):
                with torch.cuda.stream(slot_streams[s % DEPTH]):
                    ex.finish(s % DEPTH)

    # untimed: sample accounting by the kernel's own counters
    _, st = mrirt.render_brats(params, [grid], out=local, ext=my_ext, stats=True)
    counts = torch.tensor([st["live_samples"], st["shaded_samples"]], dtype=torch.int64, device=dev)
    per_rank = [counts.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, counts)
        counts = torch.stack(per_rank).sum(0)
    live, shaded = int(counts[0]), int(counts[1])

    run_frames(a.warmup)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    # HIP events on the launch stream bracket the march kernel alone (roofline);
    # the wall clock brackets the whole K steps incl. every exchange and de-tiling (value)
    run_frames(a.steps, ev)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if world > 1:
        e = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e[0])
    kernel_ms = float(np.mean([s.elapsed_time(t) for s, t in ev]))
    if grouped:
        # DEPTH frames are in flight: a frame's events bracket time it shares with its neighbours.  What a frame costs this rank is
        # the loop's own pace.
        kernel_ms = elapsed_local / a.steps * 1e3
    kernel_ms_all = [kernel_ms]
    if world > 1:
        km = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
        kms = [km.clone() for _ in range(world)]
        dist.all_gather(kms, km)
        kernel_ms_all = [float(k[0]) for k in kms]

    # untimed, after the clock has stopped: the exchange step by itself (gather of one frame's tiles to rank 0
    # + de-tiling), serialised — what a frame would pay if nothing overlapped it
    exchange = None
    if grouped:
        reps = 5
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            ex.submit(0)
            ex.finish(0)
            torch.cuda.synchronize()
        dist.barrier()
        ex_ms = (time.perf_counter() - t1) / reps * 1e3
        exchange = {"collective": "gather to rank 0 (RCCL, torch.distributed) + de-tiling kernel",
                    "bytes_to_root_per_frame": int((world - 1) * ex.max_local * a.tile * a.tile * 16),
                    "serial_ms_per_frame": round(ex_ms, 4),
                    "note": f"in the timed loop this step is asynchronous: {DEPTH} frames in flight, each on its own stream — frame k's gather "
                            "overlaps the marches of the next frames"}

    if rank == 0:
        value = live * a.steps / elapsed / 1e6
        # roofline of the dominant kernel on THIS rank's launch
        my_live, my_shaded = st["live_samples"], st["shaded_samples"]
        px = local.numel() // 4
        alg_bytes = my_live * BYTES_PER_SAMPLE + my_shaded * BYTES_PER_SHADED + px * BYTES_PER_PIXEL
        # the volume once + this rank's framebuffer (vga holds three copies of the voxels; a frame reads essentially one)
        compulsory = grid.nbytes // (3 if a.layout == "vga" else 1) + px * BYTES_PER_PIXEL
        profiled = world == 1 and a.variant == 0 and a.alpha == 16.0 and not a.force_exchange
        key = (f"{'C3' if not a.no_shade else 'K1'}:{n}:{image}:{a.march_steps}:{a.layout}:{a.math}:"
               f"{'shade' if not a.no_shade else 'plain'}")
        traffic, on_chip, note = measured_traffic(key) if profiled else (None, None, "not a profiled configuration")
        hbm_bytes = traffic if traffic is not None else compulsory
        achieved = hbm_bytes / (kernel_ms * 1e-3) / 1e9
        # The roof that binds this kernel: every live sample's taps — 8 x 16 B shaded, 2 x 16 B
        # on QUAD grids — have to reach a lane's registers through the texture data-return path, 64 B per clock per CU.  With
        # the clock the part held under the profiler this is the time the return path alone needs, as a fraction of the
        # kernel's time; HBM is NOT the limiter (its fraction is the line's `frac`, kept because the metric names it).
        tap_bytes = (8 if not a.no_shade else 2) * 16 if a.layout in ("vg", "vga", "quad") else 8 * 4 * (7 if not a.no_shade else 1)
        if on_chip is not None and on_chip.get("effective_clock_ghz"):
            on_chip = dict(on_chip)
            floor_ms = my_live * tap_bytes / (64.0 * 256 * on_chip["effective_clock_ghz"] * 1e9) * 1e3
            on_chip["td_return_frac"] = round(floor_ms / kernel_ms, 4)
            on_chip["td_return_floor_ms"] = round(floor_ms, 4)
            on_chip["td_return_basis"] = (f"{my_live} live samples x {tap_bytes} B of taps / (64 B/clk/CU x 256 CUs x {on_chip['effective_clock_ghz']} GHz "
                                          "held under the profiler) / kernel time")
        out = {
            "metric": "live Msamples/s, 512^3 fp32 volume @ 1024^2 x 512 steps (gradient shading + ERT)"
                      + ("" if image == 1024 else f" [this run: {image}^2]"),
            "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_name +
                       f": {n}^3 fp32 volume, {image}x{image} px, {a.march_steps} steps/ray, perspective, "
                       + ("central-difference gradient + Blinn-Phong + " if not a.no_shade else "") + "ERT"
                       + ("" if world == 1 else f"; image fixed as N grows, tiles over {world} ranks"),
                       "math": a.math, "layout": a.layout, "intensityAlpha": a.alpha,
                       "tiles": f"{a.tile}x{a.tile} dealt round-robin over {world} rank(s)" + (f", rows rotated by tileSkew {my_ext.get('tileSkew', 0)} (diagonal deal)" if grouped else ""),
                       "live_samples_per_frame": live, "shaded_samples_per_frame": shaded,
                       "nominal_samples_per_frame": image * image * a.march_steps,
                       "effective_Msamples_s_nominal": round(image * image * a.march_steps * a.steps / elapsed / 1e6, 1),
                       "kernel_variant": a.variant},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "basis": (("rocprofv3 PMC HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE), " + str(note)) if traffic is not None
                                   else ("compulsory bytes (volume once + framebuffer): a floor of the real traffic; " + str(note)))
                                  + "; the HBM PEAK is not what limits this kernel — two roofs 12 % apart do: the texture data-return path (on_chip.td_return_frac, "
                                    "l1_hit_rate; reached on an Infinity-Cache-resident volume) and, on this volume, the latency of the lines that come from beyond L2 "
                                    "(the FAST-math build keeps the full 2.39 GHz where STRICT is power-throttled to 2.26 and takes the same time: profiles/r04_power_clocks/)",
                         # what actually binds the kernel (PMC passes): its gathers are served by L1/L2
                         "on_chip": on_chip,
                         "kernel": "brats_march_pipe_kernel" if not a.variant & 4 else "brats_march_kernel", "kernel_ms": round(kernel_ms, 4),
                         # SURVEY 8(d)'s accounting: bytes the taps ask for, mostly served by the caches (can exceed the HBM peak)
                         "algorithmic_GBs": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "compulsory_bytes_per_launch": compulsory,
                         "bytes_per_sample": BYTES_PER_SAMPLE + (0 if a.no_shade else BYTES_PER_SHADED)},
        }
        if world > 1 or a.force_exchange:
            out["per_rank"] = {"live_samples": [int(c[0]) for c in per_rank], "march_kernel_ms": [round(k, 4) for k in kernel_ms_all],
                               "march_kernel_ms_is": f"this rank's wall time per frame of the timed loop ({DEPTH} frames in flight on {DEPTH} streams: "
                                                     "per-launch durations overlap and are not reported)"}
            out["exchange"] = exchange
        if world == 1 and not a.force_exchange and not a.no_pipelined:
            out["pipelined"] = pipelined_frames(dev, params, grid, ext, live)
        if world == 1 and not a.no_k1:
            out["k1_reference_path"] = k1_reference_path(dev)
        if world == 1 and not a.no_inr:
            out["inr_path"] = inr_path(dev)
        if world == 1 and not a.no_scaling_model and not a.no_shade and not a.force_exchange:
            out["scaling_model"] = scaling_model(dev, grid, n, a.march_steps, ext, a.tile, a.alpha)
        if not a.no_cpu_baseline and a.cpu_rows != 0 and world == 1:
            out["cpu_baseline"], band = cpu_baseline(params, vol, ext, min(a.cpu_rows, image), image)
            if not grouped:
                # the frame of the timed configuration, rendered once more (untimed) and held to the oracle's rows
                frame, stp = mrirt.render_brats(params, [grid], ext=ext, stats=True)
                out["parity"] = parity_block(frame, stp["live_samples"], band, image)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
