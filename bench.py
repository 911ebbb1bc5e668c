#!/usr/bin/env python
"""Headline benchmark: VO frame-pairs/s at 640x480 (RAFT-Stereo + RAFT-Flow + pose).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One STEP = one pass of the whole hot path over one chunk of ``--batch`` consecutive
synthetic stereo frame pairs that are already resident in HBM: RAFT-Stereo on the
chunk's new frames, disparity->depth, RAFT-Flow on every consecutive pair, keypoint
lift / flow-warp / filters, Kabsch pose per pair.  Each rank (one process per GPU)
works on its own shard of the sequence; there is no data-path collective (the
4x4 poses are gathered once per sequence, outside the per-chunk hot path), so
scaling is weak and ``value`` is the sum over ranks.

Prints ONE JSON line (rank 0).  Besides the driver's fields it carries
  roofline      the dominant hand-written kernel: algorithmic bytes or FLOPs per
                launch / its mean device time per launch, measured with events on the
                launch stream inside the timed region
  kernels       the same figures for every hand-written kernel
  cpu_baseline  the CPU oracle (stock-PyTorch RAFT restatement + NumPy geometry with
                the reference's per-point loop) timed on this box's host cores on a
                bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec
SUSTAINED_16BIT_TFLOPS = {"f16x3": 1560.0, "bf16x3": 1670.0}   # measured, random operands (power-limited clock)
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0 # v_mfma_f32_32x32x16_bf16 dense peak
# kernels whose float32 products are three bf16 matrix-core terms: their recorded FLOPs are the bf16
# multiply-adds they issue (three per product), priced against the bf16 peak
SPLIT_BF16_KERNELS = ("vo_conv3x3_bf16x3", "vo_conv5_bf16x3", "vo_conv3x3s2_bf16x3", "vo_conv3x3s2_shortcut_bf16x3", "vo_conv1x1s2_bf16x3",
                      "vo_conv7x7_bf16x3",
                      "vo_corr_volume_2d_bf16x3", "vo_corr_volume_1d_bf16x3", "vo_corr_volume_2d_presplit")


DTYPE_TEXT = {
    "f32": "f32 (every product on the float32 matrix instruction; Winograd forms for the 3x3 and 1x5/5x1 layers)",
    "f16x3": "f32 (tensors, accumulation and every other operator in float32; each product of the 3x3, 1x5/5x1, stride-2, "
             "many-channel 1x1 and 7x7 convolutions (GRU gate convolutions included) is three float16 matrix-core terms on "
             "operands cut into two float16 pieces: 22 significant bits per operand, exact power-of-two scaling for the "
             "exponent range, |x| > 65504 -> NaN; measured error of a layer = the float32 matrix instruction's; the "
             "correlation volumes likewise, each image's feature map scaled by the power of two of its largest |x| before "
             "the cut; see f32_mfma_only and bf16x3 beside it)",
    "bf16x3": "f32 tensors and accumulation; heavy products (convolutions and correlation volumes) from three bf16 "
              "matrix-core terms, 16 significant bits per operand: narrower than the reference's float32",
}


def algorithmic_work(B: int, H: int, W: int, iters: int, n_kp: int, D: int = 256):
    """Per-STEP algorithmic bytes / FLOPs and launch counts of every hand-written
    kernel for B frame pairs (= B new stereo frames + B flow pairs)."""
    h8, w8 = H // 8, W // 8
    n = h8 * w8
    h4, w4 = H // 4, W // 4
    lv2 = [(h8 >> i) * (w8 >> i) for i in range(4)]
    lv1 = [w4 >> i for i in range(4)]
    f = 4  # bytes per float32
    return {
        "vo_corr_volume_2d": dict(launches=1, flops=2.0 * n * n * D * B, bytes=(2 * n * D + n * n) * f * B),
        "vo_corr_volume_1d": dict(launches=1, flops=2.0 * h4 * w4 * w4 * D * B,
                                  bytes=(2 * h4 * w4 * D + h4 * w4 * w4) * f * B),
        # split-bf16 products: three bf16 multiply-adds per float32 product
        "vo_corr_volume_2d_bf16x3": dict(launches=1, flops=6.0 * n * n * D * B, bytes=(2 * n * D + n * n) * f * B),
        "vo_corr_volume_1d_bf16x3": dict(launches=1, flops=6.0 * h4 * w4 * w4 * D * B,
                                         bytes=(2 * h4 * w4 * D + h4 * w4 * w4) * f * B),
        # the same volume from feature rows split beforehand (the flow network's default under the split arithmetic)
        "vo_corr_volume_2d_presplit": dict(launches=1, flops=6.0 * n * n * D * B, bytes=(2 * n * D + n * n) * f * B),
        "vo_corr_pyramid_2d": dict(launches=1, bytes=B * n * sum(lv2) * f),
        "vo_corr_pyramid_1d": dict(launches=1, bytes=B * h4 * w4 * sum(lv1) * f),
        # coords + window out + the (2r+2)^2 patch each pixel needs per level
        "vo_corr_lookup_2d": dict(launches=iters, bytes=iters * B * n * (2 + 4 * 81 + 4 * 100) * f),
        "vo_corr_lookup_1d": dict(launches=iters, bytes=iters * B * h4 * w4 * (2 + 4 * 9 + 4 * 10) * f),
        "vo_convex_upsample": dict(launches=2, bytes=B * f * ((576 + 2) * n + 2 * H * W + (144 + 2) * h4 * w4 + H * W)),
        "vo_disparity_to_depth": dict(launches=1, bytes=B * H * W * 2 * f),
        "vo_pair_correspondences": dict(launches=1, bytes=B * n_kp * (16 + 8 + 8 + 48 + 1)),
        "vo_kabsch_batch": dict(launches=1, bytes=B * (n_kp * 49 + 128)),
    }


PMC_KEYS = {"vo_corr_volume_2d": "corr2d", "vo_corr_volume_1d": "corr1d", "vo_corr_pyramid_2d": "pyr2d",
            "vo_corr_pyramid_1d": "pyr1d", "vo_corr_lookup_2d": "lookup2d", "vo_corr_lookup_1d": "lookup1d",
            "vo_instance_norm_relu": "instnorm", "vo_conv3x3_winograd": "winograd", "vo_conv3x3_winograd4": "winograd4", "vo_conv5_winograd": "winograd1d",
            "vo_conv3x3_bf16x3": "conv3x3_bf16x3", "vo_conv3x3s2_bf16x3": "conv3x3s2_bf16x3", "vo_conv3x3s2_shortcut_bf16x3": "conv3x3s2_bf16x3", "vo_conv7x7_bf16x3": "conv7x7_bf16x3",
            "vo_corr_volume_2d_bf16x3": "corr2d_bf16x3", "vo_corr_volume_1d_bf16x3": "corr1d_bf16x3",
            "vo_corr_volume_2d_presplit": "corr2d_presplit",
            "vo_instance_norm_apply": "instnorm_apply", "vo_conv1x1": "conv1x1"}


def pmc_traffic(kernel: str, B: int, H: int, W: int, arithmetic: str = "f16x3"):
    """HBM-side bytes per launch from the COMMITTED rocprofv3 --pmc passes over this same workload
    (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs; FETCH_SIZE doubled per
    the gfx950 correction for wide reads) -- read from that file, not measured in this run --, or None when no
    pass exists for this kernel, shape and arithmetic."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path) or kernel not in PMC_KEYS:
        return None
    meta = json.load(open(path))
    if (meta.get("batch_pairs"), meta.get("height"), meta.get("width"), meta.get("conv_arithmetic", "bf16x3")) != (B, H, W, arithmetic):
        return None
    k = meta["kernels"].get(PMC_KEYS[kernel])
    if not k:
        return None
    # gathered reads (the lookups) are counted per touched 128-byte line, without the halving of wide streaming reads
    fetch_factor = 1.0 if PMC_KEYS[kernel] in ("lookup2d", "lookup1d") else 2.0
    return (fetch_factor * k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024.0


def mixed_precision_leg(vo32, vo16, left, right, cached, steps):
    """Secondary measurement, never the headline: the same step with float16 activations in
    the conv nets (the published models' mixed_precision switch, which the reference leaves
    off), its throughput and its deviation from the float32 path on the same frames."""
    with torch.no_grad():
        f32, d32 = vo32.flows(left), vo32.disparity_solver.compute_disparity_gray_batch(left[1:], right[1:])
        f16, d16 = vo16.flows(left), vo16.disparity_solver.compute_disparity_gray_batch(left[1:], right[1:])
    epe = (f16 - f32).pow(2).sum(1).sqrt().mean().item()
    derr = (d16 - d32).abs().mean().item()
    del f32, d32, f16, d16
    for _ in range(2):
        vo16.process_chunk(left, right, cached)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vo16.process_chunk(left, right, cached)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": (left.shape[0] - 1) / dt, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt,
            "dtype": "f16 activations, f32 weights/correlation/geometry",
            "flow_epe_px_vs_f32": epe, "disparity_err_px_vs_f32": derr, "budget_px": 0.05,
            "note": "not the reference's precision (mixed_precision=False there); reported beside, not as, the metric"}


def progress(msg):
    """One line on stderr per phase: a harness watching for a hung run sees the bench move (stdout stays one JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _time(fn, repeat):
    fn()                                   # warm-up
    t0 = time.perf_counter()
    for _ in range(repeat):
        fn()
    return (time.perf_counter() - t0) / repeat


def cpu_baseline(vo, left, right, iters, threads, stereo_name):
    """The CPU oracle (kind "port": the stock-PyTorch restatement of RAFT / RAFT-Stereo in oracle/raft_modules.py
    carrying the GPU run's weights, + NumPy geometry with the reference's per-point loop) on this box's host cores, on
    a bounded sample of the same workload.  The headline figure: 20 steady-state frame pairs (1 new stereo frame + 1
    flow pair + geometry each, four pairs per forward pass) after 3 warm-up pairs, all ``threads`` threads, with the
    argument set of the GPU run; beside it the same for the reference's default (full) RAFT-Stereo arguments, one pair
    on one thread for both, and the geometry / pose / composition stages on their own."""
    from oracle import geometry_ref, raft_modules
    flow_m = raft_modules.flow_twin(vo.flow_solver.model, "cpu")
    stereo_models = {stereo_name: raft_modules.stereo_twin(vo.disparity_solver.model, "cpu")}
    other = "full" if stereo_name == "realtime" else "realtime"
    torch.manual_seed(1)
    stereo_models[other] = (raft_modules.RAFTStereo() if other == "full"
                            else raft_modules.RAFTStereo(hidden_dims=(64, 64, 64), n_gru_layers=2)).eval()
    rgb = lambda a: a.float()[:, None].expand(-1, 3, -1, -1).contiguous().cpu()
    group, n_groups, n_warm = 4, 5, 3
    l, r = rgb(left[:group + 1]), rgb(right[:group + 1])
    small = torch.rand(1, 3, 128, 160) * 255
    K_inv = vo.pts_xform.K_inv
    kp = vo.pts_src.get_keypoints(None, vo.cfg.max_keypoints)
    H, W = left.shape[1:]

    def networks(stereo_m, n):
        """n steady-state pairs: stereo on frames 1..n, flow on pairs (0,1)..(n-1,n)."""
        t0 = time.perf_counter()
        _, disp = stereo_m(l[1:n + 1], r[1:n + 1], iters)
        t1 = time.perf_counter()
        _, flow = flow_m(l[:n], l[1:n + 1], iters)
        t2 = time.perf_counter()
        progress(f"cpu baseline: {n} pair(s), {torch.get_num_threads()} thread(s): stereo {t1 - t0:.1f}s, flow {t2 - t1:.1f}s")
        return t1 - t0, t2 - t1, disp, flow

    torch.set_num_threads(threads)
    flow_m(small, small, 1)                                           # thread-pool / allocator warm-up
    legs = {}
    for name, stereo_m in stereo_models.items():
        stereo_m(small, small, 1)
        networks(stereo_m, n_warm)
        ts = tf = 0.0
        for _ in range(n_groups):
            a, b, disp, flow = networks(stereo_m, group)
            ts, tf = ts + a, tf + b
        legs[name] = (ts / (group * n_groups), tf / (group * n_groups), disp, flow)
    t_stereo, t_flow, disp, flow = legs[stereo_name]
    depth = [geometry_ref.disparity_to_depth(disp[i, 0].numpy(), vo._fxB, 1.0) for i in range(2)]
    flow_np = flow.numpy()
    geo = lambda loop: geometry_ref.pair_pose(kp, depth[0], depth[1], flow_np[1], K_inv, loop=loop)
    t_geo_loop = _time(lambda: (geometry_ref.disparity_to_depth(disp[1, 0].numpy(), vo._fxB, 1.0), geo(True)), 5)
    t_geo_vec = _time(lambda: (geometry_ref.disparity_to_depth(disp[1, 0].numpy(), vo._fxB, 1.0), geo(False)), 20)
    per_pair = t_stereo + t_flow + t_geo_loop

    torch.set_num_threads(1)
    single = {}
    for name, stereo_m in stereo_models.items():
        a, b, _, _ = networks(stereo_m, 1)
        single[name] = (a, b)
    torch.set_num_threads(threads)
    per_pair_1 = sum(single[stereo_name]) + t_geo_loop

    # the dense 1280x720 grid (19 928 points): geometry only, synthetic maps
    rng = np.random.default_rng(0)
    Hd, Wd = 720, 1280
    kpd = geometry_ref.uniform_keypoints(np.ones((Hd, Wd), bool), 20000)
    d1 = rng.uniform(0.5, 12, (Hd, Wd)).astype(np.float32); d2 = d1 + rng.normal(0, 0.05, d1.shape).astype(np.float32)
    fl = rng.normal(0, 2, (2, Hd, Wd)).astype(np.float32)
    Kd = np.linalg.inv(np.array([[556.0, 0, 640], [0, 556.0, 360], [0, 0, 1]]))
    t_dense_loop = _time(lambda: geometry_ref.pair_pose(kpd, d1, d2, fl, Kd, loop=True), 1)
    t_dense_vec = _time(lambda: geometry_ref.pair_pose(kpd, d1, d2, fl, Kd, loop=False), 5)
    # Kabsch alone (numpy.linalg.svd per pair, camera/camera_svd_xform.py:31)
    P = rng.uniform(-3, 3, (256, 300, 3)); Q = P + rng.normal(0, 0.01, P.shape)
    t_kabsch_300 = _time(lambda: [geometry_ref.kabsch(P[b], Q[b]) for b in range(256)], 2) / 256
    Pd = rng.uniform(-3, 3, (8, 19928, 3)); Qd = Pd + rng.normal(0, 0.01, Pd.shape)
    t_kabsch_dense = _time(lambda: [geometry_ref.kabsch(Pd[b], Qd[b]) for b in range(8)], 2) / 8
    # host composition of 100 000 relative poses (test_camera_tracking.py:210)
    Tn = np.tile(np.eye(4), (100000, 1, 1))
    t_compose = _time(lambda: geometry_ref.compose_trajectory(Tn), 1)
    try:
        import cv2  # noqa: F401
        sgbm = "cv2 importable but the SGBM leg is not implemented"
    except ImportError:
        sgbm = "unavailable: cv2 is not importable on this box (stereo/stereo_disparity_openCV.py:35-47 cannot run), no substitute"
    o_s, o_f = legs[other][:2]
    return {"value": 1.0 / per_pair, "unit": "frame-pairs/s", "cores": threads, "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"{group * n_groups} steady-state frame pairs {W}x{H} ({group} per forward pass) after {n_warm} warm-up pairs, "
                      f"{iters} iters: RAFT-Stereo({stereo_name} args) {t_stereo:.2f}s + RAFT-Flow {t_flow:.2f}s + "
                      f"geometry (per-point loop) {t_geo_loop * 1e3:.1f}ms per pair, {threads} threads",
            "seconds_per_pair": per_pair,
            "other_stereo_args": {"value": 1.0 / (o_s + o_f + t_geo_loop), "unit": "frame-pairs/s", "cores": threads,
                                  "sample": f"same sample with RAFT-Stereo({other} args, default-initialised weights): "
                                            f"{o_s:.2f}s + RAFT-Flow {o_f:.2f}s per pair"},
            "single_thread": {"value": 1.0 / per_pair_1, "unit": "frame-pairs/s", "cores": 1,
                              "sample": f"1 frame pair (20 would take minutes): RAFT-Stereo({stereo_name}) {single[stereo_name][0]:.1f}s + "
                                        f"RAFT-Flow {single[stereo_name][1]:.1f}s + geometry; RAFT-Stereo({other}) "
                                        f"{single[other][0]:.1f}s",
                              "other_stereo_args_value": 1.0 / (sum(single[other]) + t_geo_loop)},
            "geometry_us_per_pair": {"loop_300pts": 1e6 * t_geo_loop, "vectorised_300pts": 1e6 * t_geo_vec,
                                     "loop_19928pts": 1e6 * t_dense_loop, "vectorised_19928pts": 1e6 * t_dense_vec},
            "kabsch_pairs_per_s": {"n300": 1.0 / t_kabsch_300, "n19928": 1.0 / t_kabsch_dense},
            "compose_100k_poses_ms": 1e3 * t_compose,
            "sgbm_config1": sgbm}


def timed_steps(vo, left, right, cached, steps, warmup):
    for _ in range(warmup):
        vo.process_chunk(left, right, cached)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vo.process_chunk(left, right, cached)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def config_leg(make, left, right, steps, what):
    """A secondary measurement of the same step with another configuration of the reference
    (never the headline): build the pipeline, first chunk untimed (halo depth), then time."""
    vo2 = make()
    B = left.shape[0] - 1
    _, _, cached = vo2.process_chunk(left, right, None)
    dt = timed_steps(vo2, left, right, cached, steps, 1)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    del vo2, cached
    torch.cuda.empty_cache()
    return {"value": B / dt, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt, "pairs_per_step": B, "steps": steps,
            "dtype": "f32", "config": what, "peak_device_memory_GiB": peak}


def mfma_util_from_profile(kernel_key, B, H, W, arithmetic="f16x3"):
    """SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE / 8) of the committed counter pass, or
    None when the pass was taken at another shape."""
    for name in ("mfma_util.json",):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        meta = json.load(open(path))
        k = meta.get("kernels", {}).get(kernel_key)
        if (k and (meta.get("batch_pairs"), meta.get("height"), meta.get("width"), meta.get("conv_arithmetic", "bf16x3")) == (B, H, W, arithmetic)
                and "mfma_busy_frac" in k):
            return {"value": k["mfma_busy_frac"], "batch_pairs_of_the_pass": meta.get("batch_pairs"),
                    "source": "profiles/" + name}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=192, help="frame pairs per step and per GPU")
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--stereo", choices=["realtime", "full"], default="realtime")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--miopen-find", action="store_true", help="let MIOpen benchmark its solvers per conv shape")
    ap.add_argument("--channels-last", action="store_true", help="run the conv nets in NHWC memory format")
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--mixed-precision", action="store_true",
                    help="time the float16-activation path as the headline instead of float32 (not the reference's precision)")
    ap.add_argument("--mixed-precision-leg", action="store_true",
                    help="add the secondary float16-activation measurement (MIOpen's float16 kernels: about two minutes "
                         "of solver set-up on a fresh machine; slower and less exact than the default arithmetic)")
    ap.add_argument("--no-mixed-precision-leg", action="store_true", help="accepted for older scripts: that leg is off by default")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the secondary measurements of the reference's default (full) argument set")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the measurement with frames uploaded from pinned host memory")
    ap.add_argument("--pose", choices=["svd", "ransac"], default="svd")
    ap.add_argument("--conv-arithmetic", choices=["f16x3", "bf16x3", "f32"], default="f16x3",
                    help="how the heavy float32 products are formed: three float16 matrix-core terms (22 significant "
                         "bits per operand, the float32 instruction's error; default), three bf16 terms (16 bits), or "
                         "float32 matrix instructions only")
    args = ap.parse_args()

    # --gpus N without a launcher around this process: start the N ranks here, one fresh process per GPU through
    # torch.distributed.run, BEFORE this process has touched the GPU (it never does: it only relays the ranks' output
    # and exit status).  Under a launcher WORLD_SIZE must agree with --gpus.
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the one JSON line and nothing else: whatever a library writes to file descriptor 1 (gloo announces
    # its ranks there) goes to stderr from here on, the line itself through a duplicate of the original descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    # One rank per GPU over RCCL.  With fewer GPUs than local ranks (a rehearsal of the multi-rank path on a one-GPU
    # box: ranks share the card, use a small --batch) the ranks rendezvous over gloo instead: RCCL refuses two ranks on
    # one device; the hot path has no collective either way.
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    shared_card = n_dev < local_world
    torch.cuda.set_device(local_rank % n_dev)
    dist = None
    if world > 1 or os.environ.get("TORCHELASTIC_RUN_ID"):       # launched by torch.distributed.run (any world size)
        import torch.distributed as dist
        if shared_card:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from visual_odometry_amd import ops, synthetic as syn
    from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
    from visual_odometry_amd.raft.models import RAFTStereoConfig

    H, W, B = args.height, args.width, args.batch
    if shared_card:
        # a rehearsal of the multi-rank path on a box with fewer GPUs than ranks: the ranks share a card, so each takes
        # its share of the chunk (the card's memory holds one full chunk, not one per rank)
        B = max(8, B // -(-local_world // n_dev))
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    def make_vo(mixed, stereo=args.stereo, iters=args.iters, pose=args.pose):
        cfg = VOConfig(iters=iters,
                       stereo=RAFTStereoConfig.realtime() if stereo == "realtime" else RAFTStereoConfig(),
                       camera_offset=syn.CAMERA_OFFSET, mixed_precision=mixed, pose=pose,
                       conv_arithmetic=args.conv_arithmetic)
        return VisualOdometry(cfg, syn.left_K(W / 640.0), syn.focal_length_px() * (W / 640.0), syn.baseline(),
                              np.ones((H, W), bool), seed=0)

    vo = make_vo(args.mixed_precision)
    n_kp = vo.keypoints.shape[0]
    if args.channels_last:
        vo.flow_solver.model.to(memory_format=torch.channels_last)
        vo.disparity_solver.model.to(memory_format=torch.channels_last)

    # synthetic chunk: B+1 consecutive stereo frames, different per rank, resident in HBM
    left, right = syn.stereo_sequence(seed=100 + rank, frames=B + 1, h=H, w=W)
    left = torch.from_numpy(left).cuda()
    right = torch.from_numpy(right).cuda()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    progress(f"{world} rank(s), {B} pairs per step: first chunk and {args.warmup} warm-up step(s)")
    # untimed: first chunk of the shard (computes the halo frame's depth too), then W warm-up steps
    _, _, cached = vo.process_chunk(left, right, None)
    for _ in range(args.warmup):
        vo.process_chunk(left, right, cached)

    rec = {}
    sync_all()
    ops.set_recorder(rec)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T, skipped, _ = vo.process_chunk(left, right, cached)
    torch.cuda.synchronize()
    ops.set_recorder(None)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], device="cpu" if shared_card else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs = world * B * args.steps
    value = pairs / elapsed
    progress(f"timed {args.steps} step(s): {pairs / elapsed:.1f} frame-pairs/s")

    # outside the timed region: the one exchange of the sharded loop, an all-gather of the ranks'
    # relative poses (RCCL when launched by torch.distributed.run), then the host composition
    from visual_odometry_amd.pipeline import compose_trajectory, gather_poses
    torch.cuda.synchronize()
    t_g = time.perf_counter()
    T_all, sk_all = gather_poses(T, skipped, world * B)
    gather_ms = 1e3 * (time.perf_counter() - t_g)
    t_c = time.perf_counter()
    G = compose_trajectory(T_all, sk_all)
    compose_ms = 1e3 * (time.perf_counter() - t_c)
    gather = {"ranks": int(dist.get_world_size()) if dist is not None else 1,
              "backend": dist.get_backend() if dist is not None else None,
              "poses_gathered": int(T_all.shape[0]), "gather_ms": gather_ms, "compose_ms": compose_ms,
              "trajectory_is_rigid": bool(np.abs(G[-1][:3, :3] @ G[-1][:3, :3].T - np.eye(3)).max() < 1e-9)}

    # per-kernel device time from the events recorded inside the timed region
    work = algorithmic_work(B, H, W, args.iters, n_kp)
    kernels = {}
    for name, evs in rec.items():
        ms = [ev[0].elapsed_time(ev[1]) for ev in evs]
        w = dict(work.get(name, {}))
        recorded = sum(ev[2] for ev in evs)
        if recorded and "bytes" not in w:          # kernels whose work depends on the tensor they are handed
            w = {"launches": len(evs) // args.steps, "bytes": recorded / args.steps}
            if sum(ev[3] for ev in evs):
                w["flops"] = sum(ev[3] for ev in evs) / args.steps
        per_step_ms = sum(ms) / args.steps
        entry = {"launches_per_step": len(ms) // args.steps, "avg_us": 1e3 * sum(ms) / len(ms),
                 "ms_per_step": per_step_ms}
        if "bytes" in w:
            entry["alg_GB_per_launch"] = w["bytes"] / w["launches"] / 1e9
            entry["GBps"] = w["bytes"] / (per_step_ms * 1e-3) / 1e9
        if "flops" in w:
            entry["TFLOPs"] = w["flops"] / (per_step_ms * 1e-3) / 1e12
        kernels[name] = entry
    work = {k: (work[k] if k in work else {"launches": v["launches_per_step"],
                                            "bytes": v.get("alg_GB_per_launch", 0) * 1e9 * v["launches_per_step"],
                                            **({"flops": v["TFLOPs"] * 1e12 * v["ms_per_step"] * 1e-3} if "TFLOPs" in v else {})})
            for k, v in kernels.items()}
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    d = kernels[dom]
    if "TFLOPs" in d:
        peak = BF16_MFMA_PEAK_TFLOPS if dom in SPLIT_BF16_KERNELS else F32_MFMA_PEAK_TFLOPS
        roofline = {"kernel": dom, "bound": "mfma", "achieved": d["TFLOPs"], "peak": peak,
                    "unit": "TFLOP/s", "frac": d["TFLOPs"] / peak, "traffic": None,
                    "avg_launch_us": d["avg_us"], "hbm_GBps": d.get("GBps")}
        if dom in SPLIT_BF16_KERNELS:
            roofline["float32_equivalent_TFLOPs"] = d["TFLOPs"] / 3.0
            roofline["note"] = ("mean over all launches of one step; achieved counts the 16-bit multiply-adds the kernel issues "
                                "(three per float32 product: xl*wh + xh*wl + xh*wh; float16 or bf16 pieces, the same instruction "
                                "rate), peak is the dense 16-bit matrix rate; float32_equivalent_TFLOPs = the direct "
                                "convolution's own FLOPs per second (f32 MFMA peak: 157.3)")
    else:
        roofline = {"kernel": dom, "bound": "hbm", "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["GBps"] / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": d["avg_us"]}
    AR = args.conv_arithmetic
    if dom in SPLIT_BF16_KERNELS and AR in SUSTAINED_16BIT_TFLOPS:
        # the nominal peak assumes 2.4 GHz; on operands that toggle the multipliers the chip holds far less (power limit)
        roofline["sustained_peak_random_operands"] = SUSTAINED_16BIT_TFLOPS[AR]
        roofline["frac_of_sustained"] = d["TFLOPs"] / SUSTAINED_16BIT_TFLOPS[AR]
        roofline["sustained_peak_source"] = ("profiles/r03_power_limit.txt: the bare matrix instruction from registers on random "
                                             "operands (tools/mfma_f16_vs_bf16.hip), committed with the repository")
    roofline["traffic"] = pmc_traffic(dom, B, H, W, AR)
    roofline["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this workload "
                                  "(tools/collect_profiles_r03.sh), committed with the repository, not measured in this run")
    if dom in PMC_KEYS and mfma_util_from_profile(PMC_KEYS[dom], B, H, W, AR):
        roofline["mfma_busy_pmc"] = mfma_util_from_profile(PMC_KEYS[dom], B, H, W, AR)
    if dom in ("vo_conv3x3_winograd", "vo_conv3x3_winograd4"):
        # flops are the multiply-adds the kernel issues in the Winograd domain: 16 per 2x2 outputs
        # (4/9 of the direct form) for F(2x2,3x3), 36 per 4x4 outputs (1/4) for F(4x4,3x3)
        roofline["direct_equivalent_TFLOPs"] = d["TFLOPs"] * (2.25 if dom == "vo_conv3x3_winograd" else 4.0)
        roofline["note"] = "mean over all 3x3 layers of one step; achieved counts the multiply-adds issued in the Winograd domain, traffic is the mean over the same launches"
    # the north star's one dense contraction, reported whether or not it dominates
    ck = kernels.get("vo_corr_volume_2d")
    roofline_corr = None
    if ck:
        roofline_corr = {"kernel": "vo_corr_volume_2d", "bound": "mfma", "achieved": ck["TFLOPs"],
                         "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ck["TFLOPs"] / F32_MFMA_PEAK_TFLOPS,
                         "avg_launch_us": ck["avg_us"], "traffic": pmc_traffic("vo_corr_volume_2d", B, H, W, AR),
                         "mfma_util_pmc": mfma_util_from_profile("corr2d", B, H, W, AR)}
    cs = kernels.get("vo_corr_volume_2d_bf16x3")
    if cs:
        # with bf16 terms the contraction's bound is the store of the volume, not the matrix cores
        roofline_corr = {"kernel": "vo_corr_volume_2d_bf16x3", "bound": "hbm", "achieved": cs["GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": cs["GBps"] / HBM_PEAK_GBS, "avg_launch_us": cs["avg_us"],
                         "bf16_TFLOPs": cs["TFLOPs"], "float32_equivalent_TFLOPs": cs["TFLOPs"] / 3.0,
                         "traffic": pmc_traffic("vo_corr_volume_2d_bf16x3", B, H, W, AR),
                         "note": "split-bf16 products (three bf16 matrix-core terms per float32 product); the float32-instruction "
                                 "kernel's figures are in the f32_mfma_only leg's run (--conv-arithmetic f32)"}
    cp = kernels.get("vo_corr_volume_2d_presplit")
    if cp:
        # rows split beforehand, 256 x 256 tiles: the matrix instructions (three per product) are the larger floor
        # (0.91 ms per 64 pairs at the bf16 peak against 0.77 ms for the volume's bytes at the HBM peak)
        roofline_corr = {"kernel": "vo_corr_volume_2d_presplit", "bound": "mfma", "achieved": cp["TFLOPs"], "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": cp["TFLOPs"] / BF16_MFMA_PEAK_TFLOPS, "avg_launch_us": cp["avg_us"],
                         "float32_equivalent_TFLOPs": cp["TFLOPs"] / 3.0, "hbm_GBps": cp["GBps"],
                         "hbm_frac": cp["GBps"] / HBM_PEAK_GBS, "traffic": pmc_traffic("vo_corr_volume_2d_presplit", B, H, W, AR),
                         "mfma_busy_pmc": mfma_util_from_profile("corr2d_presplit", B, H, W, AR),
                         "note": ("three float16 matrix-core terms per float32 product on feature rows cut into float16 pieces "
                                  "beforehand (each image's map scaled by the power of two of its largest |x|: vo_feature_rows_amax + "
                                  "vo_split_feature_rows, listed under kernels); achieved counts the 16-bit multiply-adds issued"
                                  if AR == "f16x3" else
                                  "three bf16 matrix-core terms per float32 product on feature rows split into their bf16 pieces by "
                                  "the encoder's last layer; achieved counts the bf16 multiply-adds issued") +
                                 "; the float32-instruction kernel's figures are in the f32_mfma_only leg's run (--conv-arithmetic f32)"}
    roofline["alg_per_launch"] = (work[dom].get("flops") if roofline["bound"] == "mfma" else work[dom]["bytes"]) / work[dom]["launches"]
    hand_ms = sum(k["ms_per_step"] for k in kernels.values())

    if rank == 0:
        out = {
            "metric": "VO frame-pairs/sec at 640x480 (flow+stereo+pose)", "value": value, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f16 activations, f32 weights/correlation/geometry" if args.mixed_precision else DTYPE_TEXT[args.conv_arithmetic]),
            "data": "synthetic",
            "config": {"workload": f"RAFT-Flow + RAFT-Stereo({args.stereo} args) {W}x{H}, {args.iters} iters, "
                                   f"{n_kp} grid keypoints, {'Kabsch' if args.pose == 'svd' else 'RANSAC'} pose; "
                                   f"synthetic stereo frame pairs resident in HBM",
                       "pairs_per_step_per_gpu": B, "pairs_timed": pairs, "iters": args.iters,
                       "sharding": f"{world} rank(s), contiguous pair ranges, no data-path collective" +
                                   (f"; REHEARSAL: {local_world} ranks share {n_dev} GPU(s), gloo rendezvous" if shared_card else ""),
                       "conditions": f"the same {B + 1} frames per rank are re-used by every step and stay resident in "
                                     f"HBM; the PCIe upload is outside value (with_upload reports it inside); no "
                                     f"graph capture; pose gather + composition outside the timed region (gather)"},
            "gather": gather,
            "roofline": roofline,
            "roofline_corr": roofline_corr,
            "hand_kernels_ms_per_step": hand_ms,
            "hand_kernels_share_of_step": hand_ms / (1e3 * elapsed / args.steps),
            "kernels": kernels,
            "skipped_pairs_last_step": int(skipped.sum().item()),
        }
        if world == 1 and not args.no_upload_leg:
            # the same step with the chunk's frames coming from pinned host memory, uploaded on a second
            # stream one chunk ahead of the computation (what a caller holding host frames pays)
            reps = max(2, min(args.steps, 4))
            progress("leg: with_upload")
            lh = left.cpu().repeat(reps, 1, 1)[: reps * B + 1].pin_memory()
            rh = right.cpu().repeat(reps, 1, 1)[: reps * B + 1].pin_memory()
            vo.run(lh[: B + 1], rh[: B + 1], chunk_pairs=B)
            torch.cuda.synchronize()
            t_u = time.perf_counter()
            vo.run(lh, rh, chunk_pairs=B)
            torch.cuda.synchronize()
            dt_u = time.perf_counter() - t_u
            # run() also evaluates the halo frame's stereo pair of the first chunk: one extra frame
            out["with_upload"] = {"value": reps * B / dt_u, "unit": "frame-pairs/s", "pairs": reps * B,
                                  "note": "frames from pinned host memory, chunk k+1 uploaded on a second stream "
                                          "while chunk k computes; includes the first chunk's halo stereo frame"}
            del lh, rh
        if world == 1 and not args.mixed_precision and not args.no_config_legs:
            # the same step, same pipeline object, under the other product arithmetics
            notes = {"f32": "every product on float32 matrix instructions (v_mfma_f32_32x32x2_f32), which run at the vector "
                            "rate, 1/16 of the 16-bit matrix rate",
                     "bf16x3": "three bf16 matrix-core terms per product (16 significant bits per operand), correlation "
                               "volumes included: narrower than the reference's float32, reported beside the metric",
                     "f16x3": "three float16 matrix-core terms per product (22 significant bits per operand)"}
            for kind, key in (("f32", "f32_mfma_only"), ("bf16x3", "bf16x3"), ("f16x3", "f16x3")):
                if kind == args.conv_arithmetic:
                    continue
                progress(f"leg: {key}")
                vo.set_conv_arithmetic(kind)
                try:
                    dt_k = timed_steps(vo, left, right, cached, max(1, min(args.steps, 3)), 1)
                finally:
                    vo.set_conv_arithmetic(args.conv_arithmetic)
                out[key] = {"value": B / dt_k, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt_k, "dtype": DTYPE_TEXT[kind],
                            "note": notes[kind]}
        if world == 1 and not args.mixed_precision and args.mixed_precision_leg and not args.no_mixed_precision_leg:
            progress("leg: mixed_precision")
            out["mixed_precision"] = mixed_precision_leg(vo, make_vo(True), left, right, cached, args.steps)
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_cores or torch.get_num_threads()
            progress(f"leg: cpu_baseline on {threads} thread(s)")
            out["cpu_baseline"] = cpu_baseline(vo, left, right, args.iters, threads, args.stereo)
        if world == 1 and not args.mixed_precision and not args.no_config_legs and args.stereo == "realtime":
            legs_steps = max(1, min(args.steps, 3))
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            # BASELINE config 3: the argument set the reference hard-codes (stereo_disparity_RAFT.py:33-44)
            progress("leg: full_args")
            out["full_args"] = config_leg(lambda: make_vo(False, "full", args.iters, "svd"), left, right, legs_steps,
                                          f"RAFT-Flow + RAFT-Stereo(full args: hidden 128x3, 3 GRU levels) {W}x{H}, "
                                          f"{args.iters} iters, Kabsch pose")
            # what test_camera_tracking.py:66-82 itself runs: full args, iters=16, CameraRansacXform(0.05, 1000)
            progress("leg: reference_driver_settings")
            out["reference_driver_settings"] = config_leg(
                lambda: make_vo(False, "full", 16, "ransac"), left, right, legs_steps,
                f"RAFT-Flow + RAFT-Stereo(full args) {W}x{H}, 16 iters, RANSAC pose (1000 hypotheses per pair)")
            # BASELINE config 4: 1280x720 with the dense grid (max_number=20000 -> 19 928 keypoints), realtime arguments;
            # the flow volume is 14400^2 floats = 829 MB per pair, so a step holds 16 pairs
            Bh = 16
            lh, rh = syn.stereo_sequence(seed=300, frames=Bh + 1, h=720, w=1280)
            lh, rh = torch.from_numpy(lh).cuda(), torch.from_numpy(rh).cuda()
            cfg_h = VOConfig(iters=args.iters, stereo=RAFTStereoConfig.realtime(), max_keypoints=20000,
                             camera_offset=syn.CAMERA_OFFSET)
            torch.cuda.reset_peak_memory_stats()
            progress("leg: high_resolution")
            out["high_resolution"] = config_leg(
                lambda: VisualOdometry(cfg_h, syn.left_K(2.0), syn.focal_length_px() * 2.0, syn.baseline(), np.ones((720, 1280), bool), seed=0),
                lh, rh, legs_steps, f"RAFT-Flow + RAFT-Stereo(realtime args) 1280x720, {args.iters} iters, 19928 grid keypoints, Kabsch pose")
            del lh, rh
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
