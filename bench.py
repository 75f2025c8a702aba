#!/usr/bin/env python3
"""Headline benchmark: NUNet-TLS-LSTM frame step, frames/s (one frame = one 256-bin magnitude
vector of one stream = one 256-sample hop of the 512-pt / 50 % STFT).

    python bench.py                       # 1 GPU, 200 steps, 32 warm-up
    python bench.py --gpus N              # spawns N ranks itself (torch.distributed.run, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # what the driver does

Workload (BASELINE.json configs[1]): NUNet-TLS-LSTM, the reference's trained weights, B = 256
independent streams per GPU, synthetic magnitudes 0.25*|N(0,1)| (default_rng(1234 + first stream of the rank)),
zero-initialised state, inputs resident in HBM before the timed region.  A "step" is one pass
of the hot path over the batch (256 frames per GPU).  N > 1 shards the N*256 streams over ranks (weak
scaling, no data-path collective; RCCL only reduces the timing / frame counters).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_FRAME = 2 * 73_967_252
PEAK_F32_MFMA_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (2.4 GHz)
PEAK_HBM_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_BF16_MFMA_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
# SURVEY.md section 8(d), end-to-end minimum per frame and stream: state read once + written once + frame I/O.
# Baseline variant: SURVEY's figure (2 x 1 647 616 B) charges the WHOLE dilated-dense history in and out every frame -- what the reference's
# functional state protocol moves (converter_nunet_tls.py:1420-1427 shifts every ring by one frame).  A design that keeps the histories as
# in-place rings -- this one, and any sensible one -- must move, per bottleneck and frame, prev_in [F, C], prev_out [F, G] and ONE slot
# [F, k G] of each of the six rings, read once and written once: 2 F (C + G + 21 G) floats = 24 F C with G = C / 2; the 12 stage bottlenecks
# (C = 32, F = 4 4 4 2 1 1 | 1 1 2 4 4 4) and the central one (C = 64, F = 4): 24 x (32 x 32 + 4 x 64) = 30 720 floats.  The conv states
# are the LSTM variant's without its 26 h / c vectors (204 544 floats, read once + written once).  Charging the whole history credited the
# kernel with bytes it never moves (round 5: `over_algorithmic` 0.69).
BASELINE_CONV_STATE_FLOATS = 204_544
BASELINE_RING_FLOATS_PER_FRAME = 24 * (32 * 32 + 4 * 64)
ALG_BYTES_PER_FRAME = {"lstm": 2 * 820_360 + 2 * 1_024,
                       "baseline": 4 * (2 * BASELINE_CONV_STATE_FLOATS + BASELINE_RING_FLOATS_PER_FRAME) + 2 * 1_024}
HOP_SECONDS = 0.016
PKG = os.path.join(ROOT, "nested-u-net-based-real-time-speech-enhancement-mobile-app_amd")


def synthetic_pool(batch, n, seed):
    rng = np.random.default_rng(seed)
    return (0.25 * np.abs(rng.standard_normal((n, batch, 256)))).astype(np.float32)


def kernel_source_sha16(mode, variant="lstm", streams=1):
    """Identity of the kernel a PMC traffic record belongs to: hash of the sources of the step kernel."""
    fused = ["fused_step.hip", "fused_plan.hpp", "fused_plan_lstm.inc"]      # (the one-stream plan: the kernel of the headline configuration)
    if streams > 1:
        fused = ["fused_step.hip", "fused_plan.hpp", "fused_plan_lstm_g%d.inc" % streams]      # (packed plans: LSTM variant only)
    if variant == "baseline":
        fused = ["fused_step.hip", "fused_base.hip", "fused_plan.hpp", "fused_plan_base.inc", "ddb_device.hpp", "ddb_fused.hpp"]
    files = {"fused": fused}.get(mode, [])
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(PKG, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16] if files else None


def pmc_traffic_path(variant="lstm", streams=1):
    if streams > 1:
        return os.path.join(ROOT, "profiles", "pmc_traffic_g%d.json" % streams)
    return os.path.join(ROOT, "profiles", "pmc_traffic.json" if variant == "lstm" else "pmc_traffic_%s.json" % variant)


def scale_traffic(t, B, blob_bytes):
    """HBM bytes per launch at B streams from a PMC record taken at t["batch"] streams: only the per-stream part scales -- the weight
    blob is read once per launch whatever the stream count."""
    if t["batch"] == B:
        return int(t["traffic_bytes"])
    per_stream = (t["traffic_bytes"] - blob_bytes) / t["batch"]
    return int(round(per_stream * B + blob_bytes))


def fused_kernel_name(variant, streams=1):
    if streams > 1:
        return "nutls_fused_step_g%d_kernel" % streams
    return "nutls_fused_base_step_kernel" if variant == "baseline" else "nutls_fused_step_kernel"


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_oracle(batch, threads, budget_s, max_steps=512):
    """frames/s of the oracle at `batch` streams on `threads` host threads, over at most `budget_s` seconds."""
    import torch
    from oracle.nutls_ref import NutlsRef
    torch.set_num_threads(threads)
    ref = NutlsRef(batch=batch)
    x = synthetic_pool(batch, 4, 99)
    ref.step(x[0])                        # warm-up (allocations, thread pool)
    t0, n = time.time(), 0
    while True:
        ref.step(x[n % 4])
        n += 1
        if time.time() - t0 > budget_s or n >= max_steps:
            break
    dt = time.time() - t0
    return batch * n / dt, n, dt


def cpu_baseline(budget_s=12.0):
    """the oracle (torch-CPU restatement of the reference's step) on this box's host cores, B = 1 and
    B = 64, bounded sample.  Each is quoted at the BEST thread count of a small sweep (thread sweep over 1 .. 256 threads on the
    2 x 64-core EPYC host of the GPU boxes: profiles/r04_cpu_threads.txt, written by tools/cpu_thread_sweep.py -- B = 1, the reference's
    own operating point (interpreter_proposed.py:215), is fastest on ONE thread, B = 64 on 8 .. 16; more threads are slower for these
    small GEMMs): `value` = B = 64 at the best of {8, 16} threads (`cores`), `value_b1` = B = 1 at the best of {1, 8} (`cores_b1`); the
    figure SURVEY 8(d) prescribes, torch.set_num_threads(os.cpu_count()), is reported beside them (`value_all_cores`, a short sample:
    it is the slow one)."""
    import torch
    ncpu = os.cpu_count() or 1
    all_cores = ncpu
    sweep_b1 = {t: _time_oracle(1, t, budget_s * 0.15) for t in sorted({1, min(8, ncpu)})}
    sweep_b64 = {t: _time_oracle(64, t, budget_s * 0.25) for t in sorted({min(8, ncpu), min(16, ncpu)})}
    threads_b1 = max(sweep_b1, key=lambda t: sweep_b1[t][0])
    threads = max(sweep_b64, key=lambda t: sweep_b64[t][0])
    b1, b64 = sweep_b1[threads_b1], sweep_b64[threads]
    full, full_note = b64, None
    if all_cores != threads:
        # in its own process with a deadline: on the 256-thread hosts of the GPU boxes one 64-stream step at all cores takes minutes
        # (profiles/r04_cpu_threads.txt: 45.7 frames/s at 128 threads, no row at 256 within 200 s)
        code = ("import sys; sys.path.insert(0, %r); import bench; r = bench._time_oracle(64, %d, 2.0, max_steps=8); print(r[0], r[1], r[2])"
                % (ROOT, all_cores))
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=30)
            v = r.stdout.split()
            full = (float(v[0]), int(v[1]), float(v[2]))
        except (subprocess.TimeoutExpired, ValueError, IndexError):
            full, full_note = None, "no 64-stream step finished within 30 s on %d threads" % all_cores
    torch.set_num_threads(threads)
    rec = {"value": round(b64[0], 1), "unit": "frames/s", "cores": threads, "kind": "port",
           "value_b1": round(b1[0], 1), "cores_b1": threads_b1,
           "thread_sweep": {"b64": {str(t): round(v[0], 1) for t, v in sweep_b64.items()}, "b1": {str(t): round(v[0], 1) for t, v in sweep_b1.items()}},
           "value_all_cores": round(full[0], 1) if full else None, "all_cores": all_cores,
           "host_cores": os.cpu_count(), "cpu": cpu_model_name(),
           "sample": "oracle/nutls_ref.py (torch-CPU fp32): %d steps of 64 streams in %.1f s (value, %d threads: the best of %s), %d steps of 1 stream in %.1f s "
                     "(value_b1, %d thread(s): the best of %s)" % (b64[1], b64[2], threads, sorted(sweep_b64), b1[1], b1[2], threads_b1, sorted(sweep_b1))}
    if full:
        rec["sample"] += ", %d steps of 64 streams in %.1f s on %d threads (value_all_cores)" % (full[1], full[2], all_cores)
    if full_note:
        rec["all_cores_note"] = full_note
    return rec


def bench_offline(args, rank, world, local_rank):
    """one utterance per GPU, blocks of T frames resident in HBM; value = frames/s of that utterance."""
    import torch
    import nunet_amd
    T_ = args.offline
    U = max(1, args.offline_utterances)      # independent utterances
    batched = U > 1 and not args.offline_handles
    if batched:
        # ONE handle with a batch dimension (nutls_create_offline_batch): every layer one launch over the frames of all utterances
        offs = [nunet_amd.NutlsOffline(max_frames=T_, device=local_rank, utterances=U, pipeline=args.offline_chunks)]
        pool = torch.from_numpy(np.stack([synthetic_pool(T_, 4, 1234 + rank + 7 * u) for u in range(U)], axis=1)).cuda()      # [4, U, T, 256]
        outs = [torch.empty(U, T_, 256, device="cuda")]
        streams = [torch.cuda.current_stream()]
    else:
        # (--offline-handles: one handle and one torch stream per utterance -- round 3's way of running several utterances)
        offs = [nunet_amd.NutlsOffline(max_frames=T_, device=local_rank, pipeline=args.offline_chunks) for _ in range(U)]
        pool = torch.from_numpy(synthetic_pool(T_, 4, 1234 + rank)).cuda()
        outs = [torch.empty(T_, 256, device="cuda") for _ in range(U)]
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(U - 1)]
    out = outs[0]

    def blocks(n):
        for s in range(n):
            for u in range(len(offs)):
                with torch.cuda.stream(streams[u]):
                    offs[u].process_block_device(pool[(s + u) % 4], outs[u])

    torch.cuda.synchronize()
    blocks(max(2, args.warmup // 8))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    blocks(args.steps)
    t_enq = time.perf_counter() - t0                  # the host's share: all launches of all blocks enqueued
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    if rank == 0:
        # roofline of the block as a whole: the mode is many kernels (one per layer and chunk + 13 scans); its arithmetic is the step's
        # (147.9 MFLOP per frame), executed as three bf16 MFMAs per fp32 product -- priced against the fp32-MFMA peak (the
        # arithmetic the results are equivalent to) with the executed-bf16 view beside it.  The layer-fused bytes per frame (SURVEY 8(d):
        # 4 151 808 B) put the HBM bound far below: the mode is bound by its serial scans and its launch chain.
        fps = U * T_ * args.steps / dt
        tfl = fps * FLOPS_PER_FRAME / 1e12
        offline_roofline = {"kernel": "block mode (conv_bf16x3_kernel per layer and chunk + lstm_scan_kernel x 13 per chunk)", "bound": "mfma",
                            "achieved": round(tfl, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / PEAK_F32_MFMA_TFLOPS, 4),
                            "traffic": None,
                            "executed_bf16": {"tflops": round(3 * tfl, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "frac": round(3 * tfl / PEAK_BF16_MFMA_TFLOPS, 4)},
                            "hbm_layer_fused_gbps": round(fps * 4_151_808 / 1e9, 1),
                            "scan_floor": "13 scans x T x 0.153 us per step = %.2f ms per block of %d frames (serial per stage; chunks overlap them)" % (13 * T_ * 0.153e-3, T_)}
        cpu = None
        if not args.no_cpu_baseline:
            b1 = _time_oracle(1, min(8, os.cpu_count() or 1), 6.0)      # one utterance on the CPU = the oracle frame by frame
            cpu = {"value": round(b1[0], 1), "unit": "frames/s", "cores": min(8, os.cpu_count() or 1), "kind": "port",
                   "sample": "oracle/nutls_ref.py, one stream frame by frame: %d frames in %.1f s" % (b1[1], b1[2])}
        print(json.dumps({"metric": "STFT frames/sec (512-pt, 50% hop) through the NUNet-TLS frame step", "value": round(U * T_ * args.steps / dt, 1),
                          "roofline": offline_roofline, "cpu_baseline": cpu,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(2, args.warmup // 8),
                          "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic magnitudes 0.25*|N(0,1)|, trained weights",
                          "config": {"workload": "offline / block mode: %s, %d consecutive frames per call (SURVEY 8f.2)" % (
                                         "ONE utterance" if U == 1 else ("%d utterances in one handle ([U, T, 256] per call)" % U if batched else "%d utterances, one handle and stream each" % U), T_),
                                     "frames_per_block": T_, "utterances": U, "batched_handle": batched,
                                     "pipeline_chunks": args.offline_chunks or "auto (2 from 256 frames, 3 from 768)",
                                     "mode": "per-layer kernels, frame index as stream index, single-wavefront LSTM scan, block pipeline"},
                          "rtf_per_stream": round(dt / args.steps / T_ / 0.016, 6),
                          "host_enqueue_ms_per_block": round(1e3 * t_enq / args.steps, 4)}))
    for off in offs:
        off.close()


def other_config_records(local_rank):
    """The other BASELINE.json configurations, each timed over its own >= 100 ms window on this GPU, so that the driver's default
    run records them too (the headline `value` stays configs[1]): configs[2] (baseline variant, B = 256), the per-GPU size of
    configs[3] on one GPU (B = 2048, device-resident) and configs[4] (B = 1024 streams, host buffers in and out every call).
    `roofline` as in the main record: algorithmic bytes of one launch / launch time / 8 TB/s; for the host-I/O
    configuration the launch time includes the PCIe transfers (it is the step latency the caller sees)."""
    import torch
    import nunet_amd
    from nunet_amd.weights import synthetic_weights, write_blob
    recs = []
    for (tag, variant, B, host_io) in (("configs[2]: dilated-dense baseline, batch=256, device-resident", "baseline", 256, False),
                                       ("configs[3] size on one GPU: NUNet-TLS-LSTM, batch=2048, device-resident", "lstm", 2048, False),
                                       ("configs[4]: NUNet-TLS-LSTM streaming, batch=1024, host buffers in/out per call", "lstm", 1024, True)):
        weights = write_blob(synthetic_weights("baseline", seed=4321), int8_convs=True) if variant == "baseline" else None
        eng = nunet_amd.NutlsEngine(weights, batch=B, device=local_rank, mode="fused", variant=variant)
        pool_host = synthetic_pool(B, 4, 1234)
        pool = torch.from_numpy(pool_host).cuda()
        out = torch.empty(B, 256, device="cuda")
        def timed(step):
            for i in range(8):
                step(i)
            torch.cuda.synchronize()
            k = 8
            while True:
                t0 = time.perf_counter()
                for i in range(k):
                    step(i)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if dt >= 0.1 or k >= 100000:
                    return k, dt
                k = max(k + 1, int(k * 0.11 / max(dt, 1e-6)) + 1)

        pageable = None
        if host_io:
            # the caller's frames and results in page-locked host memory (nutls_host_alloc: the fused kernel reads / writes it over the link,
            # no copy commands) -- and, beside it, in pageable numpy arrays (the runtime stages both copies), which is what rounds 1-5 timed
            kp, dtp = timed(lambda i: eng.step(pool_host[i % 4]))
            pageable = {"value": round(B * kp / dtp, 1), "ms_per_step": round(1e3 * dtp / kp, 4)}
            pin_in = [nunet_amd.host_alloc((B, 256)) for _ in range(4)]
            for dst, src in zip(pin_in, pool_host):
                dst[...] = src
            pin_out = nunet_amd.host_alloc((B, 256))
            step = lambda i: eng.step(pin_in[i % 4], out=pin_out)
        else:
            step = lambda i: eng.step(pool[i % 4], out)
        k, dt = timed(step)
        ms = 1e3 * dt / k
        alg = ALG_BYTES_PER_FRAME[variant] * B + eng.weight_blob_bytes()
        gbps = alg / (ms * 1e-3) / 1e9
        rec = {"workload": tag, "variant": variant, "streams": B, "host_io": host_io, "value": round(B * k / dt, 1), "unit": "frames/s",
               "ms_per_step": round(ms, 4), "steps": k, "window_ms": round(1e3 * dt, 2), "streams_per_workgroup": getattr(eng, "streams_per_workgroup", 1),
               "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBS, 4),
                            "algorithmic_bytes_per_launch": alg}}
        # measured HBM traffic of the launch (PMC record of THIS kernel: profiles/pmc_traffic_baseline.json / pmc_traffic_g<streams>.json)
        spw = rec["streams_per_workgroup"]
        tpath = pmc_traffic_path(variant, spw)
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            if t.get("variant", "lstm") == variant and t.get("kernel_source_sha16") == kernel_source_sha16("fused", variant, spw) and t.get("batch"):
                traffic = scale_traffic(t, B, eng.weight_blob_bytes())
                rec["roofline"]["traffic"] = traffic
                tg = traffic / (ms * 1e-3) / 1e9
                rec["roofline"]["hbm_measured"] = {"achieved": round(tg, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(tg / PEAK_HBM_GBS, 4),
                                                   "over_algorithmic": round(traffic / alg, 3)}
                if t["batch"] != B:
                    rec["roofline"]["hbm_measured"]["note"] = "per-stream part scaled by the stream count from the PMC record at B = %d (the weight blob counted once)" % t["batch"]
                if host_io:
                    rec["roofline"]["hbm_measured"]["note"] = (rec["roofline"]["hbm_measured"].get("note", "") + "; the launch time of this configuration includes the PCIe transfers of the frame and the result").lstrip("; ")
        if host_io:
            rec["step_latency_ms"] = rec["ms_per_step"]
            rec["real_time_budget_ms"] = 16.0
            rec["host_memory"] = "page-locked (nutls_host_alloc): the kernel reads the frame from / writes the result to the host buffers over PCIe"
            rec["pageable_host_buffers"] = pageable
            del pin_in, pin_out
        recs.append(rec)
        eng.close()
        del pool, out
    return recs


def parity_check(eng_cls, pool, n_streams=4, steps=6):
    """GPU vs oracle on identical inputs for the first streams of the workload (streams are
    independent, so a 4-stream engine reproduces streams 0..3 of the 256-stream run)."""
    from oracle.nutls_ref import NutlsRef
    eng, ref = eng_cls(batch=n_streams), NutlsRef(batch=n_streams)
    se = 0.0
    for s in range(steps):
        x = np.ascontiguousarray(pool[s % len(pool), :n_streams])
        d = eng.step(x) - ref.step(x).numpy()
        se += float(np.mean(d * d))
    eng.close()
    return float(np.sqrt(se / steps))


class _LauncherSelfTestEngine:
    """CPU stand-in used ONLY by `--selftest-launcher` (tests/test_sharding.py): exercises this file's rank plumbing
    (self-launch, stream sharding, barrier, counter reduction, the JSON line) without a GPU.  It computes nothing of the
    model -- it is not a fallback of the product path."""

    def __init__(self, batch):
        self.batch = batch

    def step(self, x):
        return x * 0.5

    def close(self):
        pass


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-run this command under torch.distributed.run, one rank per GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--condition-ms", type=float, default=200.0, help="untimed steps before the warm-up steps until the GPU runs at its steady clocks (0: none)")
    ap.add_argument("--batch", type=int, default=256, help="streams per GPU")
    ap.add_argument("--mode", default=None, choices=["fused", "graph", "launches"],
                    help="default: fused")
    ap.add_argument("--variant", default="lstm", choices=["lstm", "baseline"],
                    help="baseline = dilated-dense bottleneck with synthetic weights, seed 4321 (BASELINE configs[2])")
    ap.add_argument("--host-io", action="store_true",
                    help="streaming serving (BASELINE configs[4]): one step per host call, host buffers in/out (H2D + D2H timed)")
    ap.add_argument("--host-memory", choices=("pinned", "pageable"), default="pinned",
                    help="--host-io: the caller's buffers in page-locked memory from nutls_host_alloc (default: the fused kernel reads / writes them "
                         "over PCIe, no copy commands) or in pageable numpy arrays (the runtime stages both copies)")
    ap.add_argument("--frontend", action="store_true",
                    help="a step = STFT analysis + model step + inverse STFT/overlap-add of one 256-sample hop per stream, all on the GPU")
    ap.add_argument("--offline", type=int, default=0, metavar="T",
                    help="offline / block mode: ONE utterance, a step = one block of T consecutive frames (frames/s of that utterance)")
    ap.add_argument("--offline-chunks", type=int, default=0, help="offline mode: chunks of the block pipeline (0 = library default)")
    ap.add_argument("--offline-utterances", type=int, default=1, help="offline mode: independent utterances per call, the batch dimension of ONE handle (value = their total frames/s)")
    ap.add_argument("--offline-handles", action="store_true", help="offline mode with several utterances: one handle and one stream per utterance instead of one batched handle")
    ap.add_argument("--ctfa-mode", default="frame", choices=["frame", "causal32"],
                    help="causal32: the offline model's 32-frame causal CTFA in the streaming kernel (per-stream attention history kept by the library: "
                         "BASELINE configs[4]'s 'persistent CTFA state'); default: what the reference's streaming graph computes (TA / 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` records of the default line (configs[2], [3]-size, [4])")
    ap.add_argument("--profile-json", default="", help="write the per-op timeline here")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    selftest = args.selftest_launcher
    if not selftest:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d has no GPU: %d visible" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    from nunet_amd.sharding import collective_proof, reduce_throughput, stream_range

    if args.offline:
        return bench_offline(args, rank, world, local_rank)
    # the job is world * batch independent streams; this rank owns a contiguous slice of them
    lo, hi = stream_range(world * args.batch, rank, world)
    B = hi - lo
    device = torch.device("cpu") if selftest else torch.device("cuda", local_rank)
    mode = args.mode or "fused"
    if selftest:
        eng = _LauncherSelfTestEngine(B)
    else:
        import nunet_amd
        weights = None
        if args.variant == "baseline":
            from nunet_amd.weights import synthetic_weights, write_blob
            # random-init weights (none are trained), stored as the reference's export stores them: conv kernels int8
            weights = write_blob(synthetic_weights("baseline", seed=4321), int8_convs=True)
        eng = nunet_amd.NutlsEngine(weights, batch=B, device=local_rank, mode=mode, variant=args.variant)
        if args.ctfa_mode != "frame":
            eng.set_ctfa_mode(args.ctfa_mode)
    pool_host = synthetic_pool(B, 8, 1234 + lo)
    pool = torch.from_numpy(pool_host).to(device)            # inputs resident in HBM
    out = torch.empty(B, 256, device=device)

    def sync():
        if not selftest:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    pin_in = pin_out = None
    if args.host_io and not selftest and args.host_memory == "pinned":
        import nunet_amd
        pin_in = [nunet_amd.host_alloc((B, 256)) for _ in range(8)]
        for dst, src in zip(pin_in, pool_host):
            dst[...] = src
        pin_out = nunet_amd.host_alloc((B, 256))
    pcm = pcm_out = None
    if args.frontend:        # PCM hops resident in HBM: white noise at speech level
        pcm = (0.05 * torch.randn(8, B, 256, generator=torch.Generator().manual_seed(1234 + lo))).cuda()
        pcm_out = torch.empty(B, 256, device="cuda")

    def one_step(s):
        if selftest:
            out.copy_(eng.step(pool[s % 8]))
            return out
        if args.frontend:
            return eng.enhance_hop(pcm[s % 8], "edge", pcm_out)
        if args.host_io:
            if pin_in is not None:
                return eng.step(pin_in[s % 8], out=pin_out)      # page-locked host buffers: the kernel works on them over the link, synchronous
            return eng.step(pool_host[s % 8])            # numpy in / numpy out: H2D + step + D2H, synchronous
        return eng.step(pool[s % 8], out)

    # Clock conditioning (untimed, reported as `conditioning_ms`): a GPU that was idle runs its first ~50 ms of launches 1.5-6 % below its
    # steady clocks (tools/gpu_cold_start.py), and the driver's window -- 5 warm-up + 20 timed steps -- is 10 ms.  The
    # metric is a steady-state rate, so the device is brought to steady clocks with the same step before the W warm-up steps; the timed
    # region below is still exactly --steps steps behind exactly --warmup warm-up steps.
    # ... and the same W + K steps are timed once BEFORE the conditioning (`value_unconditioned`: what the driver's arguments measure on a GPU
    # that was idle), so that every line shows what the conditioning is worth on its box
    uncond = None
    if args.condition_ms > 0 and not selftest and world == 1:
        for s in range(args.warmup):
            one_step(s)
        sync()
        tu = time.perf_counter()
        for s in range(args.steps):
            one_step(s)
        sync()
        uncond = time.perf_counter() - tu
    cond_steps = 0
    if args.condition_ms > 0 and not selftest:
        tc = time.perf_counter()
        while (time.perf_counter() - tc) * 1e3 < args.condition_ms:
            for s in range(16):
                one_step(s)
            cond_steps += 16
            sync()
    for s in range(args.warmup):
        one_step(s)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(s)
    sync()
    elapsed = time.perf_counter() - t0
    barrier()
    frames = B * args.steps
    total_frames, max_elapsed = reduce_throughput(frames, elapsed, dist if world > 1 else None, device)
    proof = collective_proof(frames / elapsed, dist if world > 1 else None, device)
    if proof["ranks_reduced"] != args.gpus:
        raise SystemExit("the counter all-reduce saw %d ranks, --gpus is %d" % (proof["ranks_reduced"], args.gpus))
    if not selftest:      # which device each rank really ran on (stderr: the JSON line stays alone on stdout)
        pr = torch.cuda.get_device_properties(local_rank)
        sys.stderr.write("[bench rank %d/%d] cuda:%d %s pci %s, %d streams [%d, %d), %.0f frames/s\n" % (
            rank, world, local_rank, pr.name, getattr(pr, "pci_bus_id", "?"), B, lo, hi, frames / elapsed))
    assert args.host_io or bool(torch.isfinite(pcm_out if args.frontend else out).all())

    if rank == 0:
        value = total_frames / max_elapsed
        line = {
            "metric": "STFT frames/sec (512-pt, 50% hop) through the NUNet-TLS frame step",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * max_elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic magnitudes 0.25*|N(0,1)|, " + ("trained weights of the reference's nutls_lstm.tflite (int8 conv kernels kept int8 on the device, scale applied in fp32)"
                                                                 if args.variant == "lstm" else "synthetic weights (no trained baseline weights exist)"),
            "config": {"workload": ("NUNet-TLS-LSTM (proposed) frame step, batch=%d streams per GPU, 256-bin frames (BASELINE configs[1])" % args.batch)
                       if args.variant == "lstm" else
                       ("NUNet-TLS dilated-dense baseline frame step, synthetic weights seed 4321, batch=%d streams per GPU (BASELINE configs[2])" % args.batch),
                       "variant": args.variant, "host_io": bool(args.host_io), "stft_istft_on_gpu": bool(args.frontend),
                       "streams_per_gpu": args.batch, "total_streams": total_frames // args.steps, "parallelism": "stream-sharded x%d" % world,
                       "mode": mode, "streams_per_workgroup": getattr(eng, "streams_per_workgroup", 1), "ctfa_mode": args.ctfa_mode},
            "rtf_per_stream": round(1e3 * max_elapsed / args.steps / 16.0, 5),
            # whole-job rate x SURVEY 8(d)'s 147.93 MFLOP per frame (the graph as lowered: the CTFA frequency branch as a conv over
            # all F bins); `roofline.achieved` uses the plan's own count (144.0 M: those 1x1s at their true size) -- both stated
            "tflops": round(value * FLOPS_PER_FRAME / 1e12, 2),
            "flops_per_frame": {"survey_8d": FLOPS_PER_FRAME, "note": "tflops / frac_f32_peak use survey_8d; roofline.achieved uses roofline.flops_per_launch / streams (CTFA 1x1 convs at their true size)"},
            "frac_f32_peak": round(value * FLOPS_PER_FRAME / 1e12 / PEAK_F32_MFMA_TFLOPS / world, 4),
            # the timed window: a window shorter than 100 ms (the driver's --steps 20 is 10 ms) still carries launch ramp-up;
            # `roofline` below re-times the kernel over its own >= 100 ms window
            "timed_window_ms": round(1e3 * max_elapsed, 3), "short_window": bool(max_elapsed < 0.1),
            "conditioning_ms": args.condition_ms if cond_steps else 0, "conditioning_steps": cond_steps,
            # the same --warmup + --steps window on the GPU as this process found it, before any conditioning
            "value_unconditioned": round(B * args.steps / uncond, 1) if uncond else None,
            "ms_per_step_unconditioned": round(1e3 * uncond / args.steps, 4) if uncond else None,
            "collective": {"backend": proof["backend"], "ranks_reduced": proof["ranks_reduced"],
                           "per_rank_frames_per_s": [round(x, 1) for x in proof["per_rank"]]},
        }
        if not selftest:
            line.update(kernel_report(args, eng, pool, out, B, mode))
        if args.host_io:
            line["step_latency_ms"] = line["ms_per_step"]
            line["real_time_budget_ms"] = 16.0
            line["host_memory"] = "page-locked (nutls_host_alloc)" if pin_in is not None else "pageable"
        if not selftest and not args.no_cpu_baseline and args.variant == "lstm" and args.ctfa_mode == "frame":
            import nunet_amd
            line["cpu_baseline"] = cpu_baseline()
            line["parity_rms_vs_oracle"] = parity_check(nunet_amd.NutlsEngine, pool_host)
        default_run = (not selftest and world == 1 and mode == "fused" and args.variant == "lstm" and args.batch == 256
                       and not (args.host_io or args.frontend) and args.ctfa_mode == "frame")
        if default_run and not args.no_other_configs:
            eng.close()
            line["other_configs"] = other_config_records(local_rank)
        print(json.dumps(line))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def kernel_report(args, eng, pool, out, B, mode):
    """roofline of the dominant kernel + the encoder conv stack (north-star target) from the in-kernel timeline."""
    import torch
    stream = torch.cuda.current_stream()
    plan = eng.launch_plan()                       # per layer: algorithmic flops / bytes for the handle's batch
    step_flops = sum(p["flops"] for p in plan)
    by_layer = {p["layer"].split("#")[0]: {"flops": 0.0, "bytes": 0.0} for p in plan}
    for p in plan:
        by_layer[p["layer"].split("#")[0]]["flops"] += p["flops"]
        by_layer[p["layer"].split("#")[0]]["bytes"] += p["bytes"]
    rep = {}
    one_launch = mode == "fused"
    if one_launch:
        # The whole step is ONE kernel.  Its average duration over the timed region: HIP events on the launch stream.
        # The event window covers at least --steps launches AND at least 100 ms, so that a short driver run (--steps 20 = 12 ms)
        # does not quote the kernel on its ramp-up.
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        k = args.steps
        while True:
            ev[0].record(stream)
            for s in range(k):
                eng.step(pool[s % 8], out)
            ev[1].record(stream)
            torch.cuda.synchronize()
            win_ms = ev[0].elapsed_time(ev[1])
            if win_ms >= 100.0 or k >= 100000:
                break
            k = max(k + 1, int(k * 110.0 / max(win_ms, 1e-3)) + 1)
        n_launch, avg_ms = k, win_ms / k
        achieved = step_flops / (avg_ms * 1e-3) / 1e12
        traffic = None
        traffic_note = None
        spw = getattr(eng, "streams_per_workgroup", 1) if mode == "fused" else 1
        tpath = pmc_traffic_path(args.variant, spw)
        if os.path.exists(tpath):     # HBM bytes per launch from rocprofv3 PMC passes -- only if they were taken on THIS kernel
            t = json.load(open(tpath))
            if (t.get("mode") == mode and t.get("variant", "lstm") == args.variant
                    and t.get("kernel_source_sha16") == kernel_source_sha16(mode, args.variant, spw)):
                if t.get("batch") == B:
                    traffic = t["traffic_bytes"]
                elif spw > 1 and t.get("batch"):
                    # packed plans: one record (B = 1024); the traffic of a launch is per-stream state traffic + the L2-resident weight blob,
                    # so another multiple of the plan's round of workgroups scales with the stream count
                    traffic = scale_traffic(t, B, eng.weight_blob_bytes())
                    traffic_note = "per-stream part scaled by the stream count from the PMC record at B = %d (the weight blob counted once)" % t["batch"]
        # Lower bounds of one launch: HBM = SURVEY 8(d)'s end-to-end minimum (every state tensor read once and
        # written once, + the frame I/O) x streams + the weight blob once, at 8 TB/s; MFMA = the conv FLOPs on the pipe the
        # kernel actually uses.  The fused kernel computes every fp32 product as THREE bf16 MFMAs (error-free split of the
        # activation, int8 weights exact in bf16, fp32 accumulate), so its matrix-pipe bound is 3 x flops at the dense bf16
        # peak -- below the HBM bound: the roofline that bounds the step is HBM, and that is the primary record.  The
        # fp32-MFMA view (the arithmetic the results are equivalent to, last round's primary) stays beside it.
        kname = fused_kernel_name(args.variant, spw)
        bf16x3 = mode == "fused"
        alg_bytes = ALG_BYTES_PER_FRAME[args.variant] * B + eng.weight_blob_bytes()
        gbps = alg_bytes / (avg_ms * 1e-3) / 1e9
        t_hbm = alg_bytes / (PEAK_HBM_GBS * 1e9) * 1e3
        t_mfma = (3 * step_flops / (PEAK_BF16_MFMA_TFLOPS * 1e12) if bf16x3 else step_flops / (PEAK_F32_MFMA_TFLOPS * 1e12)) * 1e3
        mfma_rec = {"achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "what": "algorithmic fp32 FLOPs of the launch against the dense fp32-MFMA peak (the arithmetic the results are equivalent to)"}
        if bf16x3:
            mfma_rec["pipe"] = {"executed": "3 bf16 MFMAs per fp32 product (x = hi + mid + lo error-free, int8 weights exact in bf16, fp32 accumulate)",
                                "executed_tflops": round(3 * achieved, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                "frac": round(3 * achieved / PEAK_BF16_MFMA_TFLOPS, 4)}
        hbm_bound = t_hbm >= t_mfma
        rep["roofline"] = {"kernel": kname, "bound": "hbm" if hbm_bound else "mfma",
                           "achieved": round(gbps, 1) if hbm_bound else round(achieved, 2),
                           "peak": PEAK_HBM_GBS if hbm_bound else PEAK_F32_MFMA_TFLOPS, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                           "frac": round(gbps / PEAK_HBM_GBS, 4) if hbm_bound else round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                           "traffic": traffic, "launches_per_step": 1, "avg_launch_ms": round(avg_ms, 5),
                           "algorithmic_bytes_per_launch": alg_bytes, "flops_per_launch": step_flops,
                           "bounds_ms": {"hbm": round(t_hbm, 4), "mfma": round(t_mfma, 4),
                                         "note": ("time of one launch at each peak; the larger one is `bound`.  The step runs one stream per CU and is "
                                                  "limited by the dependent chain of its 154 ops (DESIGN.md section 4), not by either peak") if spw == 1 else
                                                 ("time of one launch at each peak; the larger one is `bound`.  Packed plan: a workgroup steps %d streams -- side by side "
                                                  "on one position axis where the layer's images fit LDS that often, one after the other in the outer layers -- one "
                                                  "workgroup per CU at a time; limited by the dependent chain of its op instances (DESIGN.md section 4 'Packed plans'), "
                                                  "not by either peak" % spw)},
                           "mfma": mfma_rec,
                           "event_window": {"launches": n_launch, "ms": round(win_ms, 3)}}
        if traffic:      # measured HBM traffic of the same launch (PMC), against the same peak
            tg = traffic / (avg_ms * 1e-3) / 1e9
            rep["roofline"]["hbm_measured"] = {"achieved": round(tg, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(tg / PEAK_HBM_GBS, 4),
                                               "over_algorithmic": round(traffic / alg_bytes, 3)}
            if traffic_note:
                rep["roofline"]["hbm_measured"]["note"] = traffic_note
        rep["roofline"]["streams_per_workgroup"] = spw
        if spw > 1:
            # packed plan: no profiling twin of the kernel in the default build (no in-kernel timeline, no encoder-stack record); its
            # traffic record is profiles/pmc_traffic_g<streams>.json
            return rep
        # in-kernel timeline of workgroup 0 (wall clock stamps at every op boundary)
        prof = eng.profile_fused
        names = [p["layer"] for p in eng.fused_plan()]
        for _ in range(2):
            prof()
        us = np.zeros(len(names))
        reps = 10
        for _ in range(reps):
            us += prof()
        ms = us / reps / 1e3
    else:
        eng.set_mode("launches")
        names = [p["layer"] for p in plan]
        ms = np.zeros(len(plan))
        for _ in range(2):
            eng.profile_step()
        for _ in range(10):
            ms += eng.profile_step()
        ms /= 10
        fam = {}
        for p, t in zip(plan, ms):
            f = fam.setdefault(p["family"], {"ms": 0.0, "n": 0, "flops": 0.0})
            f["ms"] += t; f["n"] += 1; f["flops"] += p["flops"]
        dom = max(fam, key=lambda k: fam[k]["ms"])
        d = fam[dom]
        avg_ms = d["ms"] / d["n"]
        achieved = d["flops"] / d["n"] / (avg_ms * 1e-3) / 1e12
        rep["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                           "launches_per_step": d["n"], "avg_launch_ms": round(avg_ms, 5), "share_of_step": round(d["ms"] / ms.sum(), 3)}
    # The 26 encoder (2,3) stride-2 convs = the north-star's "encoder conv stack".  In the one-launch modes the stamps are
    # workgroup 0's, i.e. the time ONE resident stream per CU needs; min(B, #CUs) streams run concurrently.
    resident = min(B, 256) if one_launch else B
    enc_ms = sum(t for n, t in zip(names, ms) if re.search(r"_en\d?_conv\d$", n.split("#")[0]))
    enc = [v for k, v in by_layer.items() if re.search(r"_en\d?_conv\d$", k)]
    enc_bytes = sum(v["bytes"] for v in enc) / B * resident
    enc_flops = sum(v["flops"] for v in enc) / B * resident
    enc_gbs = enc_bytes / (enc_ms * 1e-3) / 1e9
    rep["encoder_conv_stack"] = {"layers": len(enc), "ms_per_step": round(float(enc_ms), 4), "achieved": round(enc_gbs, 1),
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(enc_gbs / PEAK_HBM_GBS, 4),
                                 "tflops": round(enc_flops / (enc_ms * 1e-3) / 1e12, 2), "concurrent_streams": resident,
                                 # what `achieved` is: SURVEY 8(d)'s layer-fused ALGORITHMIC bytes of the 26 layers (fp32 activations in + out
                                 # per layer, 441 344 B per frame and stream) over the time the stamped profiling build spends in them --
                                 # an equivalent rate, not measured HBM traffic (the fused kernel keeps the current frame in LDS and the
                                 # weights int8; the whole step's measured traffic is roofline.traffic)
                                 "kind": "layer-fused algorithmic bytes / in-kernel timeline of the profiling build (equivalent rate, not PMC traffic)"}
    if one_launch and mode == "fused":
        # the stated lower bound of this stack in the one-stream-per-CU design: per layer, its MFMA issue cycles on the 4 SIMDs
        # of ONE CU at the clock the chip sustains in this kernel (1.9 GHz under MFMA load) plus one
        # barrier-to-barrier LDS round trip (partials out, row-wise epilogue back in: 2 x ~0.1 us) -- the streams are
        # independent, so a layer's positions cannot be spread over more than the stream's own CU
        per_stream = [v["flops"] / B for k, v in by_layer.items() if re.search(r"_en\d?_conv\d$", k)]
        # (the convs run on the bf16 pipe, three MFMAs per fp32 product: 512 bf16 MAC/clk/SIMD at the dense peak / 3)
        mac_clk = PEAK_BF16_MFMA_TFLOPS * 1e12 / 2.0 / (256 * 4 * 2.4e9) / 3.0
        floor_ms = sum(f / 2.0 / (mac_clk * 4.0) / 1.9e9 + 0.2e-6 for f in per_stream) * 1e3
        rep["encoder_conv_stack"]["floor_ms"] = round(floor_ms, 4)
        rep["encoder_conv_stack"]["over_floor"] = round(float(enc_ms) / floor_ms, 2)
        rep["encoder_conv_stack"]["floor"] = ("sum over layers of MACs / (%.1f fp32-equivalent MAC/clk/SIMD [bf16 MFMA peak / 3] x 4 SIMDs x 1.9 GHz) + 0.2 us "
                                              "(one exchange round trip between two barriers)" % mac_clk)
        # per-layer view: level-1 convs (64 -> 32 at the stage's full resolution) vs the deeper 32 -> 32 ones, whose cost is the op's
        # latency chain whatever they compute
        lv1 = [t for n, t in zip(names, ms) if re.search(r"_en\d?_conv1$", n.split("#")[0])]
        deep = [t for n, t in zip(names, ms) if re.search(r"_en\d?_conv[2-9]$", n.split("#")[0])]
        rep["encoder_conv_stack"]["level1_layers"] = {"n": len(lv1), "us": round(1e3 * float(sum(lv1)), 2)}
        rep["encoder_conv_stack"]["deep_layers"] = {"n": len(deep), "us": round(1e3 * float(sum(deep)), 2),
                                                   "us_each": round(1e3 * float(sum(deep)) / max(1, len(deep)), 3)}
    if one_launch and mode == "fused" and args.variant == "lstm" and getattr(eng, "streams_per_workgroup", 1) == 1 and args.ctfa_mode == "frame":
        # the same stack on the UN-instrumented instruction stream: launches of the library's stop twin (production code + one scalar compare
        # per op) that end in front of op N, differenced (nutls_profile_production).  The profiling build above stamps the critical wave of
        # every op eight times and its workgroup 0 runs in the wake of the others; this is what the ops cost the kernel `value` is measured on.
        # (timing only: the handle's streams are reset afterwards -- nothing after this point uses their state)
        pu = eng.profile_production()
        pnames = names      # (eng.fused_plan(): the op order nutls_profile_production reports in)
        penc = [u for n, u in zip(pnames, pu[1:]) if re.search(r"_en\d?_conv\d$", n.split("#")[0])]
        pdeep = [u for n, u in zip(pnames, pu[1:]) if re.search(r"_en\d?_conv[2-9]$", n.split("#")[0])]
        p_ms = 1e-3 * float(sum(penc))
        p_gbs = enc_bytes / (p_ms * 1e-3) / 1e9
        rep["encoder_conv_stack"]["production"] = {
            "ms_per_step": round(p_ms, 4), "achieved": round(p_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(p_gbs / PEAK_HBM_GBS, 4),
            "deep_layers_us_each": round(float(sum(pdeep)) / max(1, len(pdeep)), 3), "whole_step_us": round(float(pu[1:].sum()), 1),
            "launch_floor_us": round(float(pu[0]), 2),
            "kind": "the same algorithmic bytes / per-op time of the un-instrumented kernel: T(launch ending in front of op N + 1) - T(... op N), "
                    "best of 3 windows of 60 launches per N (nutls_profile_production)"}
        rep["production_timeline_us"] = {n: round(float(u), 3) for n, u in zip(pnames, pu[1:])} if args.profile_json else None
        if rep["production_timeline_us"] is None:
            del rep["production_timeline_us"]
    if args.profile_json:
        with open(args.profile_json, "w") as f:
            json.dump({"batch": B, "mode": mode, "timeline_step_ms": float(ms.sum()),
                       "ops": [{"layer": n, "ms": float(t), **by_layer.get(n.split("#")[0], {})} for n, t in zip(names, ms)]}, f, indent=1)
    return rep


if __name__ == "__main__":
    main()
