#!/usr/bin/env python3
"""bench.py -- UNet training throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size S] [--dtype bf16|fp32]

One "step" = one full training step (forward, BCE+Dice loss, backward,
grad-clip, AdamW) of the UNet on a synthetic floor-plan batch resident in HBM.
N>1: launched by torch.distributed.run, one rank per GPU, per-GPU batch fixed
(weak scaling), gradients all-reduced over RCCL in buckets overlapped with
backward.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3         # f32-input MFMA
GFLOP_TRAIN_512 = 1156.5        # per image, fwd + dgrad + wgrad


# HBM bytes per launch of the dominant kernel come from PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass
# each) summarised by scripts/pmc_summary.py into this tracked file; bench.py never runs rocprofv3 itself.
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r2_pmc_hbm_traffic.json")


def pmc_traffic(kernel_key, workload_key):
    """Measured HBM bytes per launch of `kernel_key` for `workload_key`, or None when no PMC summary was recorded for
    this workload / kernel source (the file carries a digest of the kernel sources it was measured on)."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            rec = json.load(f)
        from scripts.pmc_traffic import source_digest
        if rec.get("workload") != workload_key or rec.get("source_digest") != source_digest():
            return None
        return rec["kernels"][kernel_key]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError, ImportError):
        return None


def train_gflop_per_image(h, w):
    return GFLOP_TRAIN_512 * (h * w) / (512.0 * 512.0)


def conv_flops(args):
    # fea_conv3x3(x, ldx, w, bias, y, ldy, stats, N, H, W, Cin, Cout, dtype, stream)
    n, h, w, cin, cout = args[7:12]
    return 2.0 * n * h * w * cin * cout * 9


def cpu_baseline(size, seconds_budget=25.0):
    """The oracle (CPU port of the reference path) timed on this host's cores on
    a bounded sample of the same workload: batch 2 at the bench resolution."""
    from oracle import unet_oracle as O      # cpu_baseline leg only
    torch.manual_seed(0)
    n = 2
    sd = O.init_state_dict(0)
    params, bufs = O.split_state(sd)
    x, tgt = O.synthetic_batch(n, size, size, seed=1234)
    state = {}
    cores = torch.get_num_threads()

    def step():
        nonlocal params, bufs
        loss, _, grads, nb = O.loss_and_grads(params, bufs, x, tgt, training=True)
        clipped, _ = O.clip_grad_norm(grads, 1.0)
        params = O.adamw_step(params, clipped, state)
        bufs = nb
    t0 = time.time()
    step()                                    # warm-up
    warm = time.time() - t0
    k = max(1, min(3, int(seconds_budget / max(warm, 1e-3)) - 1))
    t0 = time.time()
    for _ in range(k):
        step()
    dt = (time.time() - t0) / k
    return {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle train step (fwd+BCE/Dice+bwd+clip+AdamW), batch {n} @ {size}x{size} fp32, {k} timed step(s) after 1 warm-up"}


def roofline(prof, nrep, a):
    """`roofline` object for the dominant kernel instantiation from the per-call HIP-event records of `nrep` steps."""
    per = {}
    dom = [0.0, 0, 0.0]                     # dominant kernel: ms, launches, flops
    dom_bytes = [0.0]                       # its algorithmic HBM bytes (activations in + out, weights once)
    for name, e0, e1, args in prof:
        ms = e0.elapsed_time(e1)
        d = per.setdefault(name, [0.0, 0, 0.0])
        d[0] += ms
        d[1] += 1
        fl = 0.0
        if name == "fea_conv3x3p":
            fl = conv_flops(args)
        elif name == "fea_conv3x3":
            fl = conv_flops(args)
            cout, wdt = args[11], args[9]
            # igemm_kernel<bf16, MODE 0, BN 128, EPI 0, TWS 5>: Cout % 128 == 0 and W >= 32
            if cout % 128 == 0 and wdt >= 32:
                dom[0] += ms; dom[1] += 1; dom[2] += fl
                n_, h_, w_, ci_, co_ = args[7:12]
                dom_bytes[0] += 2.0 * n_ * h_ * w_ * (ci_ + co_) + 2.0 * 9 * ci_ * co_
        elif name in ("fea_conv3x3_dgrad_bnred", "fea_conv3x3p_dgrad_bnred"):
            n, h, w, cin, cout = args[12:17]
            fl = 2.0 * n * h * w * cin * cout * 9
        elif name == "fea_conv3x3_wgrad":
            n, h, w, cin, cout = args[7:12]
            fl = 2.0 * n * h * w * cin * cout * 9
        d[2] += fl
    peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
    ach = dom[2] / (dom[0] * 1e-3) / 1e12
    workload_key = f"train batch {a.batch} {a.size}x{a.size} {a.dtype}"
    return {"bound": "mfma",
            "kernel": "igemm_kernel<bf16, MODE 0 (3x3), 128 columns, EPI 0, 8x32 tile> = fea_conv3x3 launches with Cout % 128 == 0 "
                      "(forward and data-gradient convolutions)",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            # HBM bytes per launch from the PMC passes ((2*FETCH_SIZE + WRITE_SIZE) * 1024 averaged over this kernel's
            # launches, MI355X_MICROARCH.md HBM section); null when no summary was recorded for this workload / source
            "traffic": pmc_traffic("igemm_kernel<bf16,0,128,0,5>", workload_key),
            "traffic_unit": "bytes per launch (HBM, PMC)", "algorithmic_bytes_per_launch": dom_bytes[0] / max(dom[1], 1),
            "avg_launch_ms": dom[0] / max(dom[1], 1), "launches_per_step": dom[1] // nrep,
            "algorithmic_flop_per_launch": "2*N*H*W*Cin*Cout*9 (summed over the launches)",
            "mfma_kernels_tflops": {k: v[2] / (v[0] * 1e-3) / 1e12 for k, v in per.items() if v[2] > 0},
            "kernel_ms_per_step": {k: v[0] / nrep for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])}}


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 from a plain shell: start N ranks (one per GPU) as a FRESH child process
    tree via torch.distributed.run, before this process has made any HIP / torch.cuda call (a process that has touched
    the GPU must never be re-exec'd), relay the child's output (rank 0's JSON line) and return its exit code."""
    have = torch.cuda.device_count()            # counting devices does not initialise the GPU
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} requested but this node exposes {have} GPU(s); not running a smaller job "
              "under the same label", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and a.gpus > 1:
        sys.exit(spawn_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
              "(or run `python bench.py --gpus N` directly, which starts the ranks itself)", file=sys.stderr)
        sys.exit(2)
    # FEA_BENCH_FORCE_DIST=1: take the RCCL / bucketed all-reduce path even with one rank (rehearsal)
    distributed = world > 1 or (os.environ.get("FEA_BENCH_FORCE_DIST") == "1" and launched)
    torch.cuda.set_device(local)
    if distributed:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        world = dist.get_world_size()          # n_gpus in the JSON line is what RCCL reports

    import fea_amd
    from fea_amd import _lib as L
    from fea_amd.data import synthetic_batch
    from fea_amd.train import UNetTrainer

    torch.manual_seed(0)                       # identical replicas on every rank
    model = fea_amd.FloorPlanSegmentationModel(compute_dtype=a.dtype).cuda()
    trainer = UNetTrainer(model, distributed=distributed)
    x, tgt = synthetic_batch(a.batch, a.size, a.size, seed=1234 + rank)
    x, tgt = x.cuda(), tgt.cuda()

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = trainer.step(x, tgt)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = trainer.step(x, tgt)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank alone (shows a straggler)
    sync()
    dt = time.perf_counter() - t0
    per_rank = [a.batch * a.steps / dt_own]
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([per_rank[0]], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(v.item()) for v in allr]
    loss_val = float(loss.item())

    # live per-kernel timing (HIP events on the launch stream, = torch's current stream).  Rank 0 only, with the
    # gradient all-reduce DETACHED: the other ranks do not run these extra steps, so no collective may be issued here.
    roof = None
    if rank == 0 and not a.no_kernel_timing:
        NREP = 3
        saved = (trainer.eng.grad_ready, trainer.buckets)
        trainer.eng.grad_ready, trainer.buckets = None, None
        L.PROFILE = []
        try:
            for _ in range(NREP):
                trainer.step(x, tgt)
            torch.cuda.synchronize()
        finally:
            prof, L.PROFILE = L.PROFILE, None
            trainer.eng.grad_ready, trainer.buckets = saved
        roof = roofline(prof, NREP, a)

    if rank == 0:
        ips = world * a.batch * a.steps / dt
        gf = train_gflop_per_image(a.size, a.size)
        out = {
            "metric": "UNet train images/sec @512x512 bf16" if (a.size == 512 and a.dtype == "bf16")
                      else f"UNet train images/sec @{a.size}x{a.size} {a.dtype}",
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"UNet (FloorPlanSegmentationModel) training step, {a.size}x{a.size}, batch {a.batch}/GPU, "
                                   f"{a.dtype} storage / fp32 accumulate, BCE+Dice loss, clip 1.0, AdamW"
                                   + (f" (BASELINE configs[2]{'/[3]' if world > 1 else ''})" if (a.batch, a.size, a.dtype) == (64, 512, "bf16")
                                      else (" (BASELINE configs[1])" if (a.batch, a.size, a.dtype) == (16, 256, "fp32") else "")),
                       "global_batch": world * a.batch, "parallelism": f"dp{world}", "per_gpu_batch": a.batch},
            "per_rank_images_per_s": per_rank,
            "whole_step_tflops_per_gpu": ips / world * gf / 1e3,
            "whole_step_mfma_frac": ips / world * gf / 1e3 / (PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS),
            "loss": loss_val,
        }
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.size)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
