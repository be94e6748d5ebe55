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
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r3_pmc_hbm_traffic.json")


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


def kernel_group(name, args, dtype):
    """C-ABI entry point + shape -> (kernel symbol as the rocprofv3 / PMC summaries name it, FLOP, algorithmic HBM bytes:
    activations in + out once, weights once) for the entry points that are ONE MFMA kernel symbol per launch."""
    bf = dtype == "bf16"
    if name == "fea_conv3x3_wgrad":
        # fea_conv3x3_wgrad(dy, lddy, x, ldx, dW, ws, ws_bytes, N, H, W, Cin, Cout, dtype, stream)
        n, h, w, cin, cout = args[7:12]
        fl = 2.0 * n * h * w * cin * cout * 9
        by = 2.0 * n * h * w * (cin + cout) + 4.0 * 9 * cin * cout
        if bf:      # wgrad2.hip: 8x32-pixel tiles when W >= 32, buffer-descriptor pieces on whole tiles (+ its slab reduction)
            tws = 5 if w >= 32 else 4
            fast = h % (256 >> tws) == 0 and w % (1 << tws) == 0
            if tws == 5 and fast and os.environ.get("FEA_AMD_WG4", "1") != "0":
                return "void wgrad4_kernel<true>", fl, by          # the 16x16x32, re-tiled form: default on whole 8x32 tiles
            return f"void wgrad2_kernel<{tws}, {'true' if fast else 'false'}>", fl, by
        return "wgrad_kernel (fp32)", fl, by
    if name == "fea_conv3x3":
        n, h, w, cin, cout = args[7:12]
        fl = 2.0 * n * h * w * cin * cout * 9
        by = 2.0 * n * h * w * (cin + cout) + 2.0 * 9 * cin * cout
        if bf:
            return f"igemm_kernel<bf16,0,{128 if cout % 128 == 0 else 64},0,{5 if w >= 32 else 4}>", fl, by
        return "igemm_kernel<f32,0>", fl, by
    if name == "fea_conv3x3p":
        n, h, w, cin, cout = args[7:12]
        st = 0 if args[6] is None else (2 if args[3] is not None else 1)
        return f"void c3p_kernel<{cin // 64}, 0, {st}>", 2.0 * n * h * w * cin * cout * 9, 2.0 * n * h * w * (cin + cout) + 2.0 * 9 * cin * cout
    if name in ("fea_conv3x3_dgrad_bnred", "fea_conv3x3p_dgrad_bnred"):
        n, h, w, cin, cout = args[12:17]
        by = 2.0 * n * h * w * (cin + 2 * cout) + 2.0 * 9 * cin * cout          # + the pre-BN tensor y
        sym = f"void c3p_kernel<{cin // 64}, 2, 0>" if name == "fea_conv3x3p_dgrad_bnred" else \
              f"igemm_kernel<bf16,0,{128 if cout % 128 == 0 else 64},2,{5 if w >= 32 else 4}>"
        return sym, 2.0 * n * h * w * cin * cout * 9, by
    return None


def roofline(prof, nrep, a):
    """`roofline` object for the DOMINANT kernel symbol of the measured step: the (entry point, kernel instantiation) group
    with the most HIP-event time over `nrep` profiled steps.  The runner-up is reported under `second`."""
    per = {}
    groups = {}                              # symbol -> [ms, launches, flops, bytes, entry point]
    for name, e0, e1, args in prof:
        ms = e0.elapsed_time(e1)
        d = per.setdefault(name, [0.0, 0, 0.0])
        d[0] += ms
        d[1] += 1
        kg = kernel_group(name, args, a.dtype)
        if kg is not None:
            sym, fl, by = kg
            d[2] += fl
            gr = groups.setdefault(sym, [0.0, 0, 0.0, 0.0, name])
            gr[0] += ms; gr[1] += 1; gr[2] += fl; gr[3] += by
    peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
    workload_key = f"train batch {a.batch} {a.size}x{a.size} {a.dtype}"

    def obj(sym):
        ms, cnt, fl, by, entry = groups[sym]
        ach = fl / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": sym, "entry_point": entry,
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                # HBM bytes per launch from the PMC passes ((2*FETCH_SIZE + WRITE_SIZE) * 1024 averaged over this kernel's
                # launches, MI355X_MICROARCH.md HBM section); null when no summary was recorded for this workload / source
                "traffic": pmc_traffic(sym, workload_key), "traffic_unit": "bytes per launch (HBM, PMC)",
                "algorithmic_bytes_per_launch": by / max(cnt, 1),
                "avg_launch_ms": ms / max(cnt, 1), "launches_per_step": cnt // nrep,
                "algorithmic_flop_per_launch": "2*N*H*W*Cin*Cout*9 (summed over the launches)"}
    order = sorted(groups, key=lambda k: -groups[k][0])
    out = obj(order[0])
    if out["entry_point"] == "fea_conv3x3_wgrad":
        out["note"] = "HIP events bracket the C-ABI call: the fixed-order slab reduction that follows the kernel is inside the time"
    if len(order) > 1:
        out["second"] = obj(order[1])
    out["mfma_kernels_tflops"] = {k: v[2] / (v[0] * 1e-3) / 1e12 for k, v in per.items() if v[2] > 0}
    out["kernel_ms_per_step"] = {k: v[0] / nrep for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])}
    return out


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 from a plain shell: start N ranks (one per GPU) as a FRESH child process
    tree via torch.distributed.run, before this process has made any HIP / torch.cuda call (a process that has touched
    the GPU must never be re-exec'd), relay the child's output (rank 0's JSON line) and return its exit code."""
    have = torch.cuda.device_count()            # counting devices does not initialise the GPU
    if have < a.gpus and os.environ.get("FEA_BENCH_SHARE_GPU") != "1":
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

    if os.environ.get("FEA_BENCH_DUMP_AFTER"):     # debugging aid: Python stacks of every thread to stderr after S seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FEA_BENCH_DUMP_AFTER"]), exit=True)
    # the host driver of this pool only supports dmabuf IPC: RCCL needs this before the first HIP call of every rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    # FEA_BENCH_BACKEND=gloo + FEA_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a box with fewer GPUs than ranks
    # (RCCL refuses two ranks on one device; gloo reduces device tensors through the host).  Never a scaling number:
    # the JSON line says so in config.parallelism.
    backend = os.environ.get("FEA_BENCH_BACKEND", "nccl")
    share = os.environ.get("FEA_BENCH_SHARE_GPU") == "1"
    if share:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()          # n_gpus in the JSON line is what RCCL reports

    import fea_amd
    from fea_amd import _lib as L
    from fea_amd.data import synthetic_batch
    from fea_amd.train import UNetTrainer

    torch.manual_seed(0)                       # identical replicas on every rank
    model = fea_amd.FloorPlanSegmentationModel(compute_dtype=a.dtype).cuda()
    trainer = UNetTrainer(model, distributed=distributed)
    # four different batches per rank, resident in HBM before the timed region, taken in turn (a step never sees the
    # batch of the step before it)
    batches = []
    for i in range(1 if os.environ.get("FEA_BENCH_ONE_BATCH") == "1" else 4):      # (A/B switch: one reused batch)
        bx, bt = synthetic_batch(a.batch, a.size, a.size, seed=1234 + rank + 1000 * i)
        batches.append((bx.cuda(), bt.cuda()))
    x, tgt = batches[0]

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        loss = trainer.step(*batches[i % len(batches)])
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = trainer.step(*batches[(a.warmup + i) % len(batches)])
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank alone (shows a straggler)
    sync()
    dt = time.perf_counter() - t0
    per_rank = [a.batch * a.steps / dt_own]
    if distributed:
        sdev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=sdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([per_rank[0]], dtype=torch.float64, device=sdev)
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
                       "global_batch": world * a.batch, "per_gpu_batch": a.batch, "distinct_batches_per_rank": len(batches),
                       "parallelism": f"dp{world}" + (f" (REHEARSAL: {backend}, ranks share GPUs)" if (share or backend != "nccl") and distributed else "")},
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
