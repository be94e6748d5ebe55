"""Host-side orchestration of the UNet forward/backward on the HIP kernels.

`UNetEngine` owns, per input shape, a *plan*: every activation / gradient /
workspace buffer laid out once in HBM (NHWC, compute dtype), so that steady
state training performs no allocation.  It mirrors, op for op, the dataflow of
the reference's `FloorPlanSegmentationModel.forward` (data_model/models.py:36-46)
and of its autograd backward (triggered at train_direct.py:104).

Data layout in HBM (N = batch, level l has spatial H/2^l x W/2^l):
  xcol            [N,H,W,Kpad]        3x3 im2col rows of the NCHW fp32 input
  y[layer]        [N,h,w,Cout]        raw conv output (pre-BN), kept for backward
  a[X.0]          [N,h,w,Cout]        relu(bn(y)) of the first conv of a block
  cat[k]          [N,h,w,2C]          concat buffer: upconv_k writes channels [0,C),
                                      enc_k.3's BN+ReLU writes the skip into [C,2C)
                                      (torch.cat([up, skip]) of models.py:42-45 is free)
  pool[k]         [N,h/2,w/2,C]       maxpool of the skip, input of the next encoder block
  a[dec_k.3], a[bottleneck.3]         dense
  dcat[k]         like cat[k]         gradient of the concat buffer (skip half lives until
                                      the encoder backward reaches level k)
  G0,G1,gmid      N*H*W*64 elements   rotating gradient buffers
  dY[0..2]        N*H*W*64 elements   ring of conv-output gradients: the weight-gradient
                                      kernels read them on a second stream while the main
                                      stream carries on with dgrad / BatchNorm backward
"""
from __future__ import annotations

import ctypes
import os
import struct
from typing import Callable, Dict, List, Optional

import torch

from . import _lib as L
from . import trace

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
KPAD_ALIGN = 64

ENC = ["enc1", "enc2", "enc3", "enc4"]
CH = {"enc1": 64, "enc2": 128, "enc3": 256, "enc4": 512, "bottleneck": 1024,
      "dec4": 512, "dec3": 256, "dec2": 128, "dec1": 64}
LEVEL = {"enc1": 0, "enc2": 1, "enc3": 2, "enc4": 3, "bottleneck": 4, "dec4": 3, "dec3": 2, "dec2": 1, "dec1": 0}
BLOCKS = ["enc1", "enc2", "enc3", "enc4", "bottleneck", "dec4", "dec3", "dec2", "dec1"]


def named_tensors(module, names) -> Dict[str, torch.Tensor]:
    """Parameters (or buffers) of `module` by dotted name, through plain attribute access.  Works on
    the replicas `nn.DataParallel` makes (train_direct.py:23-25), whose parameters are ordinary tensor
    attributes (outputs of the broadcast), not registered `nn.Parameter`s: `named_parameters()` is
    empty there."""
    out = {}
    for n in names:
        obj = module
        for part in n.split("."):
            obj = getattr(obj, part)
        out[n] = obj
    return out


def param_order(module) -> List[str]:
    """Parameter names in REVERSE-forward order: the order in which backward
    finishes their gradients (out, dec1, upconv1, dec2, ..., enc1).  The flat
    gradient buffer uses this order so that finished gradients form a growing
    prefix that can be all-reduced while backward continues."""
    names = ["out.weight", "out.bias"]
    for blk, up in (("dec1", "upconv1"), ("dec2", "upconv2"), ("dec3", "upconv3"), ("dec4", "upconv4")):
        for idx in (3, 0):
            names += [f"{blk}.{idx}.weight", f"{blk}.{idx}.bias", f"{blk}.{idx + 1}.weight", f"{blk}.{idx + 1}.bias"]
        names += [f"{up}.weight", f"{up}.bias"]
    for blk in ("bottleneck", "enc4", "enc3", "enc2", "enc1"):
        for idx in (3, 0):
            names += [f"{blk}.{idx}.weight", f"{blk}.{idx}.bias", f"{blk}.{idx + 1}.weight", f"{blk}.{idx + 1}.bias"]
    return names


def _table_rows(N, h, w) -> int:
    """Upper bound of the rows of a conv kernel's partial-sum table (`stats` / `part`): one per pixel tile for the
    one-tile-per-workgroup kernels, two per workgroup (at most 2 x 256, at most 2 per tile) for the persistent ones."""
    nt = L.call("fea_conv_num_tiles", N, h, w)
    split, t8 = L.load().fea_get_persistent_split(), N * ((h + 7) // 8) * ((w + 31) // 32)
    # (the persistent first layer: one row per wave, three workgroups per CU)
    return max(nt, min(512 * split, 2 * t8), min(3072 * split, 4 * t8))


class _Plan:
    """All device buffers for one (N, H, W) problem."""

    def __init__(self, eng: "UNetEngine", N: int, H: int, W: int):
        dt, dev = eng.dtype, eng.device
        self.N, self.H, self.W = N, H, W
        self.dims = [(H >> l, W >> l) for l in range(5)]
        es = torch.empty
        # im2col rows of the input: only the generic first-layer path needs them
        self.xcol = None if eng.first_direct else es((N, H, W, eng.kpad), dtype=dt, device=dev)
        self.x_in = None
        self.y: Dict[str, torch.Tensor] = {}
        self.a0: Dict[str, torch.Tensor] = {}
        self.a3: Dict[str, torch.Tensor] = {}
        self.cat: Dict[int, torch.Tensor] = {}
        self.dcat: Dict[int, torch.Tensor] = {}
        self.pool: Dict[int, torch.Tensor] = {}
        max_stats = 0
        for blk in BLOCKS:
            h, w = self.dims[LEVEL[blk]]
            c = CH[blk]
            for idx in (0, 3):
                self.y[f"{blk}.{idx}"] = es((N, h, w, c), dtype=dt, device=dev)
            self.a0[blk] = es((N, h, w, c), dtype=dt, device=dev)
            if blk not in ENC:
                self.a3[blk] = es((N, h, w, c), dtype=dt, device=dev)
            # forward: Cout columns; decoder dgrad of X.0 also takes sums of its 2*Cout-wide output
            max_stats = max(max_stats, _table_rows(N, h, w) * (c if blk in ENC or blk == "bottleneck" else 2 * c) * 2)
        for k, blk in enumerate(ENC, start=1):
            h, w = self.dims[k - 1]
            c = CH[blk]
            self.cat[k] = es((N, h, w, 2 * c), dtype=dt, device=dev)
            self.pool[k] = es((N, h // 2, w // 2, c), dtype=dt, device=dev)
        self.stats = es(max_stats, dtype=torch.float32, device=dev)
        self.logits = es((N, eng.num_classes, H, W), dtype=torch.float32, device=dev)
        # per-layer BN coefficients (kept from forward to backward)
        self.coef: Dict[str, Dict[str, torch.Tensor]] = {}
        for blk in BLOCKS:
            for idx in (0, 3):
                c = CH[blk]
                self.coef[f"{blk}.{idx}"] = {k: es(c, dtype=torch.float32, device=dev)
                                             for k in ("scale", "shift", "mean", "invstd")}
        self.nbytes = sum(t.numel() * t.element_size() for t in self._tensors())
        self.bwd_ready = False
        # > 0 while an autograd graph still needs this plan's activations for its backward
        # (the reference's pipeline calls the UNet once per floor before one backward,
        # models.py:190): such a plan is not reused by the next forward.
        self.leases = 0

    def _tensors(self):
        for k, v in vars(self).items():
            if k in ("P", "x_in"):               # the caller's tensors, not the plan's
                continue
            if isinstance(v, torch.Tensor):
                yield v
            elif isinstance(v, dict):
                for u in v.values():
                    if isinstance(u, torch.Tensor):
                        yield u
                    elif isinstance(u, dict):
                        yield from (w for w in u.values() if isinstance(w, torch.Tensor))
            elif isinstance(v, list):
                yield from (u for u in v if isinstance(u, torch.Tensor))

    def alloc_backward(self, eng: "UNetEngine"):
        if self.bwd_ready:
            return
        dt, dev = eng.dtype, eng.device
        N, H, W = self.N, self.H, self.W
        es = torch.empty
        big = N * H * W * 64
        self.G = [es(big, dtype=dt, device=dev) for _ in range(2)]
        self.gmid = es(big, dtype=dt, device=dev)
        self.dYs = [es(big, dtype=dt, device=dev) for _ in range(3)]
        self.dY_free = [None, None, None]     # event: the wgrad that read this slot has run
        self.dY_next = 0
        for k, blk in enumerate(ENC, start=1):
            h, w = self.dims[k - 1]
            self.dcat[k] = es((N, h, w, 2 * CH[blk]), dtype=dt, device=dev)
        code = eng.code
        ws = 0
        lib = L.load()
        for blk in BLOCKS:
            h, w = self.dims[LEVEL[blk]]
            c = CH[blk]
            cin0 = eng.block_cin(blk)
            if blk == "enc1":
                ws = max(ws, lib.fea_first_wgrad_ws_bytes(N, h, w, c) if eng.first_direct
                         else lib.fea_wgrad_ws_bytes(1, N, h, w, c, eng.kpad, code))
            else:
                ws = max(ws, lib.fea_wgrad_ws_bytes(0, N, h, w, c, cin0, code))
            ws = max(ws, lib.fea_wgrad_ws_bytes(0, N, h, w, c, c, code))
        for k in (1, 2, 3, 4):
            h, w = self.dims[k]
            c = CH[ENC[k - 1]]
            ws = max(ws, lib.fea_wgrad_ws_bytes(2, N, h, w, 2 * c, c, code))
        self.ws_bytes = int(ws)
        self.ws = es(self.ws_bytes // 4, dtype=torch.float32, device=dev)
        # rows of `part`: the BN-backward kernels use fea_bn_bwd_blocks() rows, the fused
        # dgrad epilogue (fea_conv3x3_dgrad_bnred) one row per conv pixel tile
        nb = max(max(L.call("fea_bn_bwd_blocks", N, *self.dims[LEVEL[b]], CH[b]),
                     _table_rows(N, *self.dims[LEVEL[b]])) * CH[b] * 2 for b in BLOCKS)
        nb = max(nb, 2048 * 1024 * 2)
        self.part = es(nb, dtype=torch.float32, device=dev)
        self.bncoef = es(1024 * 2, dtype=torch.float32, device=dev)
        self.head_part = es(L.call("fea_head_bwd_blocks", N, H, W) * eng.num_classes * 65, dtype=torch.float32, device=dev)
        self.bwd_ready = True
        self.nbytes = sum(t.numel() * t.element_size() for t in self._tensors())


class UNetEngine:
    def __init__(self, module, compute_dtype: torch.dtype):
        self.m = module
        self.dtype = compute_dtype
        self.code = L.dt_code(compute_dtype)
        self.in_channels = module.in_channels
        self.num_classes = module.num_classes
        self.kpad = (9 * self.in_channels + KPAD_ALIGN - 1) // KPAD_ALIGN * KPAD_ALIGN
        # bf16 and 9*C <= 64: the first layer reads the fp32 NCHW input directly
        # (fea_conv3x3_first / _first_wgrad); otherwise im2col rows + 1x1 GEMM
        self.first_direct = compute_dtype == torch.bfloat16 and 9 * self.in_channels <= 64
        self.device = None
        self.plans: Dict[tuple, _Plan] = {}
        self.packed: Dict[str, torch.Tensor] = {}
        self.packed_ver: Dict[str, int] = {}
        # eval mode, bf16: BatchNorm folded into packed weights + bias (name -> (version, w, bias))
        self.fold_bn = os.environ.get("FEA_AMD_FOLD_BN", "1") != "0"
        self.folded: Dict[str, tuple] = {}
        self.flat_grad: Optional[torch.Tensor] = None
        self.grad_views: Dict[str, torch.Tensor] = {}
        self.grad_offsets: Dict[str, tuple] = {}
        # called as grad_ready(name) right after the kernels that finish a
        # parameter's gradient have been enqueued (used for overlapped all-reduce)
        self.grad_ready: Optional[Callable[[str], None]] = None
        self.debug_taps: Optional[dict] = None
        # Packed-weight cache generation.  Parameter._version is NOT bumped by
        # fused optimizers (torch._fused_adamw_), so in training mode every
        # forward starts a new generation (weights are re-packed once per step,
        # ~31 M elements: negligible); eval-mode forwards reuse the packs until
        # the module leaves eval mode, a parameter's version/storage changes or
        # invalidate() is called.
        self._gen = 0
        # the packs a training step made one by one, replayed by the following steps in one launch (_pack_all)
        self.batch_packs = os.environ.get("FEA_AMD_BATCH_PACKS", "1") != "0"
        self._pack_jobs: Dict[str, tuple] = {}
        self._pack_table = None
        # FEA_AMD_WGRAD_STREAM=1: weight-gradient kernels run on a second stream beside the BatchNorm-backward kernels
        # of the following layers.  Off by default: the kernels do co-run (kernel trace), but the step is no shorter
        # (85.2 vs 85.4 ms) -- the shallow levels' weight gradients are themselves HBM-heavy (2.2 TB/s), so the two
        # kernels share the same bandwidth -- and a single stream keeps per-kernel profiles equal to what the step runs.
        self.overlap_wgrad = os.environ.get("FEA_AMD_WGRAD_STREAM", "0") == "1"
        self._wstream = None
        self._P: Dict[str, torch.Tensor] = {}
        self._B: Dict[str, torch.Tensor] = {}
        # persistent conv kernels for the shallow levels (bf16); FEA_AMD_PERSISTENT=0 selects the plain kernels (A/B runs)
        self.use_persistent = compute_dtype == torch.bfloat16 and os.environ.get("FEA_AMD_PERSISTENT", "1") != "0"
        self._p_rows_cache: Dict[tuple, int] = {}
        # fused-BatchNorm-reduction data gradient of the 128 -> 128 layers on the persistent kernel too (round 3: its
        # scale / shift / mean moved to LDS, no more spills); FEA_AMD_P_EPI2_128=0 selects the one-tile-per-workgroup kernel (A/B runs)
        self.p_epi2_128 = os.environ.get("FEA_AMD_P_EPI2_128", "1") != "0"
        # work-split factor of the persistent kernels (fea_set_persistent_split): FEA_AMD_PERSISTENT_SPLIT overrides the library default
        if os.environ.get("FEA_AMD_PERSISTENT_SPLIT"):
            L.call("fea_set_persistent_split", int(os.environ["FEA_AMD_PERSISTENT_SPLIT"]))
        # dec1.3: BatchNorm + ReLU applied inside the fused head + loss kernels' loaders (FEA_AMD_FUSE_HEAD_BN=0: separate pass)
        self.fuse_head_bn = os.environ.get("FEA_AMD_FUSE_HEAD_BN", "1") != "0"
        # ... and its backward as well (fea_head_loss_bwd_bnred / _bnapply; FEA_AMD_FUSE_HEAD_BNBWD=0: da stored, separate reduce / apply)
        self.fuse_head_bnbwd = os.environ.get("FEA_AMD_FUSE_HEAD_BNBWD", "1") != "0"
        # enc1.0: BatchNorm backward apply fused into the first layer's weight gradient (FEA_AMD_FUSE_FIRST_WGRAD=0: separate kernels)
        self.fuse_first_wgrad = os.environ.get("FEA_AMD_FUSE_FIRST_WGRAD", "1") != "0"
        self.backward_count = 0        # backward passes run so far (ClipAdamW checks that it steps on one backward's gradients)
        L.load()

    # ------------------------------------------------------------------ utils
    def block_cin(self, blk: str) -> int:
        return {"enc1": self.in_channels, "enc2": 64, "enc3": 128, "enc4": 256, "bottleneck": 512,
                "dec4": 1024, "dec3": 512, "dec2": 256, "dec1": 128}[blk]

    def _params(self) -> Dict[str, torch.Tensor]:
        return named_tensors(self.m, self.m._param_names)

    def _bufs(self) -> Dict[str, torch.Tensor]:
        return dict(self.m.named_buffers())

    def _ensure_device(self, x: torch.Tensor):
        if self.device != x.device:
            self.device = x.device
            self.plans.clear()
            self.packed.clear()
            self.packed_ver.clear()
            self.folded.clear()
            self.flat_grad = None

    # Forwards whose backward is still pending each hold a plan (saved activations; ~45 GB at batch
    # 64 x 512^2 bf16 once its backward buffers exist): their total is bounded in BYTES, as a
    # fraction of the device's memory, not by a count.
    MAX_LEASED_FRACTION = 0.8

    def plan(self, N, H, W) -> _Plan:
        """A plan for this shape whose buffers nobody is waiting to run backward on."""
        key = (N, H, W)
        pool = self.plans.get(key)
        if pool is None:
            for k in [k for k, v in self.plans.items() if all(q.leases == 0 for q in v)]:
                if len(self.plans) >= 2:      # keep at most two idle shapes resident
                    self.plans.pop(k)
            pool = self.plans[key] = []
        for p in pool:
            if p.leases == 0:
                return p
        if pool:
            # every resident plan of this shape is leased: one more costs what they cost (with backward buffers)
            held = sum(q.nbytes for v in self.plans.values() for q in v)
            est = max(q.nbytes for q in pool)
            total = torch.cuda.get_device_properties(self.device).total_memory
            if held + est > self.MAX_LEASED_FRACTION * total:
                raise L.FeaError(f"{len(pool)} forward passes of shape {key} are waiting for their backward and hold "
                                 f"{held / 2**30:.1f} GiB; one more (~{est / 2**30:.1f} GiB) would exceed "
                                 f"{self.MAX_LEASED_FRACTION:.0%} of the device's {total / 2**30:.0f} GiB: run backward (or drop "
                                 "the graphs / use torch.no_grad()) before calling the model again")
        p = _Plan(self, N, H, W)
        pool.append(p)
        return p

    def _ensure_flat_grad(self):
        if self.flat_grad is not None:
            return
        params = self._params()
        order = param_order(self.m)
        assert set(order) == set(params), "parameter set differs from the reference's"
        self._P = params
        off = 0
        for n in order:
            numel = params[n].numel()
            self.grad_offsets[n] = (off, numel)
            off += (numel + 63) // 64 * 64           # 256-byte aligned slices
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad_views = {n: self.flat_grad[o:o + k].view(params[n].shape) for n, (o, k) in self.grad_offsets.items()}

    # -------------------------------------------------------------- packing
    _PACK_KINDS = {"fea_pack_conv3x3": 0, "fea_pack_conv3x3_frag": 1, "fea_pack_upconv2x2": 2, "fea_pack_conv3x3_first": 3}

    def _pack(self, key: str, param: torch.nn.Parameter, numel: int, fn: str, *args):
        ver = (self._gen, param._version, param.data_ptr())
        buf = self.packed.get(key)
        if buf is None:
            buf = torch.empty(numel, dtype=self.dtype, device=self.device)
            self.packed[key] = buf
            self.packed_ver[key] = -1
        if self.packed_ver[key] != ver:
            L.call(fn, param.detach(), buf, *args, self.code, L.stream_ptr())
            self.packed_ver[key] = ver
            kind = self._PACK_KINDS.get(fn)
            if kind is not None and self.batch_packs and key not in self._pack_jobs:
                # a pack this training step needed one by one: the next step makes it in the batched launch
                a, b = args[0], args[1]
                c = args[2] if len(args) > 2 else 0
                self._pack_jobs[key] = (key.split("/")[0] + ".weight", buf, kind, a, b, c)
                self._pack_table = None
        return buf

    def _pack_all(self):
        """Training mode re-packs every weight once per step (~43 tensors, most of them far smaller than a launch is
        long): replay the packs the previous steps made, in ONE launch (`fea_pack_batch`).  Jobs are keyed like the pack
        cache; a job whose parameter or buffer is no longer the recorded one drops out and is re-learned."""
        jobs, P = self._pack_jobs, self._P
        if not jobs:
            return

        def usable(k, job):
            p = P.get(job[0])
            return (self.packed.get(k) is job[1] and p is not None and p.device == self.device
                    and p.dtype == torch.float32 and p.is_contiguous())
        stale = [k for k, job in jobs.items() if not usable(k, job)]
        for k in stale:
            del jobs[k]
            self._pack_table = None
        if not jobs or len(jobs) > 64:
            return
        ptrs = tuple(P[job[0]].data_ptr() for job in jobs.values())      # the tensors THIS forward uses (a replica's own)
        if self._pack_table is None or self._pack_table[0] != ptrs:
            recs, first = [], 0
            for src, (_pn, buf, kind, a, b, c) in zip(ptrs, jobs.values()):
                recs.append(struct.pack("<QQiiiiii", src, buf.data_ptr(), kind, a, b, c, first, 0))
                first += L.call("fea_pack_job_blocks", kind, a, b, c)
            table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(self.device)
            self._pack_table = (ptrs, table, len(recs), first)
        _, table, n, nblocks = self._pack_table
        L.call("fea_pack_batch", table, n, nblocks, self.code, L.stream_ptr())
        for k, job in jobs.items():
            p = P[job[0]]
            self.packed_ver[k] = (self._gen, p._version, p.data_ptr())

    def invalidate(self):
        """Forget packed weights (call after parameters were changed in place)."""
        self._gen += 1

    def _p_rows(self, N, h, w, cin, cout) -> int:
        """Rows of the partial-sum tables of the persistent conv kernels (csrc/conv_p.hip) for this
        GEMM shape, 0 when they do not serve it (then the one-tile-per-workgroup kernels run)."""
        if not self.use_persistent:
            return 0
        key = (N, h, w, cin, cout)
        r = self._p_rows_cache.get(key)
        if r is None:
            r = self._p_rows_cache[key] = L.call("fea_conv3x3p_rows", N, h, w, cin, cout, self.code)
        return r

    def _w_fwd(self, name: str, frag: bool = False):
        p = self._P[name + ".weight"]
        if frag:
            return self._pack(name + "/fp", p, p.numel(), "fea_pack_conv3x3_frag", p.shape[0], p.shape[1], 0)
        if name == "enc1.0" and self.first_direct:
            ks = (9 * self.in_channels + 31) // 32
            return self._pack(name + "/f", p, ks * 64 * 32, "fea_pack_conv3x3_first", 64, self.in_channels)
        if name == "enc1.0":
            return self._pack(name + "/f", p, 64 * self.kpad, "fea_pack_conv3x3_im2col", 64, self.in_channels, self.kpad)
        co, ci = p.shape[0], p.shape[1]
        return self._pack(name + "/f", p, p.numel(), "fea_pack_conv3x3", co, ci, 0)

    def _w_folded(self, name: str, frag: bool = False):
        """Packed weights and bias of conv `name` with its eval-mode BatchNorm folded in:
        w' = w * gamma / sqrt(running_var + eps), b' = (b - running_mean) * that + beta
        (models.py:28-29 / 31-32 in eval mode).  Re-made when any of the six tensors changes."""
        blk, idx = name.rsplit(".", 1)
        bn = f"{blk}.{int(idx) + 1}"
        P, B = self._P, self._B
        deps = [P[name + ".weight"], P[name + ".bias"], P[bn + ".weight"], P[bn + ".bias"],
                B[bn + ".running_mean"], B[bn + ".running_var"]]
        ver = (self._gen, tuple(t._version for t in deps), tuple(t.data_ptr() for t in deps))
        ent = self.folded.get((name, frag))
        if ent is None or ent[0] != ver:
            with torch.no_grad():
                scale = deps[2] * torch.rsqrt(deps[5] + BN_EPS)
                wf = (deps[0] * scale.view(-1, 1, 1, 1)).contiguous()
                bf = ((deps[1] - deps[4]) * scale + deps[3]).contiguous()
            co, ci = wf.shape[0], wf.shape[1]
            if name == "enc1.0":
                ks = (9 * self.in_channels + 31) // 32
                numel, fn, args = ks * 64 * 32, "fea_pack_conv3x3_first", (64, self.in_channels)
            else:
                numel, fn, args = wf.numel(), ("fea_pack_conv3x3_frag" if frag else "fea_pack_conv3x3"), (co, ci, 0)
            buf = ent[1] if ent is not None else torch.empty(numel, dtype=self.dtype, device=self.device)
            L.call(fn, wf, buf, *args, self.code, L.stream_ptr())
            ent = (ver, buf, bf)
            self.folded[(name, frag)] = ent
        return ent[1], ent[2]

    def _w_dgrad(self, name: str, frag: bool = False):
        p = self._P[name + ".weight"]
        co, ci = p.shape[0], p.shape[1]
        if frag:
            return self._pack(name + "/dp", p, p.numel(), "fea_pack_conv3x3_frag", co, ci, 1)
        return self._pack(name + "/d", p, p.numel(), "fea_pack_conv3x3", co, ci, 1)

    def _p_modes(self, pl: "_Plan"):
        """Per conv layer: (rows of the persistent forward kernel, rows of the persistent data-gradient kernel), 0 = not
        served.  X.3 layers' data gradient is the fused-BatchNorm-reduction kernel, served for 64 and 128 channels."""
        modes = getattr(pl, "p_modes", None)
        if modes is None:
            modes = {}
            for blk in BLOCKS:
                h, w = pl.dims[LEVEL[blk]]
                c, cin = CH[blk], self.block_cin(blk)
                f0 = self._p_rows(pl.N, h, w, cin, c) if blk != "enc1" else 0
                d0 = self._p_rows(pl.N, h, w, c, cin) if blk != "enc1" else 0
                f3 = self._p_rows(pl.N, h, w, c, c)
                d3 = f3 if (c == 64 or (c == 128 and self.p_epi2_128)) else 0
                modes[f"{blk}.0"], modes[f"{blk}.3"] = (f0, d0), (f3, d3)
            pl.p_modes = modes
        return modes

    def _w_up(self, name: str, bwd: int):
        p = self._P[name + ".weight"]
        ci, co = p.shape[0], p.shape[1]
        return self._pack(name + ("/b" if bwd else "/f"), p, p.numel(), "fea_pack_upconv2x2", ci, co, bwd)

    # -------------------------------------------------------------- forward
    @trace.traced("unet.forward")
    def forward(self, x: torch.Tensor, training: bool, head: bool = True) -> Optional[torch.Tensor]:
        """`head=False` (training loops that use the fused head + loss kernels, see backward(target=...)): stop after
        dec1; the logits are formed inside the loss kernels and never stored."""
        if not x.is_cuda:
            raise L.FeaError("fea_amd runs on a ROCm device only (no CPU fallback): move the module and input to 'cuda'")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input [N,{self.in_channels},H,W], got {tuple(x.shape)}")
        N, _, H, W = x.shape
        if H % 16 or W % 16:
            # the reference fails at the first torch.cat for such sizes
            raise RuntimeError(f"Sizes of tensors must match: H and W must be multiples of 16, got {H}x{W}")
        self._ensure_device(x)
        if training:
            self._gen += 1
        x = x.detach().contiguous().float()
        pl = self.plan(N, H, W)
        P, B = self._params(), self._bufs()
        self._P, self._B = P, B
        if training and self.batch_packs:
            self._pack_all()
        pl.P = P                                  # the tensors this forward used (a DataParallel replica's own copies)
        S = L.stream_ptr()
        code = self.code
        taps = self.debug_taps

        pl.x_in = x
        # training step with the fused head + loss kernels: dec1.3's activated tensor is never written
        pl.head_bn = bool(training and not head and taps is None and self.fuse_head_bn
                          and L.call("fea_head_loss_bn_supported", N, H, W, self.num_classes, code))
        fold = (not training) and self.fold_bn and self.dtype == torch.bfloat16 and taps is None
        # Training: the ~40 weight re-packs of the step (forward + data-gradient layouts, 0.4 ms of
        # small launches) run on the second stream beside the first layer instead of in front of
        # each conv.  The second stream first waits for everything enqueued so far (the optimiser).
        packs_done = None
        if training and self.overlap_wgrad and L.PROFILE is None:
            main = torch.cuda.current_stream()
            if self._wstream is None or self._wstream.device != self.device:
                self._wstream = torch.cuda.Stream(device=self.device)
            ev0 = torch.cuda.Event()
            ev0.record(main)
            pm = self._p_modes(pl)
            with torch.cuda.stream(self._wstream):
                self._wstream.wait_event(ev0)
                for blk in BLOCKS:
                    for idx in (0, 3):
                        if not (blk == "enc1" and idx == 0):
                            self._w_fwd(f"{blk}.{idx}", frag=pm[f"{blk}.{idx}"][0] > 0)
                for k in (1, 2, 3, 4):
                    self._w_up(f"upconv{k}", 0)
                packs_done = torch.cuda.Event()
                packs_done.record(self._wstream)
                for blk in BLOCKS:
                    for idx in (0, 3):
                        if not (blk == "enc1" and idx == 0):
                            self._w_dgrad(f"{blk}.{idx}", frag=pm[f"{blk}.{idx}"][1] > 0)
                for k in (1, 2, 3, 4):
                    self._w_up(f"upconv{k}", 1)
                pl.bwd_packs_done = torch.cuda.Event()
                pl.bwd_packs_done.record(self._wstream)
        else:
            pl.bwd_packs_done = None
        if not self.first_direct:
            L.call("fea_im2col3x3_nchw", x, pl.xcol, N, self.in_channels, H, W, self.kpad, code, S)

        waited = [False]

        def conv_bn(name, inp, ld_in, cin, h, w, cout, out_full, ld_full, out_pool):
            """conv -> BN statistics -> scale/shift -> ReLU(+pool)"""
            if fold and (name != "enc1.0" or self.first_direct):
                # eval mode: one kernel, relu(conv(x, w') + b') (+ pooled copy)
                rows_p = self._p_modes(pl)[name][0]
                wp, bf = self._w_folded(name, frag=rows_p > 0)
                if name == "enc1.0":
                    L.call("fea_conv3x3_first_bias_relu", pl.x_in, wp, bf, out_full, ld_full, N, self.in_channels, h, w, cout, code, S)
                else:
                    L.call("fea_conv3x3p_bias_relu" if rows_p else "fea_conv3x3_bias_relu", inp, ld_in, wp, bf, out_full, ld_full,
                           out_pool, cout if out_pool is not None else 0, N, h, w, cin, cout, code, S)
                return
            y = pl.y[name]
            stats = pl.stats if training else None
            if packs_done is not None and name != "enc1.0" and not waited[0]:
                torch.cuda.current_stream().wait_event(packs_done)
                waited[0] = True
            rows_p = self._p_modes(pl)[name][0] if training else 0
            wp = self._w_fwd(name, frag=rows_p > 0)
            bias = P[name + ".bias"]
            if rows_p:
                L.call("fea_conv3x3p", inp, ld_in, wp, bias, y, cout, stats, N, h, w, cin, cout, code, S)
            elif name == "enc1.0" and self.first_direct:
                L.call("fea_conv3x3_first", pl.x_in, wp, bias, y, cout, stats, N, self.in_channels, h, w, cout, code, S)
            elif name == "enc1.0":
                L.call("fea_conv1x1", inp, ld_in, wp, bias, y, cout, stats, N, h, w, self.kpad, cout, code, S)
            else:
                L.call("fea_conv3x3", inp, ld_in, wp, bias, y, cout, stats, N, h, w, cin, cout, code, S)
            blk, idx = name.rsplit(".", 1)
            bn = f"{blk}.{int(idx) + 1}"
            cf = pl.coef[name]
            if training:
                if name == "enc1.0" and self.first_direct:
                    rows_p = L.call("fea_conv3x3_first_rows", N, self.in_channels, h, w, cout, code)
                ntiles = rows_p or L.call("fea_conv_num_tiles", N, h, w)
                L.call("fea_bn_finalize", stats, ntiles, N * h * w, P[bn + ".weight"], P[bn + ".bias"],
                       B[bn + ".running_mean"], B[bn + ".running_var"], BN_MOMENTUM, BN_EPS,
                       cf["scale"], cf["shift"], cf["mean"], cf["invstd"], cout, S)
            else:
                L.call("fea_bn_eval_coeffs", P[bn + ".weight"], P[bn + ".bias"], B[bn + ".running_mean"],
                       B[bn + ".running_var"], BN_EPS, cf["scale"], cf["shift"], cout, S)
            if name == "dec1.3" and pl.head_bn:
                return          # the fused head + loss kernels apply BatchNorm + ReLU in their loaders (fea_head_loss_*_bn)
            L.call("fea_bn_relu_pool", y, cf["scale"], cf["shift"], out_full, ld_full, out_pool, N, h, w, cout, code, S)

        es = torch.empty(0, dtype=self.dtype).element_size()

        def off_ptr(t: torch.Tensor, ch_off: int) -> int:
            return t.data_ptr() + ch_off * es

        # encoder
        inp, ld_in, cin = pl.xcol, self.kpad, self.kpad
        for k, blk in enumerate(ENC, start=1):
            h, w = pl.dims[k - 1]
            c = CH[blk]
            conv_bn(f"{blk}.0", inp, ld_in, cin, h, w, c, pl.a0[blk], c, None)
            conv_bn(f"{blk}.3", pl.a0[blk], c, c, h, w, c, off_ptr(pl.cat[k], c), 2 * c, pl.pool[k])
            inp, ld_in, cin = pl.pool[k], c, c
        h, w = pl.dims[4]
        conv_bn("bottleneck.0", inp, 512, 512, h, w, 1024, pl.a0["bottleneck"], 1024, None)
        conv_bn("bottleneck.3", pl.a0["bottleneck"], 1024, 1024, h, w, 1024, pl.a3["bottleneck"], 1024, None)
        d, cd = pl.a3["bottleneck"], 1024
        # decoder
        for k in (4, 3, 2, 1):
            blk = f"dec{k}"
            hl, wl = pl.dims[k]            # low-res dims (input of the transposed conv)
            h, w = pl.dims[k - 1]
            c = CH[blk]
            L.call("fea_upconv2x2_fwd", d, cd, self._w_up(f"upconv{k}", 0), P[f"upconv{k}.bias"], pl.cat[k], 2 * c,
                   N, hl, wl, cd, c, code, S)
            conv_bn(f"{blk}.0", pl.cat[k], 2 * c, 2 * c, h, w, c, pl.a0[blk], c, None)
            conv_bn(f"{blk}.3", pl.a0[blk], c, c, h, w, c, pl.a3[blk], c, None)
            d, cd = pl.a3[blk], c
        if head:
            L.call("fea_head_fwd", d, 64, P["out.weight"], P["out.bias"], pl.logits, N, H, W, 64, self.num_classes, code, S)
        if training:
            torch._foreach_add_([B[f"{b}.{i}.num_batches_tracked"] for b in BLOCKS for i in (1, 4)], 1)
        if taps is not None:
            self._export_taps(pl, taps)
        pl.training = training
        self._last = (pl, training)
        return pl.logits if head else None

    def _export_taps(self, pl: _Plan, taps: dict):
        """Debug/test hook: activated outputs (NCHW fp32) of every conv layer."""
        for blk in BLOCKS:
            c = CH[blk]
            taps[f"{blk}.0.act"] = pl.a0[blk].float().permute(0, 3, 1, 2).contiguous()
            taps[f"{blk}.0.conv"] = pl.y[f"{blk}.0"].float().permute(0, 3, 1, 2).contiguous()
            taps[f"{blk}.3.conv"] = pl.y[f"{blk}.3"].float().permute(0, 3, 1, 2).contiguous()
            if blk in ENC:
                k = ENC.index(blk) + 1
                taps[f"{blk}.3.act"] = pl.cat[k][..., c:].float().permute(0, 3, 1, 2).contiguous()
            else:
                taps[f"{blk}.3.act"] = pl.a3[blk].float().permute(0, 3, 1, 2).contiguous()

    # -------------------------------------------------------------- backward
    def fused_loss_supported(self, H: int, W: int) -> bool:
        return bool(L.call("fea_head_loss_supported", H, W, self.num_classes))

    @trace.traced("unet.backward")
    def backward(self, dlogits: Optional[torch.Tensor], plan: Optional[_Plan] = None,
                 target: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Gradients of every parameter for upstream gradient `dlogits`
        ([N,classes,H,W] fp32).  Returns views into the flat gradient buffer.
        `plan`: the plan of the forward this is the backward of (default: the last forward).
        `target` (uint8 [N,classes,H,W]) with dlogits=None: the upstream is the BCE+Dice loss of the logits against
        `target`; head, loss and their backward run fused (fea_head_loss_sums / _bwd), the loss lands in `plan.loss`."""
        pl, training = self._last if plan is None else (plan, plan.training)
        if not training:
            raise L.FeaError("backward through eval-mode BatchNorm is not supported (the reference trains in train() mode)")
        pl.alloc_backward(self)
        self._ensure_flat_grad()
        if getattr(pl, "bwd_packs_done", None) is not None:
            torch.cuda.current_stream().wait_event(pl.bwd_packs_done)
            pl.bwd_packs_done = None
        N, H, W = pl.N, pl.H, pl.W
        P = pl.P
        self._P = P
        self.backward_count += 1
        S = L.stream_ptr()
        code = self.code
        g = self.grad_views
        dl = dlogits.detach().contiguous().float() if dlogits is not None else None
        if dl is None and target is None:
            raise L.FeaError("backward needs dlogits or a loss target")
        es = torch.empty(0, dtype=self.dtype).element_size()

        def off_ptr(t, ch_off):
            return t.data_ptr() + ch_off * es

        def done(*names):
            if self.grad_ready is not None:
                for n in names:
                    self.grad_ready(n)

        main = torch.cuda.current_stream()
        side = None
        if self.overlap_wgrad and L.PROFILE is None:      # per-call timing wants one stream
            if self._wstream is None or self._wstream.device != self.device:
                self._wstream = torch.cuda.Stream(device=self.device)
            side = self._wstream
        ring = {"dY": pl.dYs[0], "slot": 0}

        def next_dY():
            """Next slot of the dY ring (waits until the wgrad that last read it has run)."""
            i = pl.dY_next
            pl.dY_next = (i + 1) % len(pl.dYs)
            if pl.dY_free[i] is not None:
                main.wait_event(pl.dY_free[i])
                pl.dY_free[i] = None
            ring["dY"], ring["slot"] = pl.dYs[i], i
            return pl.dYs[i]

        def wgrad(fn, args_before_stream, names, reads_dY=True):
            """Enqueue a weight-gradient kernel (second stream when overlapping) and report
            `names` ready behind it.  Everything enqueued on the main stream so far (the dY it
            reads, the bias / BatchNorm gradients in `names`) is ordered before it."""
            run = fn if callable(fn) else (lambda *a: L.call(fn, *a))
            if side is None:
                run(*args_before_stream, S)
                done(*names)
                return
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                run(*args_before_stream, side.cuda_stream)
                if reads_dY:
                    fe = torch.cuda.Event()
                    fe.record(side)
                    pl.dY_free[ring["slot"]] = fe
                done(*names)

        fuse = self.dtype == torch.bfloat16     # dgrad kernels also produce the next BN-backward's partial sums
        pm = self._p_modes(pl)

        def bn_bwd(name, da_full, ld_da, dpool, h, w, c, reduced_tiles=0):
            """reduced_tiles > 0: pl.part already holds that many per-tile partial sums
            (written by the kernel that produced da_full)."""
            blk, idx = name.rsplit(".", 1)
            bn = f"{blk}.{int(idx) + 1}"
            cf = pl.coef[name]
            y = pl.y[name]
            nblk = L.call("fea_bn_bwd_blocks", N, h, w, c)
            dY = next_dY()
            if reduced_tiles:
                nfin = reduced_tiles
            else:
                nfin = nblk
                L.call("fea_bn_relu_bwd_reduce", da_full, ld_da, dpool, y, cf["scale"], cf["shift"], cf["mean"], cf["invstd"],
                       pl.part, N, h, w, c, code, S)
            L.call("fea_bn_bwd_finalize", pl.part, nfin, N * h * w, P[bn + ".weight"], cf["invstd"],
                   g[bn + ".weight"], g[bn + ".bias"], pl.bncoef, c, S)
            L.call("fea_bn_relu_bwd_apply", da_full, ld_da, dpool, y, cf["scale"], cf["shift"], cf["mean"], cf["invstd"],
                   pl.bncoef, dY, pl.part, N, h, w, c, code, S)
            L.call("fea_reduce_partials", pl.part, nblk, c, 0, g[name + ".bias"], S)

        def block_bwd(blk, da_full, ld_da, dpool, inp, ld_in, cin, d_in, ld_din, din_stats=None, reduced_tiles=0, pre_applied=False):
            """pre_applied: X.3's BatchNorm backward has already been applied, ring["dY"] holds its conv-output gradient."""
            h, w = pl.dims[LEVEL[blk]]
            c = CH[blk]
            n3, n0 = f"{blk}.3", f"{blk}.0"
            if not pre_applied:
                bn_bwd(n3, da_full, ld_da, dpool, h, w, c, reduced_tiles=reduced_tiles)
            dy3 = ring["dY"]

            def wg3():
                # after the dgrad that shares its input: the weight gradient then runs beside
                # the HBM-bound BatchNorm backward of X.0 instead of beside an MFMA-bound dgrad
                wgrad("fea_conv3x3_wgrad", (dy3, c, pl.a0[blk], c, g[n3 + ".weight"], pl.ws, pl.ws_bytes, N, h, w, c, c, code),
                      (n3 + ".weight", n3 + ".bias", f"{blk}.4.weight", f"{blk}.4.bias"))

            fuse_first = fuse and blk == "enc1" and self.first_direct and self.fuse_first_wgrad
            if fuse:
                cf0 = pl.coef[n0]
                rows_p = pm[n3][1]
                L.call("fea_conv3x3p_dgrad_bnred" if rows_p else "fea_conv3x3_dgrad_bnred", ring["dY"], c,
                       self._w_dgrad(n3, frag=rows_p > 0), pl.gmid, c, pl.y[n0], c,
                       cf0["scale"], cf0["shift"], cf0["mean"], cf0["invstd"], pl.part, N, h, w, c, c, code, S)
                wg3()
                if fuse_first:
                    # enc1.0 has no data gradient: its BatchNorm backward apply runs inside its weight gradient's loader
                    # (the conv-output gradient, 2 GB at batch 64 x 512^2, is neither written nor read back)
                    bn1 = f"{blk}.1"
                    L.call("fea_bn_bwd_finalize", pl.part, rows_p or L.call("fea_conv_num_tiles", N, h, w), N * h * w, P[bn1 + ".weight"],
                           cf0["invstd"], g[bn1 + ".weight"], g[bn1 + ".bias"], pl.bncoef, c, S)
                    rows1 = L.call("fea_first_wgrad_bnbwd_rows", N, h, w, c)

                    def fused_first(_a, stream):
                        # (one unit for the weight-gradient stream: the workspace and `part` are shared with the other weight gradients)
                        L.call("fea_conv3x3_first_wgrad_bnbwd", pl.gmid, c, pl.y[n0], c, cf0["scale"], cf0["shift"], cf0["mean"],
                               cf0["invstd"], pl.bncoef, pl.x_in, g[n0 + ".weight"], pl.part, pl.ws, pl.ws_bytes,
                               N, self.in_channels, h, w, c, code, stream)
                        L.call("fea_reduce_partials", pl.part, rows1, c, 0, g[n0 + ".bias"], stream)
                    wgrad(fused_first, (None,), (n0 + ".weight", n0 + ".bias", f"{blk}.1.weight", f"{blk}.1.bias"), reads_dY=False)
                    return
                bn_bwd(n0, pl.gmid, c, None, h, w, c, reduced_tiles=rows_p or L.call("fea_conv_num_tiles", N, h, w))
            else:
                L.call("fea_conv3x3", ring["dY"], c, self._w_dgrad(n3), None, pl.gmid, c, None, N, h, w, c, c, code, S)
                wg3()
                bn_bwd(n0, pl.gmid, c, None, h, w, c)
            names0 = (n0 + ".weight", n0 + ".bias", f"{blk}.1.weight", f"{blk}.1.bias")
            if d_in is not None:             # dgrad first, see wg3()
                rows_p = pm[n0][1]
                L.call("fea_conv3x3p" if rows_p else "fea_conv3x3", ring["dY"], c, self._w_dgrad(n0, frag=rows_p > 0), None,
                       d_in, ld_din, din_stats, N, h, w, c, cin, code, S)
            if blk == "enc1" and self.first_direct:
                wgrad("fea_conv3x3_first_wgrad", (ring["dY"], c, pl.x_in, g[n0 + ".weight"], pl.ws, pl.ws_bytes,
                                                  N, self.in_channels, h, w, c, code), names0)
            elif blk == "enc1":
                wgrad("fea_conv1x1_im2col_wgrad", (ring["dY"], c, pl.xcol, self.kpad, g[n0 + ".weight"], pl.ws, pl.ws_bytes,
                                                   N, h, w, self.kpad, self.in_channels, c, code), names0)
            else:
                wgrad("fea_conv3x3_wgrad", (ring["dY"], c, inp, ld_in, g[n0 + ".weight"], pl.ws, pl.ws_bytes, N, h, w, cin, c, code), names0)

        # head
        nblk = L.call("fea_head_bwd_blocks", N, H, W)
        if dl is None:
            tg = target.contiguous()
            if tg.dtype != torch.uint8 or tuple(tg.shape) != (N, self.num_classes, H, W):
                raise L.FeaError(f"target must be uint8 [{N},{self.num_classes},{H},{W}]")
            if getattr(pl, "loss_sums", None) is None:
                pl.loss_sums = torch.empty((self.num_classes, 4), dtype=torch.float64, device=self.device)
                pl.loss_part = torch.empty(nblk * self.num_classes * 4, dtype=torch.float32, device=self.device)
                pl.loss = torch.empty(1, dtype=torch.float32, device=self.device)
            head_bnbwd = False
            if getattr(pl, "head_bn", False):
                cfh = pl.coef["dec1.3"]
                L.call("fea_head_loss_sums_bn", pl.y["dec1.3"], 64, cfh["scale"], cfh["shift"], P["out.weight"], P["out.bias"], tg, None,
                       pl.loss_sums, pl.loss_part, N, H, W, 64, self.num_classes, code, S)
                if self.fuse_head_bnbwd:
                    # dec1.3's BatchNorm backward inside the head's backward too: da is never stored (two more passes over y
                    # instead of store da / reduce(da, y) / apply(da, y))
                    head_bnbwd = True
                    L.call("fea_head_loss_bwd_bnred", pl.y["dec1.3"], 64, cfh["scale"], cfh["shift"], cfh["mean"], cfh["invstd"],
                           P["out.weight"], P["out.bias"], tg, pl.loss_sums, pl.loss, pl.head_part, pl.part, 1.0,
                           N, H, W, 64, self.num_classes, code, S)
                else:
                    L.call("fea_head_loss_bwd_bn", pl.y["dec1.3"], 64, cfh["scale"], cfh["shift"], P["out.weight"], P["out.bias"], tg,
                           pl.loss_sums, pl.loss, pl.G[0], 64, pl.head_part, 1.0, N, H, W, 64, self.num_classes, code, S)
            else:
                L.call("fea_head_loss_sums", pl.a3["dec1"], 64, P["out.weight"], P["out.bias"], tg, None, pl.loss_sums, pl.loss_part,
                       N, H, W, 64, self.num_classes, code, S)
                L.call("fea_head_loss_bwd", pl.a3["dec1"], 64, P["out.weight"], P["out.bias"], tg, pl.loss_sums, pl.loss,
                       pl.G[0], 64, pl.head_part, 1.0, N, H, W, 64, self.num_classes, code, S)
        else:
            if getattr(pl, "head_bn", False):
                # the forward left dec1.3's activated tensor to the fused head + loss kernels, but the caller brings its own dlogits
                cfh = pl.coef["dec1.3"]
                L.call("fea_bn_relu_pool", pl.y["dec1.3"], cfh["scale"], cfh["shift"], pl.a3["dec1"], 64, None, N, H, W, 64, code, S)
                pl.head_bn = False
            L.call("fea_head_bwd", dl, pl.a3["dec1"], 64, P["out.weight"], pl.G[0], 64, pl.head_part, N, H, W, 64, self.num_classes, code, S)
        L.call("fea_head_bwd_finalize", pl.head_part, nblk, g["out.weight"], g["out.bias"], 64, self.num_classes, S)
        done("out.weight", "out.bias")
        if dl is None and head_bnbwd:
            cfh = pl.coef["dec1.3"]
            L.call("fea_bn_bwd_finalize", pl.part, nblk, N * H * W, P["dec1.4.weight"], cfh["invstd"],
                   g["dec1.4.weight"], g["dec1.4.bias"], pl.bncoef, 64, S)
            dY1 = next_dY()
            L.call("fea_head_loss_bwd_bnapply", pl.y["dec1.3"], 64, cfh["scale"], cfh["shift"], cfh["mean"], cfh["invstd"], pl.bncoef,
                   P["out.weight"], P["out.bias"], tg, pl.loss_sums, dY1, 64, pl.part, 1.0, N, H, W, 64, self.num_classes, code, S)
            L.call("fea_reduce_partials", pl.part, nblk, 64, 0, g["dec1.3.bias"], S)
        # decoder
        pre = 0          # > 0: pl.part holds the BN-backward partial sums of the next block's X.3 layer
        for k in (1, 2, 3, 4):
            blk = f"dec{k}"
            c = CH[blk]
            # the dgrad that writes dcat[k] also emits per-tile channel sums of it: channels
            # [0,c) summed over pixels = the transposed conv's bias gradient
            block_bwd(blk, pl.G[0], c, None, pl.cat[k], 2 * c, 2 * c, pl.dcat[k], 2 * c, din_stats=pl.stats, reduced_tiles=pre,
                      pre_applied=(k == 1 and dl is None and head_bnbwd))
            hl, wl = pl.dims[k]
            src = pl.a3[f"dec{k + 1}"] if k < 4 else pl.a3["bottleneck"]
            L.call("fea_reduce_partials_n", pl.stats, pm[f"{blk}.0"][1] or L.call("fea_conv_num_tiles", N, 2 * hl, 2 * wl), 2 * c, c, 0,
                   g[f"upconv{k}.bias"], S)
            wgrad("fea_upconv2x2_wgrad", (src, 2 * c, pl.dcat[k], 2 * c, g[f"upconv{k}.weight"], pl.ws, pl.ws_bytes,
                                          N, hl, wl, 2 * c, c, code), (f"upconv{k}.weight", f"upconv{k}.bias"), reads_dY=False)
            if fuse:
                nxt = f"dec{k + 1}.3" if k < 4 else "bottleneck.3"
                cfn = pl.coef[nxt]
                L.call("fea_upconv2x2_bwd_data_bnred", pl.dcat[k], 2 * c, self._w_up(f"upconv{k}", 1), pl.G[0], 2 * c,
                       pl.y[nxt], 2 * c, cfn["scale"], cfn["shift"], cfn["mean"], cfn["invstd"], pl.part,
                       N, hl, wl, 2 * c, c, code, S)
                pre = L.call("fea_conv_num_tiles", N, hl, wl)
            else:
                L.call("fea_upconv2x2_bwd_data", pl.dcat[k], 2 * c, self._w_up(f"upconv{k}", 1), pl.G[0], 2 * c,
                       N, hl, wl, 2 * c, c, code, S)
        # bottleneck and encoder
        block_bwd("bottleneck", pl.G[0], 1024, None, pl.pool[4], 512, 512, pl.G[1], 512, reduced_tiles=pre)
        cur = 1
        for k in (4, 3, 2, 1):
            blk = ENC[k - 1]
            c = CH[blk]
            if k > 1:
                cprev = CH[ENC[k - 2]]
                block_bwd(blk, off_ptr(pl.dcat[k], c), 2 * c, pl.G[cur], pl.pool[k - 1], cprev, cprev, pl.G[1 - cur], cprev)
            else:
                block_bwd(blk, off_ptr(pl.dcat[k], c), 2 * c, pl.G[cur], pl.xcol, self.kpad, self.kpad, None, 0)
            cur = 1 - cur
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)
        return g
