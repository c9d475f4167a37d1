"""Thin torch-tensor wrappers over the C ABI (include/pointsam_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every computation is a kernel of
libpointsam_hip.so launched on ``torch.cuda.current_stream()``.  Inputs must be CUDA(HIP) fp32 / int64
tensors; there is no CPU path.
"""
import ctypes
import torch

from . import _lib
from ._lib import check

ACT_NONE, ACT_GELU, ACT_RELU, ACT_SWIGLU = 0, 1, 2, 3

# Measurement hook (bench.py): when GEMM_PROFILE is a list, every GEMM_PROFILE_EVERY-th GEMM launch is bracketed by two
# HIP events on the launch stream and (start, end, flops, M, N, K) is appended.  Sampling keeps the perturbation of the
# timed region small (an event pair costs ~20 us of host time and the host issues ~400 launches per 14 ms step: bracketing every
# launch cost 17 %, every 7th still 7 %); bench.py samples every 3rd launch of the LAST step of the timed region only.
GEMM_PROFILE = None
GEMM_PROFILE_EVERY = 29
# Streams whose kernels must not overlap a sampled launch (BatchPipeline with several batches in flight): the sampled launch
# waits for what they have enqueued so far and they wait for its end, so the events bracket the kernel running ALONE.
GEMM_PROFILE_OTHERS = ()
GEMM_PROFILE_AFTER = 0     # launches (since the counter was reset) to skip before sampling starts
_gemm_counter = 0


def _sample_now():
    return GEMM_PROFILE is not None and _gemm_counter > GEMM_PROFILE_AFTER and _gemm_counter % GEMM_PROFILE_EVERY == 0


def _sampled_launch(launch, record):
    """Brackets one launch with HIP events on the launch stream (exclusive of GEMM_PROFILE_OTHERS) and records them."""
    cur = torch.cuda.current_stream()
    for o in GEMM_PROFILE_OTHERS:
        cur.wait_stream(o)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    launch()
    e.record()
    for o in GEMM_PROFILE_OTHERS:
        o.wait_event(e)
    record(s, e)


# GEMM arithmetic: "f32" = v_mfma_f32_32x32x2_f32 everywhere (exact fp32 products); "bf16x6" = large GEMMs on the bf16
# matrix pipe with the exact 3-way operand split (fp32-accurate, see csrc/gemm_split.hip), small ones stay on "f32".
# The mode is a CONTEXT variable (per thread / per asyncio task), entered by every model method for its own precision: two models of different
# precision -- or a threaded server -- do not see each other's setting.  `ops.GEMM_MODE` still reads the current value (module __getattr__).
import contextvars
_GEMM_MODE_VAR = contextvars.ContextVar("point_sam_amd_gemm_mode", default="f32")


def current_gemm_mode() -> str:
    return _GEMM_MODE_VAR.get()


def __getattr__(name):
    if name == "GEMM_MODE":
        return _GEMM_MODE_VAR.get()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


class _OpsModule(__import__("types").ModuleType):
    """`ops.GEMM_MODE = ...` (the pre-round-4 interface) would create a real module attribute that shadows __getattr__ for good: every later read of
    ops.GEMM_MODE would then return that stale value while the launches follow the context variable.  Assignment is refused; use `with ops.gemm_mode(m):`."""

    def __setattr__(self, name, value):
        if name == "GEMM_MODE":
            raise AttributeError("ops.GEMM_MODE is read-only (it reflects a context variable): use `with ops.gemm_mode(mode):`")
        super().__setattr__(name, value)


import sys as _sys
_sys.modules[__name__].__class__ = _OpsModule


# Key split of the fp16-pipe attention for single-cloud shapes (csrc/attention.hip, psam_attention_f16x3_ex2): 4 = up to four workgroups per
# (query block, head), combined in the kernel -- the latency of ONE stream of work; 1 = off, what the multi-stream pipelines use (their other streams
# fill the idle CUs and the split's extra work costs throughput: 133 -> 129 sessions/s at cfg #5, while the encoder latency drops 8.7 -> 8.1 ms).
_ATTN_KEYSPLIT_VAR = contextvars.ContextVar("point_sam_amd_attn_keysplit", default=4)


def current_attention_keysplit() -> int:
    return _ATTN_KEYSPLIT_VAR.get()


class attention_keysplit:
    """Context manager: cap of the attention's key split for the enclosed launches (1 = never split)."""

    def __init__(self, n: int):
        self.n = max(1, int(n))

    def __enter__(self):
        self.token = _ATTN_KEYSPLIT_VAR.set(self.n)

    def __exit__(self, *exc):
        _ATTN_KEYSPLIT_VAR.reset(self.token)


# Arrival-counter blocks (include/pointsam_hip.h, PSAM_COUNTER_BYTES): kernels with an in-kernel fix-up (split-K GEMM, key-split attention, skinny Linear +
# LayerNorm) count their workgroups in through device memory the CALLER owns -- the library allocates nothing and keeps no per-stream state (round 6).
# This host keeps one zeroed block per (device, stream) for eager launches (launches on one stream are ordered, so they may share it) and lets a scope
# bring its own: every GraphPipeline slot captures its graphs on a block of its own (`use_counters`), so any two graphs may be in flight together and a
# graph may replay on any stream.  A capture outside such a scope on a stream that never launched eagerly gets a fresh block inside the capture
# (zeroed by a node of that graph on every replay).
COUNTER_BYTES = 65536
_COUNTERS = {}
_COUNTERS_VAR = contextvars.ContextVar("point_sam_amd_counters", default=None)


def new_counters(device) -> torch.Tensor:
    return torch.zeros(COUNTER_BYTES // 4, dtype=torch.int32, device=device)


def arrival_counters(device) -> torch.Tensor:
    """The arrival-counter block of the enclosing `use_counters` scope, else the current stream's."""
    t = _COUNTERS_VAR.get()
    if t is not None:
        if t.device != torch.device(device) and not (t.device.type == "cuda" and torch.device(device).index is None):
            raise _lib.PointSamHipError(f"use_counters: the block lives on {t.device}, the launch is on {device}")
        return t
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    t = _COUNTERS.get(key)
    if t is None:
        t = new_counters(dev)
        if not torch.cuda.is_current_stream_capturing():
            _COUNTERS[key] = t
    return t


class use_counters:
    """Context manager: the enclosed launches count in through `block` (a new_counters() tensor) instead of the current stream's block."""

    def __init__(self, block: torch.Tensor):
        if block.dtype != torch.int32 or block.numel() * 4 < COUNTER_BYTES or not block.is_cuda or not block.is_contiguous():
            raise ValueError("use_counters: need a contiguous int32 device tensor of COUNTER_BYTES bytes (ops.new_counters)")
        self.block = block

    def __enter__(self):
        self.token = _COUNTERS_VAR.set(self.block)
        return self.block

    def __exit__(self, *exc):
        _COUNTERS_VAR.reset(self.token)


SPLIT_MIN_M, SPLIT_MIN_N, SPLIT_MIN_K = 256, 128, 128
SKINNY_MAX_M = 64      # up to this many rows nn.Linear runs on the skinny kernel (exact fp32 products, like "f32")
GEMM_MODES = ("f32", "bf16x6", "f16x3")
# "f16x3" = large 2-D GEMMs against a prepared static weight (F16Weight) on the fp16 matrix pipe: power-of-two row scales + 2-way
# fp16 split, 3 partial products (fp32-grade: same measured error vs fp64 as the f32 kernel, csrc/gemm_f16x3p.hip); batched and
# small ones stay on "f32".


class gemm_mode:
    """Context manager selecting the GEMM arithmetic for the enclosed launches."""

    def __init__(self, mode):
        if mode not in GEMM_MODES:
            raise ValueError(f"unknown GEMM mode {mode!r}; expected one of {GEMM_MODES}")
        self.mode = mode

    def __enter__(self):
        self.token = _GEMM_MODE_VAR.set(self.mode)

    def __exit__(self, *exc):
        _GEMM_MODE_VAR.reset(self.token)


def _gemm_call(fn_args, flops, M, N, K, what):
    global _gemm_counter
    L = _lib.load()
    split = current_gemm_mode() == "bf16x6" and M >= SPLIT_MIN_M and N >= SPLIT_MIN_N and K >= SPLIT_MIN_K
    fn = L.psam_gemm_bf16x6 if split else L.psam_gemm_f32
    _gemm_counter += 1
    if not _sample_now():
        check(fn(*fn_args), what)
        return
    _sampled_launch(lambda: check(fn(*fn_args), what), lambda s, e: GEMM_PROFILE.append((s, e, flops, M, N, K, "bf16x6" if split else "f32")))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise _lib.PointSamHipError(f"{name} must live on the GPU: the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


# ------------------------------------------------------------------------------------------ tokenizer
def fps(xyz: torch.Tensor, num_samples: int):
    """[B,N,3] -> (fps_idx [B,G] int64, centers [B,G,3]).  common.py:91-92."""
    _chk(xyz, name="xyz")
    B, N, _ = xyz.shape
    L = _lib.load()
    nbytes = L.psam_fps_workspace_bytes(B, N, num_samples)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=xyz.device)
    idx = torch.empty(B, num_samples, dtype=torch.int64, device=xyz.device)
    centers = torch.empty(B, num_samples, 3, dtype=torch.float32, device=xyz.device)
    check(L.psam_fps(xyz.data_ptr(), B, N, num_samples, idx.data_ptr(), centers.data_ptr(), ws.data_ptr(), nbytes, _stream()), "psam_fps")
    return idx, centers


def knn(centers: torch.Tensor, xyz: torch.Tensor, k: int) -> torch.Tensor:
    """[B,G,3], [B,N,3] -> knn_idx [B,G,K] int64 ascending by (d2, index).  common.py:27-56,97."""
    _chk(centers, name="centers"); _chk(xyz, name="xyz")
    B, G, _ = centers.shape
    N = xyz.shape[1]
    out = torch.empty(B, G, k, dtype=torch.int64, device=xyz.device)
    check(_lib.load().psam_knn(centers.data_ptr(), xyz.data_ptr(), B, G, N, k, out.data_ptr(), _stream()), "psam_knn")
    return out


def three_nn(xyz: torch.Tensor, centers: torch.Tensor, eps: float = 1e-8):
    """-> (idx3 [B,N,3] int64, w3 [B,N,3]).  common.py:238-255."""
    _chk(xyz, name="xyz"); _chk(centers, name="centers")
    B, N, _ = xyz.shape
    G = centers.shape[1]
    idx = torch.empty(B, N, 3, dtype=torch.int64, device=xyz.device)
    w = torch.empty(B, N, 3, dtype=torch.float32, device=xyz.device)
    check(_lib.load().psam_three_nn(xyz.data_ptr(), centers.data_ptr(), B, N, G, eps, idx.data_ptr(), w.data_ptr(), _stream()), "psam_three_nn")
    return idx, w


def group_gather(xyz, feats, centers, knn_idx, radius=None, width=None):
    """feats [B*rep,N,C] -> [B*rep,G,K,3+C].  common.py:99-120 / 126-187.  width (>= 3 + C): rows zero-padded to that many columns."""
    _chk(xyz); _chk(feats); _chk(centers); _chk(knn_idx, torch.int64)
    B, N, _ = xyz.shape
    rep = feats.shape[0] // B
    G, K = knn_idx.shape[1:]
    C = feats.shape[-1]
    width = width or 3 + C
    out = torch.empty(B * rep, G, K, width, dtype=torch.float32, device=xyz.device)
    check(_lib.load().psam_group_gather_ld(xyz.data_ptr(), feats.data_ptr(), centers.data_ptr(), knn_idx.data_ptr(), B, rep, N, G, K, C,
                                           float(radius or 0.0), out.data_ptr(), width, _stream()), "psam_group_gather")
    return out


def patch_l1(xyz, feats, centers, knn_idx, W, bias, lnw, lnb, eps, out=None, radius=None, center_idx=None, scale_out=None):
    """Fused gather + Linear(Cin,128) + LayerNorm + GELU -> [B*rep*G*K, 128].  common.py:486-489.
    center_idx [B,G] (FPS indices): centralize_features (Cin = 3 + 2C, common.py:116-118); scale_out [rows]: the rows leave g8-packed
    for the f16x3 GEMM (linear(..., x_scale=scale_out, x_packed=True))."""
    _chk(xyz); _chk(feats); _chk(centers); _chk(knn_idx, torch.int64); _chk(W)
    B, N, _ = xyz.shape
    rep = feats.shape[0] // B
    G, K = knn_idx.shape[1:]
    C = feats.shape[-1]
    cin = 3 + C * (2 if center_idx is not None else 1)
    assert W.shape == (128, cin), (W.shape, cin)
    if center_idx is not None:
        _chk(center_idx, torch.int64, "center_idx")
    rows = B * rep * G * K
    if out is None:
        out = torch.empty(rows, 128, dtype=torch.float32, device=xyz.device)
    check(_lib.load().psam_patch_l1_ex(xyz.data_ptr(), feats.data_ptr(), centers.data_ptr(), knn_idx.data_ptr(), _p(center_idx), W.data_ptr(),
                                       bias.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), eps, B, rep, N, G, K, C, float(radius or 0.0), out.data_ptr(),
                                       _p(scale_out), _stream()), "psam_patch_l1")
    return out


def nn_group_feats(xyz, centers, nn_idx, feats=None, logits=None, width=8):
    """Voronoi variant: per-point rows relative to the nearest centre, zero-padded to `width` columns (csrc/rowops.hip nn_group_feats_kernel).
    feats [B,N,C]: NNGrouper.forward (common.py:203-211) -> [B*N, width]; logits [Z,N]: MaskEncoderNN's input (prompt_encoder.py:281-287) -> [Z*N, width]."""
    _chk(xyz); _chk(centers); _chk(nn_idx, torch.int64)
    B, N, _ = xyz.shape
    G = centers.shape[1]
    if logits is None:
        _chk(feats)
        C, rep, mode = feats.shape[-1], 1, 0
    else:
        _chk(logits)
        C, rep, mode = 0, logits.shape[0] // B, 1
    assert width % 4 == 0 and width >= (4 + C if mode == 0 else 5)
    out = torch.empty(B * rep * N, width, dtype=torch.float32, device=xyz.device)
    check(_lib.load().psam_nn_group_feats(xyz.data_ptr(), centers.data_ptr(), nn_idx.data_ptr(), _p(feats), _p(logits), B, rep, N, G, C, mode, out.data_ptr(),
                                          width, _stream()), "psam_nn_group_feats")
    return out


def scatter_amax(x, idx, out_rows, rows_per_set, set_stride, idx_rep=1, include_self=False):
    """Max-pool of the rows of x [R, C] into out [out_rows, C]: row r goes to idx[(r // rows_per_set // idx_rep), r % rows_per_set] + (r //
    rows_per_set) * set_stride (torch scatter_reduce 'amax'; see include/pointsam_hip.h psam_scatter_amax)."""
    xp, ldx = _row_view(x, "x"); _chk(idx, torch.int64)
    R, C = x.shape
    out = torch.empty(out_rows, C, dtype=torch.float32, device=x.device)
    check(_lib.load().psam_scatter_amax(xp, ldx, idx.data_ptr(), R, C, rows_per_set, set_stride, idx_rep, out.data_ptr(), out_rows, int(bool(include_self)),
                                        _stream()), "psam_scatter_amax")
    return out


def group_max(x: torch.Tensor, K: int, out=None):
    """[groups*K, C] -> [groups, C]."""
    _chk(x)
    rows, C = x.shape
    groups = rows // K
    if out is None:
        out = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    check(_lib.load().psam_group_max(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), groups, K, C, _stream()), "psam_group_max")
    return out


# ------------------------------------------------------------------------------------------ dense
def _row_view(t, name):
    """2-D fp32 view with unit inner stride -> (ptr, ld)."""
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D CUDA fp32 tensor with unit inner stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")
    return t.data_ptr(), t.stride(0)


def row_scale_f16(x, K=None, out=None):
    """Per-row power-of-two scales for the f16x3 GEMM (row maximum of x[:, :K] into [2^14, 2^15))."""
    xp, ldx = _row_view(x, "x")
    rows = x.shape[0]
    if out is None:
        out = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.load().psam_row_scale_f16(xp, ldx, rows, x.shape[1] if K is None else K, out.data_ptr(), _stream()), "psam_row_scale_f16")
    return out


def _kpad(K: int) -> int:
    return (K + 31) // 32 * 32


def pack_rows_g8(x, scale, K=None, out=None):
    """g8-packed form of the row-scaled x[:, :K] (csrc/gemm_f16x3p.hip): per 8 consecutive k [hi x8 | lo x8] fp16 in the 32-bit
    containers; K is zero-padded to the next multiple of 32 (the GEMM's slab)."""
    xp, ldx = _row_view(x, "x")
    rows = x.shape[0]
    K = x.shape[1] if K is None else K
    if out is None:
        out = torch.empty(rows, _kpad(K), dtype=torch.float32, device=x.device)
    op, ldo = _row_view(out, "out")
    check(_lib.load().psam_pack_rows_f16x2_g8(xp, ldx, scale.data_ptr(), rows, K, op, ldo, _stream()), "psam_pack_rows_f16x2_g8")
    return out


def scale_pack_rows_g8(x, K=None, out=None, scale=None):
    """(packed, scale): row scales and the g8-packed form of fp32 rows in ONE pass (K <= 6144, K % 4 == 0)."""
    xp, ldx = _row_view(x, "x")
    rows = x.shape[0]
    K = x.shape[1] if K is None else K
    if out is None:
        out = torch.empty(rows, _kpad(K), dtype=torch.float32, device=x.device)
    if scale is None:
        scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    op, ldo = _row_view(out, "out")
    check(_lib.load().psam_scale_pack_rows_g8(xp, ldx, rows, K, op, ldo, scale.data_ptr(), _stream()), "psam_scale_pack_rows_g8")
    return out, scale


class F16Weight:
    """A static nn.Linear weight [N, K] prepared ONCE for the "f16x3" GEMM (csrc/gemm_f16x3p.hip): per-row power-of-two scales and the
    g8-packed, K-padded (multiple of 32) hi|lo fp16 form.  Owned by the model that built it (no process-global cache); the fp32
    original stays available for launches below the split thresholds."""

    def __init__(self, W: torch.Tensor):
        _row_view(W, "W")
        self.fp32 = W
        self.N, self.K = W.shape
        self.Kp = _kpad(self.K)
        self._scale = self._packed = None      # packed on first use: weights that only ever run through a coarse C-ABI stage (psam_eva_block & co. hold
                                                # their own packed blob) never get this second copy

    def _materialise(self):
        if torch.cuda.is_current_stream_capturing():
            raise _lib.PointSamHipError("F16Weight: first use inside a graph capture (run one eager pass first, as GraphPipeline does)")
        self._scale = row_scale_f16(self.fp32)
        self._packed = pack_rows_g8(self.fp32, self._scale)
        assert self._packed.shape[1] == self.Kp
        torch.cuda.current_stream(self.fp32.device).synchronize()      # other streams may use it right after (the pipelines run several)

    @property
    def scale(self):
        if self._scale is None:
            self._materialise()
        return self._scale

    @property
    def packed(self):
        if self._packed is None:
            self._materialise()
        return self._packed

    @staticmethod
    def eligible(N: int, K: int) -> bool:
        return N >= SPLIT_MIN_N and K >= SPLIT_MIN_K

    @property
    def shape(self):
        return (self.N, self.K)


def _f16x3p_call(fn_args, flops, M, N, K, fuse=None):
    global _gemm_counter
    import ctypes
    L = _lib.load()
    _gemm_counter += 1
    fn_args = fn_args[:-1] + (ctypes.byref(fuse) if fuse is not None else None, fn_args[-1])
    if not _sample_now():
        check(L.psam_gemm_f16x3p_ex(*fn_args), "psam_gemm_f16x3p")
        return
    _sampled_launch(lambda: check(L.psam_gemm_f16x3p_ex(*fn_args), "psam_gemm_f16x3p"), lambda s, e: GEMM_PROFILE.append((s, e, flops, M, N, K, "f16x3")))


_SPLITK = {}


def splitk_factor(M: int, N: int, K: int, act: int) -> int:
    """Split-K factor the library suggests for a packed-operand GEMM shape (1: none); a pure function of the shape and the device."""
    key = (M, N, K, act, torch.cuda.current_device())
    ks = _SPLITK.get(key)
    if ks is None:
        ks = _SPLITK[key] = int(_lib.load().psam_gemm_f16x3p_splitk(M, N, K, act))
    return ks


def fuse_supported(M: int, N: int) -> bool:
    """Shapes for which the fused GEMM extras (packed output, LayerNorm partials, folded LayerNorm) exist."""
    return M % 256 == 0 and N % 128 == 0


def row_ln_bound(gamma, beta) -> float:
    """Bound on |LayerNorm(x) * gamma + beta| over a row of len(gamma) columns: a normalised element is at most sqrt(n - 1).
    (GELU / ReLU after it only shrink magnitudes.)  The k2 of a packed output behind a row-LayerNorm epilogue."""
    n = gamma.numel()
    return float((n - 1) ** 0.5 * gamma.abs().max().item() + beta.abs().max().item())


def fused_row_ln(N: int) -> bool:
    """Whether linear(..., row_ln=...) exists for N output columns (psam_gemm_f16x3p_fused_row_ln): 256 always, 512 with the register epilogue."""
    return bool(_lib.load().psam_gemm_f16x3p_fused_row_ln(int(N)))


def stat_segs(N: int) -> int:
    return (N // 2 + 31) // 32


def ln_stats_finalize(stats, cols, eps):
    """[M, segs, 2] per-segment (mean, centred sum of squares) -> (mean [M], rstd [M]) of a LayerNorm over `cols` columns."""
    M, segs, _ = stats.shape
    mean = torch.empty(M, dtype=torch.float32, device=stats.device)
    rstd = torch.empty(M, dtype=torch.float32, device=stats.device)
    check(_lib.load().psam_ln_stats_finalize(stats.data_ptr(), M, segs, cols, eps, mean.data_ptr(), rstd.data_ptr(), _stream()), "psam_ln_stats_finalize")
    return mean, rstd


def linear(x, W, bias=None, act=ACT_NONE, residual=None, out=None, rowbias=None, rowgroup=0, K=None, x_scale=None, x_packed=False,
           pack_out=None, stats=None, ln_fold=None, group_max_out=None, group_max_k=0, no_store=False, row_ln=None, hyper=None):
    """y = act(x @ W[:, :K]^T + bias + rowbias[row // rowgroup]) + residual.  x [M,>=K], W [N,>=K] row views, or W an F16Weight
    (prepared static weight).
    "f16x3" mode with an F16Weight and M above the threshold runs the packed-operand GEMM (csrc/gemm_f16x3p.hip): x is either already
    g8-packed by its producer (x_packed=True with x_scale: LayerNorm / scale_pack_rows_g8 output, [M, >= K padded to 32]) or is
    scaled and packed here in one extra pass.  Fused extras of that GEMM (fuse_supported shapes; include/pointsam_hip.h psam_gemm_fuse_t):
    pack_out=(scale_out [M], k1, k2) or (scale_out [M], bound [M]): `out` receives g8-packed rows + their bound-derived scales; stats=(buf [M, stat_segs(N), 2], cols):
    LayerNorm partials of the SwiGLU-gated rows; ln_fold=(mean [M], rstd [M], c [N]): LayerNorm of x folded into the GEMM;
    group_max_out [M / group_max_k, N] (group_max_k 32 | 64): per-column max over consecutive row groups of the output, no_store: the
    [M, N] output itself is not written (out may then be None).  N == 256 only: row_ln=(gamma, beta, eps): LayerNorm of every output
    row before the activation (with pack_out pass k1 = 0, k2 = row_ln_bound(gamma, beta));
    hyper=(hyper [Z, C, 256], masks [Z, C, rows_per_z], rows_per_z): masks[z, c, n] = <hyper[z, c], out[z * rows_per_z + n]>."""
    fw = None
    if isinstance(W, F16Weight):
        fw, W = W, W.fp32
    xp, ldx = _row_view(x, "x")
    wp, ldw = _row_view(W, "W")
    M = x.shape[0]
    N = W.shape[0]
    if K is None:
        K = W.shape[1]
        assert x_packed or x.shape[1] == K, (x.shape, W.shape)
    elif fw is not None and K != fw.K:
        fw = None
    fused = pack_out is not None or stats is not None or ln_fold is not None or group_max_out is not None or row_ln is not None or hyper is not None
    if no_store and (group_max_out is not None or hyper is not None) and out is None:
        op, ldo = (group_max_out if group_max_out is not None else hyper[1]).data_ptr(), N       # never dereferenced
    else:
        if out is None:
            out = torch.empty(M, N // 2 if act == ACT_SWIGLU else N, dtype=torch.float32, device=x.device)
        op, ldo = _row_view(out, "out")
    rp, ldr = (0, 0) if residual is None else _row_view(residual, "residual")
    rbp, ldrb = (0, 0) if rowbias is None else _row_view(rowbias, "rowbias")
    if current_gemm_mode() == "f16x3" and fw is not None and M >= SPLIT_MIN_M:
        if x_packed:
            if x_scale is None or x.shape[1] < fw.Kp or (ldx & 7) or (xp & 31):
                raise ValueError("x_packed needs x_scale and g8-packed rows padded to a multiple of 32 columns, 32-byte aligned")
            xa, sa = x, x_scale
        else:
            xa, sa = scale_pack_rows_g8(x, K)
        fuse = None
        hyper_parts = None
        if not fused and rowbias is None and act != ACT_SWIGLU and N % 4 == 0 and ldo % 4 == 0 and ldr % 4 == 0:
            ks = splitk_factor(M, N, fw.Kp, act)
            if ks > 1:      # few tiles, long K: partial planes + a fixed-order reduction (psam_gemm_fuse_t.splitk)
                ws = torch.empty(ks, M * N, dtype=torch.float32, device=x.device)
                fuse = _lib.GemmFuse()
                fuse.splitk_ws, fuse.splitk_plane, fuse.splitk = ws.data_ptr(), M * N, ks
                fuse.counters = arrival_counters(x.device).data_ptr()      # in-kernel fix-up instead of a reduction launch
        if fused:
            fuse = _lib.GemmFuse()
            fuse.no_store = int(bool(no_store))
            if group_max_out is not None:
                fuse.gmax_out, fuse.gmax_ld, fuse.gmax_k = group_max_out.data_ptr(), group_max_out.stride(0), int(group_max_k)
            if row_ln is not None:
                fuse.row_ln_g, fuse.row_ln_b, fuse.row_ln_eps = row_ln[0].data_ptr(), row_ln[1].data_ptr(), float(row_ln[2])
            if hyper is not None:
                hy, mk, rpz = hyper
                if not (hy.is_contiguous() and mk.is_contiguous() and hy.shape[-1] == N and mk.shape[:2] == hy.shape[:2] and mk.shape[2] == rpz):
                    raise ValueError("hyper [Z, C, N] / masks [Z, C, rows_per_z] must be contiguous and consistent")
                fuse.hyper, fuse.masks, fuse.hyper_c, fuse.hyper_rows = hy.data_ptr(), mk.data_ptr(), int(hy.shape[1]), int(rpz)
                planes = _lib.load().psam_gemm_f16x3p_hyper_planes(N, int(row_ln is not None))
                if planes > 1:      # partial products per 64-column wave tile: N / 64 planes, added below in a fixed order
                    hyper_parts = torch.empty(planes, mk.numel(), dtype=torch.float32, device=mk.device)
                    fuse.masks, fuse.hyper_pstride = hyper_parts.data_ptr(), mk.numel()
            if pack_out is not None and len(pack_out) == 2:      # (scale_out, per-row bound of the output): see psam_gemm_fuse_t.out_bound
                fuse.out_scale, fuse.out_bound, fuse.pack_out = pack_out[0].data_ptr(), pack_out[1].data_ptr(), 1
            elif pack_out is not None:
                fuse.out_scale, fuse.out_k1, fuse.out_k2, fuse.pack_out = pack_out[0].data_ptr(), float(pack_out[1]), float(pack_out[2]), 1
            if stats is not None:
                fuse.stats, fuse.stat_cols = stats[0].data_ptr(), int(stats[1])
            if ln_fold is not None:
                fuse.ln_mean, fuse.ln_rstd, fuse.ln_c = ln_fold[0].data_ptr(), ln_fold[1].data_ptr(), ln_fold[2].data_ptr()
        _f16x3p_call((xa.data_ptr(), xa.stride(0), sa.data_ptr(), fw.packed.data_ptr(), fw.packed.stride(0), fw.scale.data_ptr(), op, ldo, _p(bias),
                      rp, ldr, rbp, ldrb, rowgroup, M, N, fw.Kp, 1.0, act, _stream()), 2.0 * M * N * K, M, N, K, fuse)
        if hyper_parts is not None:
            check(_lib.load().psam_sum_planes(hyper_parts.data_ptr(), hyper_parts.shape[0], hyper_parts.shape[1], hyper_parts.shape[1],
                                              hyper[1].data_ptr(), _stream()), "psam_sum_planes")
        return out
    if fused:
        raise ValueError("fused GEMM extras exist only on the f16x3 packed-operand path")
    if x_packed:
        raise ValueError("x_packed activations can only feed an f16x3 GEMM with a prepared F16Weight (M above the split threshold)")
    if (M <= SKINNY_MAX_M and K % 16 == 0 and rowbias is None and act in (ACT_NONE, ACT_GELU, ACT_RELU) and (ldx | ldw) % 4 == 0
            and (xp | wp) % 16 == 0):
        # a handful of rows (the decoder's output tokens): N / 16 workgroups, whole-K load rounds (csrc/gemm.hip linear_skinny_kernel)
        check(_lib.load().psam_linear_skinny(xp, ldx, wp, ldw, _p(bias), rp, ldr, op, ldo, M, N, K, act, _stream()), "psam_linear_skinny")
        return out
    _gemm_call((xp, ldx, 0, 0, wp, ldw, 0, 0, op, ldo, 0, 0, _p(bias), rp, ldr, 0, 0, rbp, ldrb, rowgroup, M, N, K, 1, 1, 1.0, act, _stream()),
               2.0 * M * N * K, M, N, K, "psam_gemm_f32")
    return out


def gemm_batched(A, W, out, M, N, K, lda, ldw, ldc, sA, sW, sC, batch, alpha=1.0):
    """out[z] = alpha * A[z] @ W[z]^T (raw strided-batched form; tensors give base pointers only)."""
    _gemm_call((A.data_ptr(), lda, sA, 0, W.data_ptr(), ldw, sW, 0, out.data_ptr(), ldc, sC, 0, 0, 0, 0, 0, 0, 0, 0, 0, M, N, K, batch, 1, alpha,
                ACT_NONE, _stream()), 2.0 * M * N * K * batch, M, N, K, "psam_gemm_f32(batched)")
    return out


def layernorm_can_pack(cols: int) -> bool:
    return 256 <= cols <= 4096


def packed_cols(cols: int) -> int:
    """Columns of the buffer that receives a g8-packed row of `cols` values (zero-padded to the GEMM's 32-k slab)."""
    return _kpad(cols)


def layernorm(x, w, b, eps, act=ACT_NONE, residual=None, out=None, scale_out=None, pack=False, bound_out=None):
    """y = act(LN(x + residual)) over the last dim of a 2-D row view.  scale_out ([rows] fp32, optional) receives the f16x3
    row scales of y (what row_scale_f16(y) would compute), for the GEMM that consumes y.  pack=True: out receives the g8-packed
    form of the scaled rows instead of fp32 (linear(..., x_scale=scale_out, x_packed=True)); out needs packed_cols(cols) columns
    of room per row for the zero padding (or exactly cols when cols % 8 == 0 and the padding already holds zeros).
    bound_out=(buf [rows], c2, c1, c0) (with pack): buf[r] = c2 t^2 + c1 t + c0, t = ||y[r]||_2 -- the per-row output bound of the GEMM that
    consumes y (linear(..., pack_out=(scale_out, buf)))."""
    xp, ldx = _row_view(x, "x")
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, _kpad(cols) if pack else cols, dtype=torch.float32, device=x.device)
    op, ldo = _row_view(out, "out")
    rp, ldr = (0, 0) if residual is None else _row_view(residual, "residual")
    if bound_out is not None:
        bb, c2, c1, c0 = bound_out
        check(_lib.load().psam_layernorm_ex2(xp, ldx, rp, ldr, w.data_ptr(), b.data_ptr(), op, ldo, rows, cols, eps, act, _p(scale_out), 1 if pack else 0,
                                             bb.data_ptr(), float(c2), float(c1), float(c0), _stream()), "psam_layernorm")
        return out
    check(_lib.load().psam_layernorm_ex(xp, ldx, rp, ldr, w.data_ptr(), b.data_ptr(), op, ldo, rows, cols, eps, act, _p(scale_out), 1 if pack else 0,
                                        _stream()), "psam_layernorm")
    return out


def swiglu_ln(gx, xoff, H, w, b, eps, out):
    gp, ldg = _row_view(gx, "gx")
    op, ldo = _row_view(out, "out")
    check(_lib.load().psam_swiglu_ln(gp, ldg, xoff, w.data_ptr(), b.data_ptr(), op, ldo, gx.shape[0], H, eps, _stream()), "psam_swiglu_ln")
    return out


def _f16x3_head_dim(hd: int) -> bool:
    """Head dims of psam_attention_f16x3: 64, or a multiple of 8 in (64, 128] (zero-padded to 128 inside the kernel: the giant encoder's 88)."""
    return hd == 64 or (64 < hd <= 128 and hd % 8 == 0)


def attention_can_pack(hd: int) -> bool:
    return current_gemm_mode() == "f16x3" and _f16x3_head_dim(hd)


def attention(q, k, v, out, B, H, Lq, Lk, hd, scale, pack=None):
    """q/k/v/out: 2-D row views [B*L, >=H*hd] (may be column slices of a fused qkv buffer).
    pack=(a_scale [B*Lk], k1, k2, o_scale [B*Lq]) ("f16x3", hd 64 / 128, self-attention): out receives the g8-packed rows for the output
    projection and o_scale their (bound-derived, per-cloud) scales -- linear(out, W, x_scale=o_scale, x_packed=True)."""
    qp, ldq = _row_view(q, "q"); kp, ldk = _row_view(k, "k"); vp, ldv = _row_view(v, "v"); op, ldo = _row_view(out, "out")
    L = _lib.load()
    args = (qp, ldq, Lq * ldq, kp, ldk, Lk * ldk, vp, ldv, Lk * ldv, op, ldo, Lq * ldo, B, H, Lq, Lk, hd, scale)
    if pack is not None and not attention_can_pack(hd):
        raise ValueError("packed attention output needs the f16x3 kernel (head dim 64, or a multiple of 8 in (64, 128])")
    if pack is not None or (current_gemm_mode() == "f16x3" and _f16x3_head_dim(hd)):
        a_scale, k1, k2, o_scale = pack if pack is not None else (None, 0.0, 0.0, None)
        ks = current_attention_keysplit()
        nb = int(L.psam_attention_f16x3_keysplit_ws_bytes(B, H, Lq, Lk, hd, ks)) if ks > 1 else 0
        ks_ws = torch.empty(nb, dtype=torch.uint8, device=out.device) if nb else None      # single-cloud shape: key split with in-kernel combine (scratch and
        check(L.psam_attention_f16x3_ex2(*args, _p(a_scale), float(k1), float(k2), _p(o_scale), ks if nb else 1, _p(ks_ws), nb,      # counter block are the caller's)
                                         arrival_counters(out.device).data_ptr() if nb else None, _stream()), "psam_attention")
        return out
    check(L.psam_attention_f32(*args, _stream()), "psam_attention")
    return out


def attention_packed_supported(hd: int, rows: int, width: int) -> bool:
    """Self-attention straight from the qkv GEMM's packed output: head dim 64, and the qkv GEMM must be able to pack ([rows, 3 * width])."""
    return current_gemm_mode() == "f16x3" and hd == 64 and fuse_supported(rows, 3 * width)


def attention_packed(qkv_packed, scale_rows, out, out_scale, B, H, L, hd, scale, v_bound):
    """qkv_packed [B*L, >= 3*H*hd] g8-packed q | k | v with one common power-of-two scale (scale_rows [B*L], all equal): the output of
    linear(..., pack_out=(scale_rows, 0.0, bound)).  out [B*L, >= H*hd] receives the g8-packed attention output, out_scale [B*L] its
    (constant) scale f16_row_scale(v_bound): linear(out, W_proj, x_scale=out_scale, x_packed=True)."""
    qp, ld = _row_view(qkv_packed, "qkv_packed")
    op, ldo = _row_view(out, "out")
    check(_lib.load().psam_attention_packed(qp, ld, scale_rows.data_ptr(), op, ldo, out_scale.data_ptr(), B, H, L, hd, float(scale), float(v_bound), _stream()),
          "psam_attention_packed")
    return out


class EvaBlock:
    """One EVA02 (SwiGLU) transformer block prepared for psam_eva_block (csrc/blocks.hip): packed weights + bounds, built once at load by the
    library itself (psam_eva_block_prepare) from the state-dict tensors.  w: name -> fp32 device tensor; prefix 'pc_encoder.transformer.blocks.i'."""

    def __init__(self, w, prefix: str, dim: int, heads: int, hidden: int, eps: float):
        import ctypes
        L = _lib.load()
        wt = _lib.EvaBlockWeights()
        self.keep = []
        names = dict(norm1_w="norm1.weight", norm1_b="norm1.bias", q_w="attn.q_proj.weight", q_b="attn.q_proj.bias", k_w="attn.k_proj.weight",
                     v_w="attn.v_proj.weight", v_b="attn.v_proj.bias", proj_w="attn.proj.weight", proj_b="attn.proj.bias", norm2_w="norm2.weight",
                     norm2_b="norm2.bias", fc1_g_w="mlp.fc1_g.weight", fc1_g_b="mlp.fc1_g.bias", fc1_x_w="mlp.fc1_x.weight", fc1_x_b="mlp.fc1_x.bias",
                     mlp_norm_w="mlp.norm.weight", mlp_norm_b="mlp.norm.bias", fc2_w="mlp.fc2.weight", fc2_b="mlp.fc2.bias")
        for slot, n in names.items():
            t = w[f"{prefix}.{n}"]
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"{prefix}.{n}: contiguous fp32 expected")
            self.keep.append(t)
            setattr(wt, slot, t.data_ptr())
        wt.dim, wt.heads, wt.hidden, wt.eps = int(dim), int(heads), int(hidden), float(eps)
        self.plan = _lib.EvaBlockPlan()
        self.blob = torch.empty(int(L.psam_eva_block_prepared_bytes(dim, hidden)), dtype=torch.uint8, device=self.keep[0].device)
        check(L.psam_eva_block_prepare(ctypes.byref(wt), ctypes.byref(self.plan), self.blob.data_ptr(), self.blob.numel(), _stream()), "psam_eva_block_prepare")
        self.dim, self.hidden = int(dim), int(hidden)

    @staticmethod
    def supported(dim: int, heads: int, hidden: int) -> bool:
        return dim % heads == 0 and dim // heads == 64 and dim % 32 == 0 and 256 <= dim <= 4096 and ((hidden + 31) // 32 * 32) % 64 == 0      # 256..4096: the packed LayerNorm (layernorm_can_pack)

    def run(self, x, B: int, L: int, ws=None):
        """x [B*L, dim] fp32, updated in place."""
        import ctypes
        lib = _lib.load()
        _chk(x, name="x")
        M = B * L
        if x.shape != (M, self.dim) or not x.is_contiguous():
            raise ValueError("x must be a contiguous [B*L, dim] tensor")
        need = int(lib.psam_eva_block_ws_bytes(M, self.dim, self.hidden))
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        check(lib.psam_eva_block(ctypes.byref(self.plan), self.blob.data_ptr(), x.data_ptr(), B, L, ws.data_ptr(), ws.numel(), _stream()), "psam_eva_block")
        return x


class EvaGeluBlock:
    """One block of the giant encoder (fused qkv with q / v bias, GELU MLP; timm eva_giant_patch14_560) prepared for psam_eva_gelu_block
    (csrc/blocks.hip).  w: name -> fp32 device tensor; prefix 'pc_encoder.transformer.blocks.i'."""
    PRECISION_F16X3 = 2

    def __init__(self, w, prefix: str, dim: int, heads: int, hidden: int, eps: float):
        import ctypes
        L = _lib.load()
        wt = _lib.EvaGeluBlockWeights()
        self.keep = []
        names = dict(norm1_w="norm1.weight", norm1_b="norm1.bias", qkv_w="attn.qkv.weight", q_bias="attn.q_bias", v_bias="attn.v_bias", proj_w="attn.proj.weight",
                     proj_b="attn.proj.bias", norm2_w="norm2.weight", norm2_b="norm2.bias", fc1_w="mlp.fc1.weight", fc1_b="mlp.fc1.bias", fc2_w="mlp.fc2.weight",
                     fc2_b="mlp.fc2.bias")
        for slot, n in names.items():
            t = w[f"{prefix}.{n}"]
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"{prefix}.{n}: contiguous fp32 expected")
            self.keep.append(t)
            setattr(wt, slot, t.data_ptr())
        wt.dim, wt.heads, wt.hidden, wt.precision, wt.eps = int(dim), int(heads), int(hidden), self.PRECISION_F16X3, float(eps)
        self.plan = _lib.EvaGeluBlockPlan()
        self.blob = torch.empty(int(L.psam_eva_gelu_block_prepared_bytes(dim, hidden)), dtype=torch.uint8, device=self.keep[0].device)
        check(L.psam_eva_gelu_block_prepare(ctypes.byref(wt), ctypes.byref(self.plan), self.blob.data_ptr(), self.blob.numel(), _stream()), "psam_eva_gelu_block_prepare")
        self.dim, self.hidden = int(dim), int(hidden)

    @staticmethod
    def supported(dim: int, heads: int, hidden: int) -> bool:
        hd = dim // heads
        return dim % heads == 0 and 256 <= dim <= 4096 and dim % 32 == 0 and hidden % 32 == 0 and (hd == 64 or (64 < hd <= 128 and hd % 8 == 0))

    def run(self, x, B: int, L: int, ws=None):
        """x [B*L, dim] fp32, updated in place."""
        import ctypes
        lib = _lib.load()
        _chk(x, name="x")
        M = B * L
        if x.shape != (M, self.dim) or not x.is_contiguous():
            raise ValueError("x must be a contiguous [B*L, dim] tensor")
        need = int(lib.psam_eva_gelu_block_ws_bytes(M, self.dim, self.hidden))
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        # the key-split cap is per CONTEXT (a latency caller next to a multi-stream pipeline on the same model): a private copy of the plan (a small
        # POD the library reads during the call only) carries it; the shared plan is never written after _prepare
        plan = _lib.EvaGeluBlockPlan.from_buffer_copy(self.plan)
        plan.attn_keysplit = current_attention_keysplit()
        check(lib.psam_eva_gelu_block(ctypes.byref(plan), self.blob.data_ptr(), x.data_ptr(), B, L, ws.data_ptr(), ws.numel(), arrival_counters(x.device).data_ptr(), _stream()),
              "psam_eva_gelu_block")
        return x


class CPatchEncoder:
    """A PatchEncoder (common.py:477-506) prepared for psam_patch_encoder (csrc/blocks.hip).  prefix: 'pc_encoder.patch_embed.patch_encoder' |
    'mask_encoder.patch_encoder'."""

    def __init__(self, w, prefix: str, eps: float):
        import ctypes
        L = _lib.load()
        wt = _lib.PatchEncoderWeights()
        self.keep = []
        for slot, n in (("c10", "conv1.0"), ("c11", "conv1.1"), ("c13", "conv1.3"), ("c20", "conv2.0"), ("c21", "conv2.1"), ("c23", "conv2.3")):
            for suf, part in (("_w", ".weight"), ("_b", ".bias")):
                t = w[f"{prefix}.{n}{part}"]
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError(f"{prefix}.{n}{part}: contiguous fp32 expected")
                self.keep.append(t)
                setattr(wt, slot + suf, t.data_ptr())
        c10, c20, c23 = w[prefix + ".conv1.0.weight"], w[prefix + ".conv2.0.weight"], w[prefix + ".conv2.3.weight"]
        wt.cin, wt.h0, wt.h1, wt.cout, wt.eps = int(c10.shape[1]), int(c10.shape[0]), int(c20.shape[0]), int(c23.shape[0]), float(eps)
        self.h0, self.h1, self.cout = wt.h0, wt.h1, wt.cout
        self.plan = _lib.PatchEncoderPlan()
        self.blob = torch.empty(int(L.psam_patch_encoder_prepared_bytes(wt.h0, wt.h1, wt.cout)), dtype=torch.uint8, device=c10.device)
        check(L.psam_patch_encoder_prepare(ctypes.byref(wt), ctypes.byref(self.plan), self.blob.data_ptr(), self.blob.numel(), _stream()), "psam_patch_encoder_prepare")

    @staticmethod
    def supported(h0: int, h1: int, cout: int) -> bool:
        return h0 == 128 and 256 <= h1 <= 4096 and h1 % 128 == 0 and cout >= 128 and cout % 128 == 0

    def run(self, xyz, feats, centers, knn_idx, radius=None, center_idx=None):
        import ctypes
        lib = _lib.load()
        _chk(xyz); _chk(feats); _chk(centers); _chk(knn_idx, torch.int64)
        B, N, _ = xyz.shape
        rep = feats.shape[0] // B
        G, K = knn_idx.shape[1:]
        groups = B * rep * G
        out = torch.empty(groups, self.cout, dtype=torch.float32, device=xyz.device)
        ws = torch.empty(int(lib.psam_patch_encoder_ws_bytes(groups * K, groups, self.h0, self.h1)), dtype=torch.uint8, device=xyz.device)
        check(lib.psam_patch_encoder(ctypes.byref(self.plan), self.blob.data_ptr(), xyz.data_ptr(), feats.data_ptr(), centers.data_ptr(), knn_idx.data_ptr(),
                                     _p(center_idx), B, rep, N, G, K, feats.shape[-1], float(radius or 0.0), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
              "psam_patch_encoder")
        return out


class CUpscale:
    """The mask decoder's upscaling + hyper-network products prepared for psam_upscale_masks (csrc/blocks.hip; mask_decoder.py:146-176)."""

    def __init__(self, w, eps: float):
        import ctypes
        L = _lib.load()
        wt = _lib.UpscaleWeights()
        self.keep = []
        for slot, n in (("u0", "0"), ("u1", "1"), ("u3", "3")):
            for suf, part in (("_w", ".weight"), ("_b", ".bias")):
                t = w[f"mask_decoder.output_upscaling.{n}{part}"]
                self.keep.append(t)
                setattr(wt, slot + suf, t.data_ptr())
        wt.dim, wt.eps = int(w["mask_decoder.output_upscaling.0.weight"].shape[0]), float(eps)
        self.dim = wt.dim
        self.plan = _lib.UpscalePlan()
        self.blob = torch.empty(int(L.psam_upscale_masks_prepared_bytes(wt.dim)), dtype=torch.uint8, device=self.keep[0].device)
        check(L.psam_upscale_masks_prepare(ctypes.byref(wt), ctypes.byref(self.plan), self.blob.data_ptr(), self.blob.numel(), _stream()), "psam_upscale_masks_prepare")

    def run(self, keys, idx3, w3, hyper, masks, rep, Z, N, G, C):
        import ctypes
        lib = _lib.load()
        _chk(keys); _chk(idx3, torch.int64); _chk(w3); _chk(hyper); _chk(masks)
        ws = torch.empty(int(lib.psam_upscale_masks_ws_bytes(Z, N, G, C, self.dim)), dtype=torch.uint8, device=keys.device)
        check(lib.psam_upscale_masks(ctypes.byref(self.plan), self.blob.data_ptr(), keys.data_ptr(), idx3.data_ptr(), w3.data_ptr(), hyper.data_ptr(), rep, Z, N, G, C,
                                     masks.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "psam_upscale_masks")
        return masks


class CTwoWay:
    """The decoder's TwoWayTransformer prepared for psam_twoway_decoder (csrc/blocks.hip; transformer.py:61-100).  w: name -> fp32 tensor."""

    def __init__(self, w, prefix: str, depth: int, dim: int, heads: int, mlp: int, downsample: int, eps: float):
        import ctypes
        L = _lib.load()
        if depth > _lib.TWOWAY_MAX_DEPTH:
            raise ValueError("psam_twoway_decoder: depth > PSAM_TWOWAY_MAX_DEPTH")
        self.keep = []

        def P(name):
            t = w[name]
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"{name}: contiguous fp32 expected")
            self.keep.append(t)
            return t.data_ptr()

        def attn(dst, p):
            for slot, n in (("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"), ("o", "out_proj")):
                setattr(dst, slot + "_w", P(f"{p}.{n}.weight")); setattr(dst, slot + "_b", P(f"{p}.{n}.bias"))

        self.layers = (_lib.TwoWayLayerW * depth)()
        for i in range(depth):
            lp, lw = f"{prefix}.layers.{i}", self.layers[i]
            attn(lw.self_attn, lp + ".self_attn"); attn(lw.t2i, lp + ".cross_attn_token_to_image"); attn(lw.i2t, lp + ".cross_attn_image_to_token")
            for j in (1, 2, 3, 4):
                setattr(lw, f"n{j}_w", P(f"{lp}.norm{j}.weight")); setattr(lw, f"n{j}_b", P(f"{lp}.norm{j}.bias"))
            lw.m1_w, lw.m1_b, lw.m2_w, lw.m2_b = P(lp + ".mlp.lin1.weight"), P(lp + ".mlp.lin1.bias"), P(lp + ".mlp.lin2.weight"), P(lp + ".mlp.lin2.bias")
        wt = _lib.TwoWayWeights()
        wt.depth, wt.dim, wt.heads, wt.mlp, wt.downsample, wt.eps = depth, dim, heads, mlp, downsample, float(eps)
        wt.layers = ctypes.cast(self.layers, ctypes.POINTER(_lib.TwoWayLayerW))
        attn(wt.final_attn, prefix + ".final_attn_token_to_image")
        wt.nf_w, wt.nf_b = P(prefix + ".norm_final_attn.weight"), P(prefix + ".norm_final_attn.bias")
        self.dim, self.mlp = dim, mlp
        self.plan = _lib.TwoWayPlan()
        self.blob = torch.empty(int(L.psam_twoway_decoder_prepared_bytes(depth, dim, mlp, downsample)), dtype=torch.uint8, device=self.keep[0].device)
        check(L.psam_twoway_decoder_prepare(ctypes.byref(wt), ctypes.byref(self.plan), self.blob.data_ptr(), self.blob.numel(), _stream()), "psam_twoway_decoder_prepare")

    def run(self, tokens, keys, pos, rep, Z, T, G):
        """tokens [Z*T, E], keys [Z*G, E] (updated in place), pos [Z/rep, G, E] -> queries [Z*T, E]."""
        import ctypes
        lib = _lib.load()
        _chk(tokens); _chk(keys); _chk(pos)
        queries = torch.empty(Z * T, self.dim, dtype=torch.float32, device=tokens.device)
        ws = torch.empty(int(lib.psam_twoway_decoder_ws_bytes(Z, T, G, self.dim, self.mlp)), dtype=torch.uint8, device=tokens.device)
        check(lib.psam_twoway_decoder(ctypes.byref(self.plan), self.blob.data_ptr(), tokens.data_ptr(), keys.data_ptr(), pos.data_ptr(), rep, Z, T, G, queries.data_ptr(),
                                      ws.data_ptr(), ws.numel(), arrival_counters(tokens.device).data_ptr(), _stream()), "psam_twoway_decoder")
        return queries, keys


class TwoWayLayerWeights:
    """The weight pointers of one TwoWayAttentionBlock's token side (or of the final token -> image attention: final=True) as a
    psam_twoway_tokens_t skeleton (csrc/experiments/twoway.hip); keeps the tensors alive.  w: name -> fp32 tensor; prefix: e.g.
    'mask_decoder.transformer.layers.0' or, with final=True, 'mask_decoder.transformer' (final_attn_token_to_image / norm_final_attn)."""

    def __init__(self, w, prefix: str, final: bool = False):
        self.final = final
        self.keep = []
        a = self.args = _lib.TwoWayTokens()
        def put(slot, name):
            t = w[name]
            if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16:
                raise ValueError(f"{name}: the two-way token kernel needs contiguous, 16-byte aligned fp32 weights")
            self.keep.append(t)
            setattr(a, slot, t.data_ptr())
            return t
        if final:
            att, norm = prefix + ".final_attn_token_to_image", prefix + ".norm_final_attn"
        else:
            att, norm = prefix + ".cross_attn_token_to_image", prefix + ".norm2"
            for s, n in (("sq", "q_proj"), ("sk", "k_proj"), ("sv", "v_proj"), ("so", "out_proj")):
                put(s + "_w", f"{prefix}.self_attn.{n}.weight"); put(s + "_b", f"{prefix}.self_attn.{n}.bias")
            put("n1_g", prefix + ".norm1.weight"); put("n1_b", prefix + ".norm1.bias")
            m1 = put("m1_w", prefix + ".mlp.lin1.weight"); put("m1_b", prefix + ".mlp.lin1.bias")
            put("m2_w", prefix + ".mlp.lin2.weight"); put("m2_b", prefix + ".mlp.lin2.bias")
            put("n3_g", prefix + ".norm3.weight"); put("n3_b", prefix + ".norm3.bias")
            put("ik_w", prefix + ".cross_attn_image_to_token.k_proj.weight"); put("ik_b", prefix + ".cross_attn_image_to_token.k_proj.bias")
            put("iv_w", prefix + ".cross_attn_image_to_token.v_proj.weight"); put("iv_b", prefix + ".cross_attn_image_to_token.v_proj.bias")
            a.mlp = int(m1.shape[0])
        cq = put("cq_w", att + ".q_proj.weight"); put("cq_b", att + ".q_proj.bias")
        put("co_w", att + ".out_proj.weight"); put("co_b", att + ".out_proj.bias")
        put("n2_g", norm + ".weight"); put("n2_b", norm + ".bias")
        self.embed, self.inner = int(cq.shape[1]), int(cq.shape[0])
        a.mode = 1 if final else 0

    @staticmethod
    def supported(E: int, inner_cross: int, heads: int, Z: int, T: int, G: int) -> bool:
        return E == 256 and inner_cross == 128 and heads in (1, 2, 4, 8, 16, 32) and Z * T <= 64 and max(T, G) <= 4096


def _need_experiments(what: str):
    if not _lib.has_experiments():
        raise _lib.PointSamHipError(f"{what} is a measured-and-rejected path: build the library with PSAM_BUILD_EXPERIMENTS=1 (python -m point_sam_amd.build)")


def twoway_tokens_ws(mlp: int, device) -> torch.Tensor:
    _need_experiments("psam_twoway_tokens")
    return torch.empty(int(_lib.load().psam_twoway_tokens_ws_floats(int(mlp))), dtype=torch.float32, device=device)


def twoway_tokens(lw: TwoWayLayerWeights, queries, pe, kimg, vimg, Z, T, G, heads, eps, ws, ktok=None, vtok=None, skip_pe=False):
    """One launch for the token side of a two-way layer (csrc/experiments/twoway.hip; transformer.py:144-175): queries [Z*T, 256] updated in place; pe the
    token embeddings (query_pe); kimg / vimg [Z*G, 128] row views of the patch tokens' k / v projections; ktok / vtok [Z*T, 128] receive the
    k / v projections for the image -> token attention (not for the final attention)."""
    import ctypes
    _need_experiments("psam_twoway_tokens")
    a = lw.args
    _chk(queries, name="queries"); _chk(pe, name="pe")
    kp, ldk = _row_view(kimg, "kimg"); vp, ldv = _row_view(vimg, "vimg")
    a.Z, a.T, a.G, a.heads, a.skip_pe, a.eps = int(Z), int(T), int(G), int(heads), int(bool(skip_pe)), float(eps)
    a.queries, a.pe = queries.data_ptr(), pe.data_ptr()
    a.kimg, a.ldk, a.sk, a.vimg, a.ldv, a.sv = kp, ldk, G * ldk, vp, ldv, G * ldv
    a.ktok, a.vtok = (0, 0) if lw.final else (ktok.data_ptr(), vtok.data_ptr())
    a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    check(_lib.load().psam_twoway_tokens(ctypes.byref(a), _stream()), "psam_twoway_tokens")
    return queries


class Mlp3Weights:
    """Three Linear layers (ReLU between) of M stacked MLPs for psam_mlp3: w? [M, out, in] (the reference's layout), b? [M, out]."""

    def __init__(self, layers):
        # layers: list over MLPs of [(W1, b1), (W2, b2), (W3, b3)] with W [out, in]
        st = lambda i: (torch.stack([m[i][0] for m in layers]).contiguous(), torch.stack([m[i][1] for m in layers]).contiguous())
        (self.w1, self.b1), (self.w2, self.b2), (self.w3, self.b3) = st(0), st(1), st(2)
        self.M, self.dh, self.din = self.w1.shape
        self.dout = self.w3.shape[1]
        if self.w2.shape[1:] != (self.dh, self.dh) or self.w3.shape[2] != self.dh:
            raise ValueError("Mlp3Weights: layer shapes do not chain")


def mlp3(x, ldx, sx, mw: Mlp3Weights, out, ldo, so, Z):
    """out[z, m] = W3_m relu(W2_m relu(W1_m x[z, m] + b1) + b2) + b3: row (z, m) of x at x.data_ptr() + 4 * (z * ldx + m * sx) (fp32
    elements), of out at + 4 * (z * ldo + m * so).  One launch for all M MLPs and Z rows (mask_decoder.py:171-180,189-211)."""
    for t in (x, out):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise _lib.PointSamHipError("mlp3: operands must be fp32 tensors on the GPU (there is no CPU fallback)")
    check(_lib.load().psam_mlp3(x.data_ptr(), ldx, sx, mw.w1.data_ptr(), mw.b1.data_ptr(), mw.w2.data_ptr(), mw.b2.data_ptr(), mw.w3.data_ptr(),
                                mw.b3.data_ptr(), out.data_ptr(), ldo, so, Z, mw.M, mw.din, mw.dh, mw.dout, _stream()), "psam_mlp3")
    return out


def mlp3_pair(xa, ldxa, sxa, mwa: Mlp3Weights, outa, ldoa, soa, xb, ldxb, sxb, mwb: Mlp3Weights, outb, ldob, sob, Z):
    """mlp3 of stack a and of stack b over the same Z rows in ONE launch (the hyper-networks and the IoU head, mask_decoder.py:167-182): the same
    bits as two mlp3 calls."""
    for t in (xa, outa, xb, outb):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise _lib.PointSamHipError("mlp3_pair: operands must be fp32 tensors on the GPU (there is no CPU fallback)")
    def args(x, ldx, sx, mw, out, ldo, so):
        return _lib.Mlp3Args(x.data_ptr(), mw.w1.data_ptr(), mw.b1.data_ptr(), mw.w2.data_ptr(), mw.b2.data_ptr(), mw.w3.data_ptr(), mw.b3.data_ptr(), out.data_ptr(),
                             ldx, sx, ldo, so, mw.M, mw.din, mw.dh, mw.dout)
    a, b = args(xa, ldxa, sxa, mwa, outa, ldoa, soa), args(xb, ldxb, sxb, mwb, outb, ldob, sob)
    check(_lib.load().psam_mlp3_pair(ctypes.byref(a), ctypes.byref(b), Z, _stream()), "psam_mlp3_pair")
    return outa, outb


def skinny_ln_supported(M, N, K) -> bool:
    """linear_skinny_ln's shapes."""
    return 0 < M <= SKINNY_MAX_M and N == 256 and K % 16 == 0


def linear_skinny_ln(x, W, bias, ln_w, ln_b, eps, residual=None, out=None):
    """out [M, 256] = LayerNorm(x W^T + bias + residual) * ln_w + ln_b for M <= 64 rows in ONE launch (csrc/gemm.hip linear_skinny_ln_kernel: the last
    workgroup to finish its columns normalises the rows) -- the `queries = norm(queries + out_proj(attn))` steps of the decoder (transformer.py:153-176)."""
    for t, n in ((ln_w, "ln_w"), (ln_b, "ln_b")):
        _chk(t, name=n)
    M, K = x.shape
    N = W.shape[0]
    xp, ldx = _row_view(x, "x"); wp, ldw = _row_view(W, "W")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    op, ldo = _row_view(out, "out")
    rp, ldr = (0, 0) if residual is None else _row_view(residual, "residual")
    L = _lib.load()
    tmp = torch.empty(L.psam_linear_skinny_ln_tmp_floats(M, K), dtype=torch.float32, device=x.device)
    check(L.psam_linear_skinny_ln(xp, ldx, wp, ldw, _p(bias), rp, ldr, ln_w.data_ptr(), ln_b.data_ptr(), eps, tmp.data_ptr(), op, ldo, M, N, K,
                                  arrival_counters(x.device).data_ptr(), _stream()), "psam_linear_skinny_ln")
    return out


def linear_rows_multi(x, jobs, xadd_rows_per_set=1, xadd_rep=1):
    """[y_i] = [act_i((x + xadd_i) W_i^T + bias_i)] for jobs = [(W, bias | None, xadd | None, act), ..] over the same rows of x in ONE launch, exact fp32
    products (csrc/gemm.hip linear_rows_multi_kernel: 32 x 64 tiles; a few hundred to a few thousand rows).  xadd_i: sets of xadd_rows_per_set rows, set
    (row // (xadd_rep * xadd_rows_per_set)) is added to row `row` (the decoder's key_pe)."""
    xp, ldx = _row_view(x, "x")
    M, K = x.shape
    js = _lib.SkinnyJobs()
    js.n = len(jobs)
    outs, ldw, ldxa = [], None, ldx
    for i, (W, bias, xadd, act) in enumerate(jobs):
        wp, lw = _row_view(W, "W")
        if ldw not in (None, lw):
            raise ValueError("linear_rows_multi: the weights must share their row stride")
        ldw = lw
        y = torch.empty(M, W.shape[0], dtype=torch.float32, device=x.device)
        outs.append(y)
        ap = 0
        if xadd is not None:
            ap, ldxa = _row_view(xadd, "xadd")
        js.job[i] = _lib.SkinnyJob(xp, ap, wp, _p(bias), y.data_ptr(), y.stride(0), W.shape[0], act)
    check(_lib.load().psam_linear_rows_multi(ctypes.byref(js), ldx, ldxa, xadd_rows_per_set, xadd_rep, ldw, M, K, _stream()), "psam_linear_rows_multi")
    return outs


def linear_ln256(x, W, bias, ln_w, ln_b, eps, residual=None, out=None):
    """out [M, 256] = LayerNorm(x W^T + bias + residual) * ln_w + ln_b for any M and a short K (K % 16 == 0, K <= 512) in one launch (csrc/gemm.hip
    linear_ln256_kernel: a workgroup per 16 whole rows, exact fp32 products) -- `keys = norm4(keys + out_proj(attn))`, transformer.py:170-175."""
    for t, n in ((ln_w, "ln_w"), (ln_b, "ln_b")):
        _chk(t, name=n)
    M, K = x.shape
    N = W.shape[0]
    xp, ldx = _row_view(x, "x"); wp, ldw = _row_view(W, "W")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    op, ldo = _row_view(out, "out")
    rp, ldr = (0, 0) if residual is None else _row_view(residual, "residual")
    check(_lib.load().psam_linear_ln256(xp, ldx, wp, ldw, _p(bias), rp, ldr, ln_w.data_ptr(), ln_b.data_ptr(), eps, op, ldo, M, N, K, _stream()), "psam_linear_ln256")
    return out


def attention_small(q, k, v, out, Z, H, Lq, Lk, hd, scale):
    qp, ldq = _row_view(q, "q"); kp, ldk = _row_view(k, "k"); vp, ldv = _row_view(v, "v"); op, ldo = _row_view(out, "out")
    check(_lib.load().psam_attention_small(qp, ldq, Lq * ldq, kp, ldk, Lk * ldk, vp, ldv, Lk * ldv, op, ldo, Lq * ldo, Z, H, Lq, Lk, hd, scale,
                                           _stream()), "psam_attention_small")
    return out


# ------------------------------------------------------------------------------------------ encodings etc.
def pos_l1(centers, W, bias, out=None):
    _chk(centers)
    rows = centers.numel() // 3
    if out is None:
        out = torch.empty(rows, 128, dtype=torch.float32, device=centers.device)
    check(_lib.load().psam_pos_l1(centers.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), rows, _stream()), "psam_pos_l1")
    return out


def fourier_pe(coords, gauss, out, rows_per_batch, batch_stride, labels=None, emb0=None, emb1=None, flag=None):
    """coords [..., 3] -> writes [sin,cos] (+label embedding) rows into ``out`` (see header for the row mapping)."""
    _chk(coords)
    rows = coords.numel() // 3
    F = gauss.shape[1]
    if labels is not None:
        _chk(labels, torch.int64, "labels")
    check(_lib.load().psam_fourier_pe(coords.data_ptr(), gauss.data_ptr(), F, _p(labels), _p(emb0), _p(emb1), out.data_ptr(), rows, rows_per_batch,
                                      batch_stride, _p(flag), _stream()), "psam_fourier_pe")
    return out


def add_bcast(a, rep, b, out, Z, R, C, sa=None, sb=None, ldb=None, so=None):
    sa = R * C if sa is None else sa
    so = R * C if so is None else so
    if b is None:
        sb, ldb = 0, 0
    else:
        sb = R * C if sb is None else sb
        ldb = C if ldb is None else ldb
    check(_lib.load().psam_add_bcast(a.data_ptr(), sa, rep, _p(b), sb, ldb, out.data_ptr(), so, Z, R, C, _stream()), "psam_add_bcast")
    return out


def interp3(src, idx3, w3, out, rep, scale_out=None, ln=None, act=ACT_NONE):
    """src [Z,G,C], idx3/w3 [B,N,3] -> out [Z,N,C]; scale_out [Z*N] (C == 256): out receives the g8-packed rows + their scales;
    ln=(gamma, beta, eps) (C == 256): LayerNorm and the activation `act` applied to every interpolated row."""
    Z, G, C = src.shape
    N = idx3.shape[1]
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    check(_lib.load().psam_interp3_ex(src.data_ptr(), idx3.data_ptr(), w3.data_ptr(), out.data_ptr(), rep, Z, N, G, C, _p(scale_out), _p(g), _p(b),
                                      float(eps), int(act), _stream()), "psam_interp3")
    return out


# ------------------------------------------------------------------------------------------ click simulation (eval protocol)
def error_regions(gt: torch.Tensor, logits):
    """gt [Z,N] uint8/bool, logits [Z,N] f32 or None -> (fn, fp) uint8 [Z,N].  common.py:388-405."""
    gt = _chk(gt.to(torch.uint8), torch.uint8, "gt")
    if logits is not None:
        _chk(logits, name="logits")
    fn, fp = torch.empty_like(gt), torch.empty_like(gt)
    check(_lib.load().psam_error_regions(gt.data_ptr(), _p(logits), fn.data_ptr(), fp.data_ptr(), gt.numel(), _stream()), "psam_error_regions")
    return fn, fp


def border_farthest(xyz: torch.Tensor, region: torch.Tensor):
    """xyz [B,N,3], region [Z,N] uint8 -> (idx [Z] int64, dist2 [Z]); -1 where the region or its complement is empty.
    common.py:443-474."""
    _chk(xyz, name="xyz")
    region = _chk(region.to(torch.uint8), torch.uint8, "region")
    B, N, _ = xyz.shape
    Z = region.shape[0]
    L = _lib.load()
    nbytes = L.psam_border_farthest_workspace_bytes(Z, N)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=xyz.device)
    idx = torch.empty(Z, dtype=torch.int64, device=xyz.device)
    dist = torch.empty(Z, dtype=torch.float32, device=xyz.device)
    check(L.psam_border_farthest(xyz.data_ptr(), region.data_ptr(), B, Z // B, N, idx.data_ptr(), dist.data_ptr(), ws.data_ptr(), nbytes, _stream()),
          "psam_border_farthest")
    return idx, dist
