"""Per-kernel parity on a real MI355X: every C-ABI entry point against the CPU oracle (bit-exact for index work,
fp32 tolerance for floating point) on the same seeded inputs.  Run with ``pytest -m gpu``."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pointsam_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from point_sam_amd import ops as _ops
    _ops._lib.load()
    return _ops


def cu(t):
    return t.cuda().contiguous()


def _close(got, want, atol, rtol=0.0, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max abs err {err.max():.3e} (max |want| {want.abs().max():.3e})"


def _cloud(B, N, seed, dup=0):
    xyz, rgb, _, _ = O.synthetic_batch(B, N, seed=seed)
    if dup:  # exact duplicate points, like demo/static/models/*.ply
        g = torch.Generator().manual_seed(seed + 99)
        src = torch.randint(0, N, (dup,), generator=g)
        dst = torch.randint(0, N, (dup,), generator=g)
        xyz[:, dst] = xyz[:, src]
    return xyz, rgb


# ------------------------------------------------------------------------------------------------ tokenizer
@pytest.mark.parametrize("B,N,G,dup", [(2, 1024, 32, 0), (1, 777, 32, 0), (2, 4096, 128, 0), (1, 5000, 64, 300), (2, 32768, 512, 0),
                                       (1, 20000, 256, 5000), (1, 40000, 48, 0), (1, 64, 64, 0)])
def test_fps_bit_exact(ops, B, N, G, dup):
    xyz, _ = _cloud(B, N, seed=N + G, dup=dup)
    want = O.fps(xyz, G)
    idx, centers = ops.fps(cu(xyz), G)
    got = idx.cpu()
    if not torch.equal(got, want):
        first = (got != want).nonzero()[0].tolist()
        raise AssertionError(f"FPS differs: {(got != want).sum().item()} of {got.numel()} indices, first at {first}: got {got[tuple(first)]} want {want[tuple(first)]}")
    assert torch.equal(centers.cpu(), O.batch_index_select(xyz, want))


@pytest.mark.parametrize("B,N,G,dup", [(1, 131072, 2048, 0), (2, 70000, 300, 20000), (3, 33000, 64, 0), (1, 300000, 40, 0), (1, 600000, 24, 0),
                                           (3, 32768, 200, 5000), (2, 30000, 100, 0)])
def test_fps_cooperative_bit_exact(ops, B, N, G, dup):
    """N > 32768: the multi-workgroup FPS (per-iteration exchange of tagged candidate keys between workgroups) must give the
    oracle's indices bit for bit, and the same as the single-workgroup streaming kernel."""
    xyz, _ = _cloud(B, N, seed=N + G, dup=dup)
    want = O.fps(xyz, G)
    idx, centers = ops.fps(cu(xyz), G)
    assert torch.equal(idx.cpu(), want)
    assert torch.equal(centers.cpu(), O.batch_index_select(xyz, want))
    L = ops._lib.load()
    for mode in (0, 2):      # single-workgroup kernel; cooperative kernel with its workgroups spread over the XCDs
        L.psam_fps_set_cooperative(mode)
        try:
            idx1, _ = ops.fps(cu(xyz), G)
        finally:
            L.psam_fps_set_cooperative(1)
        assert torch.equal(idx1, idx), mode


def _clustered_cloud(B, N, seed):
    """Points in a few tight clusters plus a thin uniform background, quantised to a coarse lattice: many exactly equal distances (ties decided by the
    index), cells of very unequal population."""
    g = torch.Generator().manual_seed(seed)
    k = 7
    centres = torch.rand(B, k, 3, generator=g) * 1.6 - 0.8
    which = torch.randint(0, k, (B, N), generator=g)
    xyz = torch.gather(centres, 1, which[..., None].expand(-1, -1, 3)) + 0.02 * torch.randn(B, N, 3, generator=g)
    bg = torch.rand(B, N, generator=g) < 0.05
    xyz[bg] = torch.rand(int(bg.sum()), 3, generator=g) * 2 - 1
    return (xyz * 64).round() / 64


@pytest.mark.parametrize("B,N,G,kind", [(1, 131072, 2048, "surface"), (2, 70000, 500, "dup"), (1, 131072, 700, "clustered"), (2, 32768, 512, "surface"),
                                        (3, 50000, 64, "clustered"), (1, 40000, 300, "identical"), (1, 100000, 256, "line"), (1, 29000, 128, "surface")])
def test_fps_pruned_bit_exact(ops, B, N, G, kind):
    """The cooperative FPS with exact spatial pruning (csrc/tokenizer.hip fps_coop_pruned_kernel: the cloud bucketed by grid cell, a wave skips its scan
    when the new centre's distance to the wave's bounding box -- in the scan's own arithmetic -- cannot lower any of its running minima) gives the
    oracle's indices bit for bit and the un-pruned kernel's: surface-like clouds, clouds with duplicates and with many exactly tied distances (lattice
    coordinates in clusters), a cloud of identical points (zero-extent box), a cloud on a line (two zero extents), several clouds per call, and the two
    smallest cooperative sizes."""
    if kind == "surface":
        xyz, _ = _cloud(B, N, seed=N + G)
    elif kind == "dup":
        xyz, _ = _cloud(B, N, seed=N + G, dup=N // 3)
    elif kind == "clustered":
        xyz = _clustered_cloud(B, N, seed=N + G)
    elif kind == "identical":
        xyz = torch.full((B, N, 3), 0.375)
    else:
        t = torch.rand(B, N, 1, generator=torch.Generator().manual_seed(N))
        xyz = torch.cat([t * 1.5 - 0.75, torch.full((B, N, 1), 0.25), torch.full((B, N, 1), -0.5)], -1)
    want = O.fps(xyz, G)
    L = ops._lib.load()
    out = {}
    try:
        for mode in (1, 0, 1):
            L.psam_fps_set_pruning(mode)
            idx, centers = ops.fps(cu(xyz), G)
            if mode in out:
                assert torch.equal(idx, out[mode][0]), "pruned FPS: not repeatable"
            out[mode] = (idx, centers)
    finally:
        L.psam_fps_set_pruning(-1)
    got = out[1][0].cpu()
    if not torch.equal(got, want):
        first = (got != want).nonzero()[0].tolist()
        raise AssertionError(f"pruned FPS differs from the oracle: {(got != want).sum().item()} of {got.numel()} indices, first at {first}: got {got[tuple(first)]} want {want[tuple(first)]}")
    assert torch.equal(out[0][0].cpu(), want), "un-pruned cooperative FPS differs from the oracle"
    assert torch.equal(out[1][1].cpu(), O.batch_index_select(xyz, want))


@pytest.mark.parametrize("seed", range(8))
def test_fps_pruned_random_shapes(ops, seed):
    """Pruned cooperative FPS on randomly drawn shapes and coordinate ranges: clouds scaled by 1e-3 .. 1e3, shifted far from the origin (coordinates that
    cancel in x - c), flattened onto an axis-aligned plane (one zero box extent), mixed with duplicates -- the bounding-box argument must hold for any finite
    fp32 coordinates, so the indices equal the oracle's bit for bit."""
    g = torch.Generator().manual_seed(1000 + seed)
    B = 1 + seed % 2
    N = int(torch.randint(28673, 90000, (1,), generator=g))
    G = int(torch.randint(24, 160, (1,), generator=g))
    xyz, _ = _cloud(B, N, seed=seed, dup=(N // 5 if seed % 3 == 0 else 0))
    xyz = xyz * (10.0 ** float(torch.randint(-3, 4, (1,), generator=g)))
    if seed % 4 == 1:
        xyz = xyz + torch.tensor([1000.0, -250.0, 31.0])
    if seed % 4 == 2:
        xyz[..., int(torch.randint(0, 3, (1,), generator=g))] = 0.125
    want = O.fps(xyz, G)
    L = ops._lib.load()
    try:
        L.psam_fps_set_pruning(1)
        idx, centers = ops.fps(cu(xyz), G)
    finally:
        L.psam_fps_set_pruning(-1)
    got = idx.cpu()
    assert torch.equal(got, want), f"B={B} N={N} G={G}: {(got != want).sum().item()} of {got.numel()} indices differ, first at {(got != want).nonzero()[0].tolist()}"
    assert torch.equal(centers.cpu(), O.batch_index_select(xyz, want))


def test_fps_cooperative_under_memory_load(ops):
    """The cooperative FPS exchanges keys between workgroups through device memory; run it several times while another stream
    saturates HBM / the fabric with large copies and GEMMs (uneven load is what exposes an unordered key store vs barrier arrival)
    -- every run must still give the oracle's indices bit for bit."""
    B, N, G = 2, 131072, 600
    xyz, _ = _cloud(B, N, seed=77)
    want = O.fps(xyz, G)
    x = cu(xyz)
    big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    a = torch.randn(4096, 1024, device="cuda")
    w = ops.F16Weight(torch.randn(3072, 1024, device="cuda"))
    side = torch.cuda.Stream()
    stop = torch.cuda.Event()
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(6):
                big[: 256 << 20].copy_(big[256 << 20:], non_blocking=True)
                with ops.gemm_mode("f16x3"):
                    ops.linear(a, w)
        idx, _ = ops.fps(x, G)
        stop.record()
        assert torch.equal(idx.cpu(), want), f"run {rep}: cooperative FPS under load differs from the oracle"
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,N,G,K,dup", [(2, 1024, 32, 16, 0), (1, 777, 32, 16, 0), (2, 4096, 128, 32, 0), (1, 32768, 96, 64, 0),
                                         (1, 5000, 64, 64, 2500), (1, 3000, 16, 256, 0), (1, 100, 8, 100, 40), (1, 20000, 32, 1000, 0)])
def test_knn_bit_exact(ops, B, N, G, K, dup):
    xyz, _ = _cloud(B, N, seed=3 * N + K, dup=dup)
    centers = O.batch_index_select(xyz, O.fps(xyz, G))
    _, want = O.knn(centers, xyz, K, "exact")
    got = ops.knn(cu(centers), cu(xyz), K).cpu()
    if not torch.equal(got, want):
        rows = (got != want).any(-1).nonzero()
        raise AssertionError(f"kNN differs in {len(rows)} of {B * G} groups; first {rows[0].tolist()}: got {got[tuple(rows[0])][:8]} want {want[tuple(rows[0])][:8]}")


@pytest.mark.parametrize("B,N,G,K,dup", [(2, 32768, 512, 64, 0), (3, 4999, 64, 37, 0), (1, 131072, 96, 256, 0), (2, 6000, 128, 64, 2500), (1, 1001, 16, 1001, 0), (1, 2600, 8, 100, 2590)])
def test_knn_band_kernel_equals_four_pass_kernel(ops, B, N, G, K, dup):
    """The band kernel (csrc/tokenizer.hip knn_band_kernel: one histogram sweep, then the two lower radix passes and the cut on the selected bin's points
    only -- two distance evaluations per pair instead of four; common.py:27-56) gives the four-pass kernel's indices bit for bit: aligned and unaligned
    clouds (a batch whose clouds start off 16-byte boundaries, N % 4 != 0), K = N, and clouds of duplicates whose band overflows the candidate list or
    whose K-th distance is tied (the general path inside the same kernel)."""
    L = ops._lib.load()
    xyz, _ = _cloud(B, N, seed=11 * N + K, dup=dup)
    if N == 2600:
        xyz[:, :dup] = xyz[:, :1]      # 2590 copies of one point: every distance to them is equal -- the band overflows the candidate list and the K-th value is tied
    centers = O.batch_index_select(xyz, O.fps(xyz, G))
    out = {}
    try:
        for mode in (0, 1):
            L.psam_knn_force_band(mode)
            out[mode] = ops.knn(cu(centers), cu(xyz), K)
        torch.cuda.synchronize()
    finally:
        L.psam_knn_force_band(-1)
    assert torch.equal(out[0], out[1]), f"{(out[0] != out[1]).any(-1).sum().item()} of {B * G} groups differ"


def test_knn_all_points_identical(ops):
    """Degenerate tie: every distance equal -> the K lowest indices."""
    xyz = torch.full((1, 500, 3), 0.25)
    got = ops.knn(cu(xyz[:, :4]), cu(xyz), 37).cpu()
    assert torch.equal(got, torch.arange(37).expand(1, 4, 37))


@pytest.mark.parametrize("B,N,G", [(2, 1024, 32), (1, 777, 17), (2, 32768, 512), (1, 5000, 2048)])
def test_three_nn(ops, B, N, G):
    xyz, _ = _cloud(B, N, seed=N + 7 * G)
    centers = O.batch_index_select(xyz, O.fps(xyz, G))
    want_i, want_w = O.interp_weights(xyz, centers, "exact")
    got_i, got_w = ops.three_nn(cu(xyz), cu(centers))
    assert torch.equal(got_i.cpu(), want_i), f"{(got_i.cpu() != want_i).sum().item()} index mismatches"
    _close(got_w, want_w, 1e-6, what="3-NN weights")


def test_group_gather_and_patch_l1(ops):
    B, N, G, K = 2, 2048, 48, 16
    xyz, rgb = _cloud(B, N, seed=5)
    centers = O.batch_index_select(xyz, O.fps(xyz, G))
    _, kidx = O.knn(centers, xyz, K, "exact")
    want = O.group_points(xyz, rgb, centers, kidx)
    got = ops.group_gather(cu(xyz), cu(rgb), cu(centers), cu(kidx))
    assert torch.equal(got.cpu(), want)
    g = torch.Generator().manual_seed(1)
    for C, rep in ((3, 1), (1, 3)):
        feats = rgb if C == 3 else torch.randn(B * rep, N, 1, generator=g)
        W = torch.randn(128, 3 + C, generator=g) * 0.5
        b, lw, lb = torch.randn(128, generator=g) * 0.1, 1 + 0.1 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
        if C == 3:
            grouped = want
        else:
            rel = want[..., :3].repeat_interleave(rep, 0)
            ki = kidx.repeat_interleave(rep, 0)
            grouped = torch.cat([rel, torch.gather(feats[..., 0], 1, ki.reshape(B * rep, -1)).reshape(B * rep, G, K, 1)], -1)
        ref = F.gelu(F.layer_norm(F.linear(grouped, W, b), (128,), lw, lb, 1e-5)).reshape(-1, 128)
        out = ops.patch_l1(cu(xyz), cu(feats), cu(centers), cu(kidx), cu(W), cu(b), cu(lw), cu(lb), 1e-5)
        _close(out, ref, 2e-5, what=f"patch_l1 C={C}")


# ------------------------------------------------------------------------------------------------ dense
@pytest.mark.parametrize("cfg", [0, 1, 2, -1])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (257, 130, 36), (64, 200, 4), (1000, 96, 516), (33, 7, 128), (512, 2752, 64)])
def test_gemm_shapes(ops, cfg, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ops._lib.load().psam_gemm_force_config(cfg)
    try:
        y = ops.linear(cu(x), cu(W), cu(b))
    finally:
        ops._lib.load().psam_gemm_force_config(-1)
    want = (x.double() @ W.double().T + b.double())
    _close(y, want, 1e-5 * math.sqrt(K) * 4, what=f"gemm {M}x{N}x{K} cfg{cfg}")


def test_gemm_asymmetric_identity(ops):
    """A = I with an asymmetric W catches transposed / permuted output layouts."""
    n = 128
    W = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    y = ops.linear(cu(torch.eye(n)), cu(W))
    assert torch.equal(y.cpu(), W.T.contiguous())


def test_gemm_epilogues_and_views(ops):
    g = torch.Generator().manual_seed(0)
    M, N, K, grp = 192, 160, 96, 16
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, 2 * K, generator=g), torch.randn(N, generator=g)
    res, rb = torch.randn(M, N, generator=g), torch.randn(M // grp, N, generator=g)
    Wd = cu(W)
    base = x.double() @ W[:, K:].double().T
    y = ops.linear(cu(x), Wd[:, K:], cu(b), act=ops.ACT_GELU)                       # column-sliced weight view
    _close(y, F.gelu(base + b.double()), 1e-4, what="gelu epilogue")
    y = ops.linear(cu(x), Wd[:, K:], None, act=ops.ACT_RELU, rowbias=cu(rb), rowgroup=grp)
    _close(y, F.relu(base + rb.double().repeat_interleave(grp, 0)), 1e-4, what="rowbias+relu")
    xr = cu(res.clone())
    ops.linear(cu(x), Wd[:, K:], cu(b), residual=xr, out=xr)                        # in-place residual
    _close(xr, base + b.double() + res.double(), 1e-4, what="residual in place")
    # batched, strided: masks[z] = hyper[z] @ up[z]^T
    Z, C, Np, E = 3, 3, 500, 64
    hyper, up = torch.randn(Z, C, E, generator=g), torch.randn(Z, Np, E, generator=g)
    out = torch.empty(Z, C, Np, device="cuda")
    ops.gemm_batched(cu(hyper), cu(up), out, C, Np, E, E, E, Np, C * E, Np * E, C * Np, Z)
    _close(out, hyper.double() @ up.double().transpose(1, 2), 1e-4, what="batched gemm")


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 392, 516), (4096, 1024, 1024), (300, 130, 36), (128, 128, 32), (520, 640, 2752)])
def test_gemm_bf16x6_matches_fp64_like_f32(ops, M, N, K):
    """The split-bf16 GEMM must be as accurate as the f32-MFMA GEMM (both compared with an fp64 reference), also on
    operands with a wide dynamic range, ragged M/N and a K tail."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))   # rows spanning ~4 decades
    W = torch.randn(N, K, generator=g) * torch.exp(torch.randn(1, K, generator=g)) / K ** 0.5
    b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    want = F.gelu(x.double() @ W.double().T + b.double()) + res.double()
    scale = (x.double().abs() @ W.double().abs().T + 1.0)
    L = ops._lib.load()
    args = lambda y: (cu(x).data_ptr(), K, 0, 0, cu(W).data_ptr(), K, 0, 0, y.data_ptr(), N, 0, 0, cu(b).data_ptr(), cu(res).data_ptr(), N, 0, 0, 0, 0, 0,
                      M, N, K, 1, 1, 1.0, 1, torch.cuda.current_stream().cuda_stream)
    xd, Wd, bd, rd = cu(x), cu(W), cu(b), cu(res)
    errs = {}
    for name, fn in (("f32", L.psam_gemm_f32), ("bf16x6", L.psam_gemm_bf16x6)):
        y = torch.empty(M, N, device="cuda")
        rc = fn(xd.data_ptr(), K, 0, 0, Wd.data_ptr(), K, 0, 0, y.data_ptr(), N, 0, 0, bd.data_ptr(), rd.data_ptr(), N, 0, 0, 0, 0, 0, M, N, K, 1, 1, 1.0, 1,
                torch.cuda.current_stream().cuda_stream)
        assert rc == 0, L.psam_last_error_string()
        errs[name] = ((y.cpu().double() - want).abs() / scale).max().item()
    assert errs["bf16x6"] < 3e-7 * math.sqrt(K) + 1e-7, errs
    assert errs["bf16x6"] < 4 * errs["f32"] + 1e-7, errs


F16X3P_CFGS = [-1, 0, 4, 9, 12, 14, 21, 23, 28]   # tile / ring configurations of csrc/gemm_f16x3p.hip (-1 = the host's choice)


@pytest.mark.parametrize("cfg", F16X3P_CFGS)
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 392, 516), (4096, 1024, 1024), (300, 130, 160), (520, 640, 2752), (777, 3072, 256)])
def test_gemm_f16x3_matches_fp64_like_f32(ops, cfg, M, N, K):
    """The packed-operand, scaled 2-way fp16 split GEMM must be in the accuracy class of the f32-MFMA GEMM (both against fp64): operands
    with a wide dynamic range across rows AND along k, ragged M/N, K not a multiple of the 32-k slab (zero padding), bias + GELU +
    residual epilogue -- for every tile / ring configuration."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))
    W = torch.randn(N, K, generator=g) * torch.exp(torch.randn(1, K, generator=g)) / K ** 0.5
    b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    want = F.gelu(x.double() @ W.double().T + b.double()) + res.double()
    scale = (x.double().abs() @ W.double().abs().T + 1.0)
    L = ops._lib.load()
    fw = ops.F16Weight(cu(W))
    errs = {}
    L.psam_gemm_f16x3p_force_config(cfg)
    try:
        for mode in ("f32", "f16x3"):
            with ops.gemm_mode(mode):
                y = ops.linear(cu(x), fw, cu(b), act=ops.ACT_GELU, residual=cu(res))
            errs[mode] = ((y.cpu().double() - want).abs() / scale).max().item()
    finally:
        L.psam_gemm_f16x3p_force_config(-1)
    assert errs["f16x3"] < 3e-7 * math.sqrt(K) + 1e-7, errs
    assert errs["f16x3"] < 4 * errs["f32"] + 1e-7, errs


def test_gemm_f16x3_extreme_row_scales(ops):
    """Row scales make the fp16 split range-free: rows from 1e-33 to 1e+29 (times 1e-6..1e+6 weight rows: the whole finite fp32 range of the outputs), an all-zero row, and a row whose elements span
    18 binary orders of magnitude (small elements go subnormal in the lo plane: absolute error 2^-39 of the row maximum)."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 512, 256, 256
    x = torch.randn(M, K, generator=g)
    x *= (10.0 ** torch.linspace(-33, 29, M)).unsqueeze(1)
    x[7] = 0
    x[9] = torch.randn(K, generator=g) * (2.0 ** -torch.arange(K).remainder(19).float())
    W = torch.randn(N, K, generator=g) * (10.0 ** torch.linspace(-6, 6, N)).unsqueeze(1)
    with ops.gemm_mode("f16x3"):
        y = ops.linear(cu(x), ops.F16Weight(cu(W))).cpu().double()
    want = x.double() @ W.double().T
    scale = x.double().abs() @ W.double().abs().T
    assert torch.isfinite(y).all()
    assert (y[7] == 0).all()
    rel = ((y - want).abs() / scale.clamp_min(1e-300))
    rel[7] = 0
    assert rel.max().item() < 2e-6, rel.max().item()
    s = ops.row_scale_f16(cu(x)).cpu()
    assert s[7] == 1 and (torch.log2(s) == torch.log2(s).round()).all()
    m = x.abs().amax(1) * s
    ok = (m >= 2.0 ** 14) & (m < 2.0 ** 15)
    ok[7] = True
    assert ok.all()


@pytest.mark.parametrize("cols", [128, 300, 1024, 2752, 5000])
def test_layernorm_emits_f16x3_row_scale(ops, cols):
    g = torch.Generator().manual_seed(cols)
    x = (torch.randn(77, cols, generator=g) * torch.exp(3 * torch.randn(77, 1, generator=g))).cuda()
    w, b = torch.randn(cols, generator=g).cuda(), torch.randn(cols, generator=g).cuda()
    for act in (ops.ACT_NONE, ops.ACT_GELU):
        rs = torch.empty(77, device="cuda")
        y = ops.layernorm(x, w, b, 1e-5, act=act, scale_out=rs)
        assert torch.equal(rs, ops.row_scale_f16(y))
        assert torch.equal(y, ops.layernorm(x, w, b, 1e-5, act=act))


def _unpack_g8(p, scale, K):
    """[rows, Kp] g8-packed containers -> (hi + lo) / scale as fp64 [rows, K] (test helper: the inverse of the packing)."""
    rows, Kp = p.shape
    h = p.contiguous().view(torch.float16).view(rows, Kp // 8, 2, 8).double()        # [rows, group, hi|lo, 8]
    return ((h[:, :, 0] + h[:, :, 1]).reshape(rows, Kp) / scale.double()[:, None])[:, :K]


@pytest.mark.parametrize("M,K", [(300, 128), (256, 516), (1000, 1024), (77, 2752), (130, 6144), (64, 36)])
def test_g8_packing(ops, M, K):
    """pack_rows_g8 (static weights) and scale_pack_rows_g8 (activations, one pass): identical bits; the decoded value is the input to
    2^-21 relative (hi + lo = 22 significand bits); the K padding up to the 32-k slab is zero."""
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))).cuda()
    s = ops.row_scale_f16(x)
    p = ops.pack_rows_g8(x, s)
    Kp = (K + 31) // 32 * 32
    assert p.shape == (M, Kp)
    dec = _unpack_g8(p, s, Kp)
    assert (dec[:, K:] == 0).all()
    rel = ((dec[:, :K] - x.double()).abs() / x.double().abs().amax(1, keepdim=True)).max().item()
    assert rel < 2.0 ** -21, rel
    if K % 4 == 0:
        p2, s2 = ops.scale_pack_rows_g8(x)
        assert torch.equal(s2, s) and torch.equal(p2.view(torch.int32), p.view(torch.int32))


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 392, 516), (4096, 1024, 1024), (520, 640, 2752)])
def test_gemm_f16x3_packed_activations(ops, M, N, K):
    """An activation packed by its producer (x_packed=True) gives bit-identical results to packing inside linear()."""
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b, res = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    fw = ops.F16Weight(W)
    xp, sa = ops.scale_pack_rows_g8(x)
    with ops.gemm_mode("f16x3"):
        ref = ops.linear(x, fw, b, act=ops.ACT_GELU, residual=res)
        got = ops.linear(xp, fw, b, act=ops.ACT_GELU, residual=res, x_scale=sa, x_packed=True)
    assert torch.equal(ref, got)
    with pytest.raises(ValueError):
        with ops.gemm_mode("f32"):
            ops.linear(xp, fw, x_scale=sa, x_packed=True)


@pytest.mark.parametrize("M,N,K,act", [(512, 1408, 6144, 0), (512, 1408, 1408, 1), (1024, 1024, 2784, 0), (300, 260, 1024, 2)])
def test_gemm_f16x3_split_k(ops, M, N, K, act):
    """Few tiles and a long K loop (one cloud through a wide encoder: fc2 / proj of the giant ViT at M = 512): split-K with a fixed-order
    reduction.  fp32-grade against fp64 like the unsplit launch, bitwise reproducible, and chosen by the library for these shapes."""
    import ctypes
    from point_sam_amd import _lib
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b, res = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    fw = ops.F16Weight(W)
    ks = ops.splitk_factor(M, N, fw.Kp, act)
    assert ks > 1, "these shapes are the ones split-K exists for"
    with ops.gemm_mode("f16x3"):
        got = ops.linear(x, fw, b, act=act, residual=res)
        again = ops.linear(x, fw, b, act=act, residual=res)
    assert torch.equal(got, again)
    # the two forms of the reduction -- in-kernel fix-up by the last workgroup of a tile (default) and partial planes + a reduction launch -- add the
    # same partials in the same order: same bits, also in place (C == residual, the encoder's x += proj(..))
    L0 = _lib.load()
    try:
        with ops.gemm_mode("f16x3"):
            forms = {}
            for fx in (0, 1):
                L0.psam_gemm_f16x3p_force_splitk_fixup(fx)
                inpl = res.clone()
                ops.linear(x, fw, b, act=act, residual=inpl, out=inpl)
                forms[fx] = (ops.linear(x, fw, b, act=act, residual=res), inpl)
    finally:
        L0.psam_gemm_f16x3p_force_splitk_fixup(-1)
    assert torch.equal(forms[0][0], forms[1][0]) and torch.equal(forms[0][1], forms[1][1]) and torch.equal(forms[1][0], forms[1][1]) and torch.equal(forms[1][0], got)
    z = x.double() @ W.double().t() + b.double()
    want = (torch.nn.functional.gelu(z) if act == 1 else z.clamp_min(0) if act == 2 else z) + res.double()
    with ops.gemm_mode("f32"):
        f32 = ops.linear(x, W, b, act=act, residual=res)
    err, err32 = (got.double() - want).abs().max().item(), (f32.double() - want).abs().max().item()
    print(f"\n[split-K {M}x{N}x{K} ks={ks}] max err {err:.2e} (f32 kernel {err32:.2e})")
    assert err < 2 * err32 + 1e-6
    # the unsplit launch through the C ABI agrees to fp32 rounding of the partial sums; a workspace that is too small is refused
    xp, sa = ops.scale_pack_rows_g8(x)
    L = _lib.load()
    one = torch.empty(M, N, device="cuda")
    args = lambda out, fuse: (xp.data_ptr(), xp.stride(0), sa.data_ptr(), fw.packed.data_ptr(), fw.packed.stride(0), fw.scale.data_ptr(), out.data_ptr(), N,
                              b.data_ptr(), res.data_ptr(), N, None, 0, 0, M, N, fw.Kp, 1.0, act, fuse, None)
    assert L.psam_gemm_f16x3p_ex(*args(one, None)) == 0
    torch.cuda.synchronize()
    assert (one - got).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    fuse = _lib.GemmFuse()
    ws = torch.empty(ks * M * N - 4, device="cuda")
    fuse.splitk_ws, fuse.splitk_plane, fuse.splitk = ws.data_ptr(), M * N - 4, ks
    assert L.psam_gemm_f16x3p_ex(*args(one, ctypes.byref(fuse))) != 0


@pytest.mark.parametrize("cols,ld", [(1024, 1024), (2730, 2752), (300, 320), (4096, 4096)])
def test_layernorm_packed_output(ops, cols, ld):
    """LayerNorm with pack=True == LayerNorm followed by row_scale_f16 + pack_rows_g8 (bit for bit), also in place on a padded
    buffer (the SwiGLU output of the EVA02 blocks: 2730 of 2752 columns); the K padding is written as zeros."""
    g = torch.Generator().manual_seed(cols)
    buf = torch.full((300, ld), 7.0)      # non-zero padding: the kernel must overwrite it with zeros
    buf[:, :cols] = torch.randn(300, cols, generator=g) * torch.exp(torch.randn(300, 1, generator=g))
    w, b = torch.randn(cols, generator=g).cuda(), torch.randn(cols, generator=g).cuda()
    x = buf.cuda()
    y = ops.layernorm(x[:, :cols], w, b, 1e-6, out=torch.empty_like(x)[:, :cols])    # same (float4) kernel path as the packed run
    sa = ops.row_scale_f16(y)
    want = ops.pack_rows_g8(y, sa)                      # [300, Kp]
    Kp = want.shape[1]
    rs = torch.empty(300, device="cuda")
    xin = x.clone()
    ops.layernorm(xin[:, :cols], w, b, 1e-6, out=xin[:, :cols], scale_out=rs, pack=True)      # in place, padded
    assert torch.equal(rs, sa)
    got = xin[:, :Kp].contiguous().view(torch.int32)
    if cols % 4 == 0:
        assert torch.equal(got, want.view(torch.int32))
    else:   # the ragged last float4 is evaluated element-wise in the unpacked kernel (FMA contraction may differ by an ulp)
        g8 = (cols // 8) * 8
        assert torch.equal(got[:, :g8], want.view(torch.int32)[:, :g8])
        assert torch.equal(got[:, g8 + 8:], want.view(torch.int32)[:, g8 + 8:])
        dec = _unpack_g8(xin[:, :Kp], rs, Kp)
        assert torch.allclose(dec[:, g8:cols].float(), y[:, g8:], rtol=2e-6, atol=1e-7) and (dec[:, cols:] == 0).all()
    rs2 = torch.empty(300, device="cuda")
    out = ops.layernorm(x[:, :cols], w, b, 1e-6, scale_out=rs2, pack=True)                      # out of place: a [rows, Kp] buffer
    assert out.shape == (300, Kp) and torch.equal(out.view(torch.int32), got) and torch.equal(rs2, sa)


def test_gemm_f16x3_epilogues(ops):
    g = torch.Generator().manual_seed(3)
    M, D, H, Hp, grp = 384, 128, 170, 192, 64
    x = torch.randn(M, D, generator=g)
    Wg, Wx = torch.randn(H, D, generator=g) / D ** 0.5, torch.randn(H, D, generator=g) / D ** 0.5
    bg, bx = torch.randn(H, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    pad = lambda t: torch.cat([t, torch.zeros((Hp - H,) + tuple(t.shape[1:]))], 0)
    W1 = torch.stack([pad(Wg).view(Hp // 32, 32, D), pad(Wx).view(Hp // 32, 32, D)], 1).reshape(2 * Hp, D)
    b1 = torch.stack([pad(bg).view(Hp // 32, 32), pad(bx).view(Hp // 32, 32)], 1).reshape(2 * Hp)
    rb = torch.randn(M // grp, 2 * Hp, generator=g)
    fw = ops.F16Weight(cu(W1))
    L = ops._lib.load()
    want = F.silu(F.linear(x.double(), Wg.double(), bg.double())) * F.linear(x.double(), Wx.double(), bx.double())
    for cfg in F16X3P_CFGS:
        L.psam_gemm_f16x3p_force_config(cfg)
        try:
            with ops.gemm_mode("f16x3"):
                if cfg in (12, 23):     # three-tile-wide wave tiles cannot pair gate / value tiles: rejected, not mis-computed
                    with pytest.raises(ops._lib.PointSamHipError):
                        ops.linear(cu(x), fw, cu(b1), act=ops.ACT_SWIGLU)
                else:
                    u = ops.linear(cu(x), fw, cu(b1), act=ops.ACT_SWIGLU)
                    _close(u[:, :H], want, 1e-4, what=f"f16x3 swiglu epilogue cfg{cfg}")
                    assert (u[:, H:] == 0).all()
                y = ops.linear(cu(x), fw, None, act=ops.ACT_RELU, rowbias=cu(rb), rowgroup=grp)
        finally:
            L.psam_gemm_f16x3p_force_config(-1)
        _close(y, F.relu(x.double() @ W1.double().T + rb.double().repeat_interleave(grp, 0)), 1e-4, what=f"f16x3 rowbias+relu cfg{cfg}")


@pytest.mark.parametrize("M,D,H", [(512, 256, 300), (256, 1024, 2730), (1024, 384, 650)])
def test_fused_mlp_two_gemms(ops, M, D, H):
    """EVA02 MLP as two GEMMs and nothing in between (psam_gemm_fuse_t): fc1 emits the SwiGLU-gated rows g8-packed with bound-derived
    scales plus LayerNorm partials; fc2 runs with the inner LayerNorm folded in.  Against fp64 and against the unfused sequence
    (fc1 -> LayerNorm kernel -> fc2); the bound-derived scale never overflows fp16 and keeps the decoded rows fp32-grade."""
    g = torch.Generator().manual_seed(M + D + H)
    Hp = (H + 63) // 64 * 64
    h = torch.randn(M, D, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    Wg, Wx = torch.randn(H, D, generator=g) / D ** 0.5, torch.randn(H, D, generator=g) / D ** 0.5
    bg, bx = torch.randn(H, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    gam, bet = 1 + 0.2 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    W2, b2 = torch.randn(D, H, generator=g) / H ** 0.5, torch.randn(D, generator=g) * 0.1
    res = torch.randn(M, D, generator=g)
    eps = 1e-6
    u64 = F.silu(F.linear(h.double(), Wg.double(), bg.double())) * F.linear(h.double(), Wx.double(), bx.double())
    want = F.linear(F.layer_norm(u64, (H,), gam.double(), bet.double(), eps), W2.double(), b2.double()) + res.double()
    pad = lambda t: torch.cat([t, torch.zeros((Hp - H,) + tuple(t.shape[1:]))], 0)
    W1 = torch.stack([pad(Wg).view(Hp // 32, 32, D), pad(Wx).view(Hp // 32, 32, D)], 1).reshape(2 * Hp, D)
    b1 = torch.stack([pad(bg).view(Hp // 32, 32), pad(bx).view(Hp // 32, 32)], 1).reshape(2 * Hp)
    W2p = torch.zeros(D, Hp); W2p[:, :H] = W2
    gp = torch.zeros(Hp, dtype=torch.float64); gp[:H] = gam.double()
    w2g = W2p.double() * gp[None, :]
    ln_c = cu(w2g.sum(1).float())
    ln_d = cu((W2.double() @ bet.double() + b2.double()).float())
    k1 = float(2.0 ** 15 * math.sqrt(D) * W1.double().norm(dim=1).max())
    k2 = float(b1.abs().max())
    fw1, fw2, fw2g = ops.F16Weight(cu(W1)), ops.F16Weight(cu(W2p)), ops.F16Weight(cu(w2g.float()))
    assert ops.fuse_supported(M, 2 * Hp) and ops.fuse_supported(M, D)
    with ops.gemm_mode("f16x3"):
        hp, sh = ops.scale_pack_rows_g8(cu(h))
        # unfused: fc1 -> LayerNorm (packing) -> fc2
        u = ops.linear(hp, fw1, cu(b1), act=ops.ACT_SWIGLU, x_scale=sh, x_packed=True)
        rs = torch.empty(M, device="cuda")
        ops.layernorm(u[:, :H], cu(gam), cu(bet), eps, out=u[:, :H], scale_out=rs, pack=True)
        y0 = ops.linear(u, fw2, cu(b2), residual=cu(res), x_scale=rs, x_packed=True)
        # fused
        up = torch.empty(M, Hp, device="cuda"); su = torch.empty(M, device="cuda")
        st = torch.empty(M, ops.stat_segs(2 * Hp), 2, device="cuda")
        ops.linear(hp, fw1, cu(b1), act=ops.ACT_SWIGLU, x_scale=sh, x_packed=True, out=up, pack_out=(su, k1, k2), stats=(st, H))
        mean, rstd = ops.ln_stats_finalize(st, H, eps)
        y1 = ops.linear(up, fw2g, ln_d, residual=cu(res), x_scale=su, x_packed=True, ln_fold=(mean, rstd, ln_c))
    # the packed gated rows decode to u (fp32-grade), with power-of-two scales that keep every element below 2^15
    dec = _unpack_g8(up, su, Hp)
    assert (dec[:, H:] == 0).all()
    assert ((dec[:, :H].cpu() - u64).abs().max(1).values / u64.abs().max(1).values.clamp_min(1e-30)).max().item() < 1e-5
    assert (torch.log2(su) == torch.log2(su).round()).all() and (u64.abs().max(1).values * su.cpu().double() < 2.0 ** 15).all()
    want_rstd = 1.0 / torch.sqrt(u64.var(1, unbiased=False) + eps)
    assert ((mean.cpu().double() - u64.mean(1)).abs() * want_rstd).max().item() < 1e-5          # in units of the row's standard deviation
    assert ((rstd.cpu().double() - want_rstd).abs() / want_rstd).max().item() < 1e-5
    e0 = ((y0.cpu().double() - want).abs().max() / want.abs().max()).item()
    e1 = ((y1.cpu().double() - want).abs().max() / want.abs().max()).item()
    print(f"\n[fused MLP M={M} D={D} H={H}] rel err vs fp64: unfused {e0:.2e}, fused {e1:.2e}")
    assert e1 < 3e-6 and e1 < 4 * e0 + 5e-7, (e0, e1)


def test_interp3_bitwise_stable_beside_gemm_stream(ops):
    """interpolate_features at the benchmark's size (LayerNorm + GELU + packed output, the decoder's launch; and the plain form) must give the SAME
    BITS alone and while another stream launches GEMM workgroups.  Until round 4 the kernel multiplied by its second weight with
    `v_pk_mul_f32 .. op_sel:[1,0]`, one of the packed forms that return wrong lanes 48-63 beside such workgroups in the probe
    (point_sam_amd/isa_lint.py, scripts/exp/r04_pk_opsel.hip).  The pre-fix build also passes this test (the kernel's short-lived waves were never
    caught, profiles/r04/r04_hazard.txt) -- the ISA lint is what keeps the form out; this holds the line on the behaviour."""
    g = torch.Generator().manual_seed(0)
    Z, G, C, N = 8, 512, 256, 32768
    src = cu(torch.randn(Z, G, C, generator=g))
    idx3 = cu(torch.randint(0, G, (Z, N, 3), generator=g))
    w3 = torch.rand(Z, N, 3, generator=g) + 0.05
    w3 = cu(w3 / w3.sum(-1, keepdim=True))
    gam, bet = cu(torch.randn(C, generator=g)), cu(torch.randn(C, generator=g) * 0.1)
    M, D = 4096, 1024
    wq, bq = ops.F16Weight(cu(torch.randn(3 * D, D, generator=g) / 32)), cu(torch.zeros(3 * D))
    with ops.gemm_mode("f16x3"):
        hp, sh = ops.scale_pack_rows_g8(cu(torch.randn(M, D, generator=g)))
        qkv = torch.empty(M, 3 * D, device="cuda")

        def run():
            a = torch.empty(Z, N, C, device="cuda"); sa = torch.empty(Z * N, device="cuda"); b = torch.empty(Z, N, C, device="cuda")
            ops.interp3(src, idx3, w3, a, 1, scale_out=sa, ln=(gam, bet, 1e-6), act=ops.ACT_GELU)
            ops.interp3(src, idx3, w3, b, 1)
            return a, sa, b

        ref = run()
        torch.cuda.synchronize()
        s2 = torch.cuda.Stream()
        for it in range(6):
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                for _ in range(40):
                    ops.linear(hp, wq, bq, x_scale=sh, x_packed=True, out=qkv)
            out = run()
            torch.cuda.synchronize()
            for name, x, y in zip(("packed", "scales", "plain"), out, ref):
                nbad = int((x.view(torch.int32) != y.view(torch.int32)).sum())
                assert nbad == 0, f"run {it}: {nbad} words of the {name} output changed beside the GEMM stream"


def test_fused_mlp_bitwise_stable_beside_other_streams(ops):
    """The fused EVA02 MLP GEMMs (fc1: SwiGLU + row statistics + packed output; fc2: folded LayerNorm + residual) at the benchmark's
    size must give the SAME BITS whether they run alone or while another stream's kernels (GEMMs, attention, LayerNorm) share the CUs.
    (Round 3: a software-pipelined epilogue loop passed every single-stream test and produced 1e-2 errors in exactly this situation --
    two workgroups of different kernels per CU; only lanes 48-63 of one lane-shuffle result were stale.  scripts/exp/r03_race.py.)"""
    L = ops._lib.load()
    g = torch.Generator().manual_seed(0)
    M, D, H = 4096, 1024, 2730
    Hp = (H + 31) // 32 * 32
    h = torch.randn(M, D, generator=g)
    W1 = torch.randn(2 * Hp, D, generator=g) / 32
    b1 = torch.randn(2 * Hp, generator=g) * 0.1
    w2g = torch.randn(D, Hp, generator=g) / 52
    ln_c, ln_d, res = cu(w2g.sum(1)), cu(torch.randn(D, generator=g) * 0.1), cu(torch.randn(M, D, generator=g))
    k1, k2 = float(2.0 ** 15 * math.sqrt(D) * W1.double().norm(dim=1).max()), float(b1.abs().max())
    fw1, fw2g, b1 = ops.F16Weight(cu(W1)), ops.F16Weight(cu(w2g)), cu(b1)
    wq, bq = ops.F16Weight(cu(torch.randn(3 * D, D, generator=g) / 32)), cu(torch.zeros(3 * D))
    qkv = cu(torch.randn(M, 3 * D, generator=g))
    with ops.gemm_mode("f16x3"):
        hp, sh = ops.scale_pack_rows_g8(cu(h))

        def mlp():
            up = torch.empty(M, Hp, device="cuda"); su = torch.empty(M, device="cuda"); st = torch.empty(M, ops.stat_segs(2 * Hp), 2, device="cuda")
            ops.linear(hp, fw1, b1, act=ops.ACT_SWIGLU, x_scale=sh, x_packed=True, out=up, pack_out=(su, k1, k2), stats=(st, H))
            mean, rstd = ops.ln_stats_finalize(st, H, 1e-6)
            y = ops.linear(up, fw2g, ln_d, residual=res, x_scale=su, x_packed=True, ln_fold=(mean, rstd, ln_c))
            return dict(up=up, su=su, st=st[:, :(H + 31) // 32], mean=mean, rstd=rstd, y=y)

        def noise(kind):
            if kind == "qkv":
                ops.linear(hp, wq, bq, x_scale=sh, x_packed=True)
            elif kind == "mlp":
                mlp()
            elif kind == "ln":
                ops.layernorm(res, ln_d, ln_d, 1e-6)
            elif kind == "attn":
                o = torch.empty(M, D, device="cuda")
                ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, 8, 16, 512, 512, 64, 0.125)
            else:
                attn_packed()

        sq = torch.full((M,), 2.0 ** 11, device="cuda")
        qkvp = ops.pack_rows_g8(qkv, sq)

        def attn_packed():
            o = torch.empty(M, D, device="cuda"); so = torch.empty(M, device="cuda")
            ops.attention_packed(qkvp, sq, o, so, 8, 16, 512, 64, 0.125, 8.0)
            return o

        s2 = torch.cuda.Stream()
        try:
            ref_attn = attn_packed()
            for cfg in (-1, 21, 28):
                L.psam_gemm_f16x3p_force_config(cfg)
                ref = mlp()
                torch.cuda.synchronize()
                for kind in ("qkv", "mlp", "attn", "ln", "attn_packed"):
                    for it in range(4):
                        s2.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(s2):
                            for _ in range(6):
                                noise(kind)
                        out = mlp()
                        oa = attn_packed()
                        torch.cuda.synchronize()
                        for k in ref:
                            assert torch.equal(out[k].view(torch.int32), ref[k].view(torch.int32)), (cfg, kind, it, k)
                        assert torch.equal(oa.view(torch.int32), ref_attn.view(torch.int32)), (cfg, kind, it, "packed-operand attention")
        finally:
            L.psam_gemm_f16x3p_force_config(-1)


def test_gemm_f16x3_register_epilogue_bitwise(ops):
    """The register-only GEMM epilogue (csrc/gemm_epilogue_t.h: MFMA operands swapped, one output row per lane, no LDS transposition, no lane
    shuffles) must give the SAME BITS as the LDS-transposition epilogue (csrc/gemm_epilogue.h) for every fused option set of the encoder's
    GEMMs at the benchmark's size -- packed q|k|v, projection + residual, fc1 (SwiGLU gate + row statistics + packed output), fc2 with the
    folded LayerNorm -- and for ragged shapes (edge tiles); the hyper-network products (different summation order) agree to rounding and
    with fp64 (mask_decoder.py:171-176).  timm Eva block shapes as restated in oracle/pointsam_oracle.py (pc_encoder.py:138-139)."""
    L = ops._lib.load()
    g = torch.Generator().manual_seed(0)
    M, D, H = 4096, 1024, 2730
    Hp = (H + 31) // 32 * 32
    h = torch.randn(M, D, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))
    W1 = torch.randn(2 * Hp, D, generator=g) / 32
    b1 = cu(torch.randn(2 * Hp, generator=g) * 0.1)
    w2g = torch.randn(D, Hp, generator=g) / 52
    ln_c, ln_d, res = cu(w2g.sum(1)), cu(torch.randn(D, generator=g) * 0.1), cu(torch.randn(M, D, generator=g))
    k1, k2 = float(2.0 ** 15 * math.sqrt(D) * W1.double().norm(dim=1).max()), float(b1.abs().max())
    fw1, fw2g = ops.F16Weight(cu(W1)), ops.F16Weight(cu(w2g))
    wq, bq = ops.F16Weight(cu(torch.randn(3 * D, D, generator=g) / 32)), cu(torch.randn(3 * D, generator=g) * 0.1)
    wp = ops.F16Weight(cu(torch.randn(D, D, generator=g) / 32))
    xr, wr, br, rr = cu(torch.randn(777, 260, generator=g)), ops.F16Weight(cu(torch.randn(392, 260, generator=g) / 16)), cu(torch.randn(392, generator=g)), cu(torch.randn(777, 392, generator=g))
    rbias = cu(torch.randn(M // 64, D, generator=g))
    # upscaling MLP tail: Linear + GELU + hyper products over [Z * N, 256]
    Z, Npts, C, E = 2, 4096, 3, 256
    u1 = torch.randn(Z * Npts, E, generator=g)
    w3, b3, hyper = torch.randn(E, E, generator=g) / 16, torch.randn(E, generator=g) * 0.1, torch.randn(Z, C, E, generator=g)
    fw3 = ops.F16Weight(cu(w3))
    want_masks = torch.einsum("zce,zne->zcn", hyper.double(), F.gelu(u1.double() @ w3.double().T + b3.double()).view(Z, Npts, E))
    out = {}
    try:
        with ops.gemm_mode("f16x3"):
            hp, sh = ops.scale_pack_rows_g8(cu(h))
            u1p, s1 = ops.scale_pack_rows_g8(cu(u1))
            for ep in (0, 1):
                L.psam_gemm_f16x3p_force_epilogue(ep)
                up = torch.empty(M, Hp, device="cuda"); su = torch.empty(M, device="cuda"); st = torch.empty(M, ops.stat_segs(2 * Hp), 2, device="cuda")
                ops.linear(hp, fw1, b1, act=ops.ACT_SWIGLU, x_scale=sh, x_packed=True, out=up, pack_out=(su, k1, k2), stats=(st, H))
                o = dict(up=up, su=su, st=st[:, :(H + 31) // 32].contiguous())
                if ep == 0:
                    up0, su0 = up, su
                    mean0, rstd0 = ops.ln_stats_finalize(st, H, 1e-6)
                o["fc2"] = ops.linear(up0, fw2g, ln_d, residual=res, x_scale=su0, x_packed=True, ln_fold=(mean0, rstd0, ln_c))
                sq = torch.empty(M, device="cuda"); qo = torch.empty(M, 3 * D, device="cuda")
                ops.linear(hp, wq, bq, x_scale=sh, x_packed=True, out=qo, pack_out=(sq, 0.0, 50.0))
                o["qkv_packed"], o["qkv_scale"] = qo, sq
                o["proj"] = ops.linear(hp, wp, bq[:D].contiguous(), residual=res, x_scale=sh, x_packed=True)
                o["gelu"] = ops.linear(hp, wp, bq[:D].contiguous(), act=ops.ACT_GELU, x_scale=sh, x_packed=True)
                o["ragged_gelu_res"] = ops.linear(xr, wr, br, act=ops.ACT_GELU, residual=rr)
                o["ragged_plain"] = ops.linear(xr, wr, None)
                o["rowbias"] = ops.linear(hp, wp, None, rowbias=rbias, rowgroup=64, x_scale=sh, x_packed=True)      # PatchEncoder conv2.0: a bias row per group of 64 rows
                o["ragged_rowbias"] = ops.linear(xr, wr, br, rowbias=rr[:26].contiguous(), rowgroup=30)
                masks = torch.empty(Z, C, Npts, device="cuda")
                ops.linear(u1p, fw3, cu(b3), act=ops.ACT_GELU, x_scale=s1, x_packed=True, hyper=(cu(hyper), masks, Npts), no_store=True)
                o["masks"] = masks
                out[ep] = o
            torch.cuda.synchronize()
    finally:
        L.psam_gemm_f16x3p_force_epilogue(-1)
    for k in out[0]:
        a, b = out[0][k], out[1][k]
        if k == "masks":
            for t in (a, b):
                assert ((t.cpu().double() - want_masks).abs().max() / want_masks.abs().max()).item() < 2e-6
            assert ((a - b).abs().max() / a.abs().max()).item() < 2e-6
        else:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{k}: {(a.view(torch.int32) != b.view(torch.int32)).sum().item()} words differ"


@pytest.mark.parametrize("M,D,H", [(4096, 1024, 2730), (2048, 1024, 2730), (2048, 512, 4090), (8192, 1024, 2730)])
def test_gemm_f16x3_continuous(ops, M, D, H):
    """The persistent kernel of the batch-sized encoder GEMMs (csrc/experiments/gemm_f16x3c.hip: resident workgroups draw whole tiles from per-XCD queues and run one
    continuous stream of K slabs across tile boundaries -- the next tile's first slabs are in LDS before the finished tile's epilogue runs) against the
    one-workgroup-per-tile kernel on the encoder's fused GEMMs -- packed q|k|v, projection + residual, fc1 (SwiGLU gate + row statistics + packed
    output), fc2 with the folded LayerNorm, a GELU (timm Eva block shapes, pc_encoder.py:138-139): the SAME BITS (same products, same order per tile),
    from launch to launch and while another stream keeps the CUs busy (which workgroup draws which tile must not matter).  Shapes: 1.5 tiles per
    resident workgroup (4096 rows), fewer tiles than workgroups, an odd number of K slabs per tile (the K = 1056 launch: the ring parity flips at every
    tile boundary), several tiles per workgroup (8192 rows).  (Measured the same time as the one-workgroup-per-tile kernel -- the GEMM is power-bound,
    profiles/r05/r05_power_gemm.txt -- hence an experiments-build kernel.)"""
    if not ops._lib.has_experiments():
        pytest.skip("measured-and-not-adopted path: the library was built without PSAM_BUILD_EXPERIMENTS=1")
    L = ops._lib.load()
    g = torch.Generator().manual_seed(3)
    Hp = (H + 31) // 32 * 32
    h = torch.randn(M, D, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))
    W1, b1 = torch.randn(2 * Hp, D, generator=g) / D ** 0.5, cu(torch.randn(2 * Hp, generator=g) * 0.1)
    w2g = torch.randn(D, Hp, generator=g) / Hp ** 0.5
    ln_c, ln_d, res = cu(w2g.sum(1)), cu(torch.randn(D, generator=g) * 0.1), cu(torch.randn(M, D, generator=g))
    k1, k2 = float(2.0 ** 15 * math.sqrt(D) * W1.double().norm(dim=1).max()), float(b1.abs().max())
    fw1, fw2g = ops.F16Weight(cu(W1)), ops.F16Weight(cu(w2g))
    wq, bq = ops.F16Weight(cu(torch.randn(3 * D, D, generator=g) / D ** 0.5)), cu(torch.randn(3 * D, generator=g) * 0.1)
    wp = ops.F16Weight(cu(torch.randn(D, D, generator=g) / D ** 0.5))
    xo, wo = cu(torch.randn(M, 1056, generator=g)), ops.F16Weight(cu(torch.randn(1152, 1056, generator=g) / 32))      # 33 slabs per tile
    s2 = torch.cuda.Stream()

    def layer(hp, sh):
        up = torch.empty(M, Hp, device="cuda"); su = torch.empty(M, device="cuda"); st = torch.zeros(M, ops.stat_segs(2 * Hp), 2, device="cuda")
        ops.linear(hp, fw1, b1, act=ops.ACT_SWIGLU, x_scale=sh, x_packed=True, out=up, pack_out=(su, k1, k2), stats=(st, H))
        mean, rstd = ops.ln_stats_finalize(st, H, 1e-6)
        o = dict(up=up, su=su, st=st[:, :(H + 31) // 32].contiguous())
        o["odd_k"] = ops.linear(xo, wo, None)
        o["fc2"] = ops.linear(up, fw2g, ln_d, residual=res, x_scale=su, x_packed=True, ln_fold=(mean, rstd, ln_c))
        sq = torch.empty(M, device="cuda"); qo = torch.empty(M, 3 * D, device="cuda")
        ops.linear(hp, wq, bq, x_scale=sh, x_packed=True, out=qo, pack_out=(sq, 0.0, 50.0))
        o["qkv"], o["sq"] = qo, sq
        o["proj"] = ops.linear(hp, wp, bq[:D].contiguous(), residual=res, x_scale=sh, x_packed=True)
        o["gelu"] = ops.linear(hp, wp, bq[:D].contiguous(), act=ops.ACT_GELU, x_scale=sh, x_packed=True)
        o["plain"] = ops.linear(hp, wp, None, x_scale=sh, x_packed=True)
        return o

    out = {}
    try:
        with ops.gemm_mode("f16x3"):
            hp, sh = ops.scale_pack_rows_g8(cu(h))
            for mode in (0, 1):
                L.psam_gemm_f16x3p_force_continuous(mode)
                out[mode] = layer(hp, sh)
            torch.cuda.synchronize()
            for rep in range(6):      # launch to launch, with another stream's GEMMs on the chip
                s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    for _ in range(3):
                        ops.linear(hp, wq, bq, x_scale=sh, x_packed=True)
                again = layer(hp, sh)
                torch.cuda.current_stream().wait_stream(s2)
                torch.cuda.synchronize()
                for k, v in again.items():
                    assert torch.equal(v.view(torch.int32), out[1][k].view(torch.int32)), f"{k}: the persistent kernel's result differs between launches (repetition {rep})"
    finally:
        L.psam_gemm_f16x3p_force_continuous(-1)
    for k in out[0]:
        a, b = out[0][k], out[1][k]
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{k}: {(a.view(torch.int32) != b.view(torch.int32)).sum().item()} words differ from the one-workgroup-per-tile kernel"
    ref = F.gelu(h.double() @ wp.fp32.cpu().double().T + bq[:D].cpu().double())
    assert ((out[1]["gelu"].cpu().double() - ref).abs().max() / ref.abs().max()).item() < 3e-6


def test_gemm_row_ln_512(ops):
    """Linear (+ per-group row bias) -> LayerNorm -> GELU -> packed rows as ONE GEMM on full-row 128x512 tiles (register epilogue, two-pass row
    statistics across the row band's waves): PatchEncoder's conv2.0 / conv2.1 (common.py:493-496).  Against fp64; the packed rows decode to
    fp32-grade values under the a-priori LayerNorm bound.  (Off in the model by default: measured slower, profiles/r04/r04_rowln512.txt.)"""
    if not ops._lib.has_experiments():
        pytest.skip("measured-and-rejected path: the library was built without PSAM_BUILD_EXPERIMENTS=1")
    L = ops._lib.load()
    g = torch.Generator().manual_seed(5)
    M, N, K, grp = 1024, 512, 128, 64
    x = torch.randn(M, K, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))
    W = torch.randn(N, K, generator=g) / K ** 0.5
    rb = torch.randn(M // grp, N, generator=g)
    gam, bet = 1 + 0.3 * torch.randn(N, generator=g), 0.2 * torch.randn(N, generator=g)
    eps = 1e-6
    pre = x.double() @ W.double().T + rb.double().repeat_interleave(grp, 0)
    want = F.gelu(F.layer_norm(pre, (N,), gam.double(), bet.double(), eps))
    fw = ops.F16Weight(cu(W))
    bound = 1.001 * ops.row_ln_bound(cu(gam), cu(bet))
    L.psam_gemm_f16x3p_force_epilogue(1)
    try:
        with ops.gemm_mode("f16x3"):
            xp, sx = ops.scale_pack_rows_g8(cu(x))
            out = torch.empty(M, N, device="cuda"); so = torch.empty(M, device="cuda")
            ops.linear(xp, fw, None, act=ops.ACT_GELU, rowbias=cu(rb), rowgroup=grp, x_scale=sx, x_packed=True, out=out,
                       row_ln=(cu(gam), cu(bet), eps), pack_out=(so, 0.0, bound))
            plain = ops.linear(xp, fw, None, act=ops.ACT_GELU, rowbias=cu(rb), rowgroup=grp, x_scale=sx, x_packed=True, row_ln=(cu(gam), cu(bet), eps))
    finally:
        L.psam_gemm_f16x3p_force_epilogue(-1)
    dec = _unpack_g8(out, so, N).cpu().double()
    assert (torch.log2(so) == torch.log2(so).round()).all() and (want.abs().max(1).values * so.cpu().double() < 2.0 ** 15).all()
    e_pack = ((dec - want).abs().max() / want.abs().max()).item()
    e_plain = ((plain.cpu().double() - want).abs().max() / want.abs().max()).item()
    print(f"\n[row LN 512] rel err vs fp64: packed {e_pack:.2e}, fp32 {e_plain:.2e}")
    assert e_pack < 3e-6 and e_plain < 3e-6


def test_gemm_row_epilogues_upscaling_chain(ops):
    """The decoder's upscaling MLP inside GEMM epilogues (N = 256: a wave owns whole rows): Linear -> LayerNorm -> GELU with the result
    packed against the LayerNorm's bound, then Linear -> GELU -> hyper-network dot products, against fp64 (mask_decoder.py:53-59,164-176)."""
    g = torch.Generator().manual_seed(21)
    Z, Npts, E, C = 2, 640, 256, 3
    M = Z * Npts
    x = torch.randn(M, E, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))
    W0, b0 = torch.randn(E, E, generator=g) / 16, torch.randn(E, generator=g) * 0.1
    gam, bet = 1 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    W3, b3 = torch.randn(E, E, generator=g) / 16, torch.randn(E, generator=g) * 0.1
    hyper = torch.randn(Z, C, E, generator=g)
    u1 = F.gelu(F.layer_norm(F.linear(x.double(), W0.double(), b0.double()), (E,), gam.double(), bet.double(), 1e-5))
    u2 = F.gelu(F.linear(u1, W3.double(), b3.double()))
    want = torch.einsum("zce,zne->zcn", hyper.double(), u2.view(Z, Npts, E))
    fw0, fw3 = ops.F16Weight(cu(W0)), ops.F16Weight(cu(W3))
    bound = ops.row_ln_bound(gam, bet)
    with ops.gemm_mode("f16x3"):
        xp, sx = ops.scale_pack_rows_g8(cu(x))
        u1p = torch.empty(M, E, device="cuda"); s1 = torch.empty(M, device="cuda")
        ops.linear(xp, fw0, cu(b0), act=ops.ACT_GELU, x_scale=sx, x_packed=True, out=u1p, pack_out=(s1, 0.0, bound),
                   row_ln=(cu(gam), cu(bet), 1e-5))
        masks = torch.empty(Z, C, Npts, device="cuda")
        ops.linear(u1p, fw3, cu(b3), act=ops.ACT_GELU, x_scale=s1, x_packed=True, hyper=(cu(hyper), masks, Npts), no_store=True)
        full = ops.linear(u1p, fw3, cu(b3), act=ops.ACT_GELU, x_scale=s1, x_packed=True)      # the same GEMM with its output stored
    dec = _unpack_g8(u1p, s1, E).cpu()
    assert ((dec - u1).abs().max() / u1.abs().max()).item() < 2e-6
    m = u1.abs().max(1).values * s1.cpu().double()
    assert (s1 == s1[0]).all() and 2.0 ** 14 <= bound * s1[0].item() < 2.0 ** 15 and (m < 2.0 ** 15).all(), "scale from the LayerNorm bound"
    _close(full, u2, 2e-5, what="second GEMM")
    err = ((masks.cpu().double() - want).abs().max() / want.abs().max()).item()
    assert err < 2e-6, err


def test_gemm_bf16x6_epilogues(ops):
    g = torch.Generator().manual_seed(3)
    M, D, H, Hp, grp = 384, 128, 170, 192, 64
    x = torch.randn(M, D, generator=g)
    Wg, Wx = torch.randn(H, D, generator=g) / D ** 0.5, torch.randn(H, D, generator=g) / D ** 0.5
    bg, bx = torch.randn(H, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    pad = lambda t: torch.cat([t, torch.zeros((Hp - H,) + tuple(t.shape[1:]))], 0)
    W1 = torch.stack([pad(Wg).view(Hp // 32, 32, D), pad(Wx).view(Hp // 32, 32, D)], 1).reshape(2 * Hp, D)
    b1 = torch.stack([pad(bg).view(Hp // 32, 32), pad(bx).view(Hp // 32, 32)], 1).reshape(2 * Hp)
    rb = torch.randn(M // grp, 2 * Hp, generator=g)
    with ops.gemm_mode("bf16x6"):
        u = ops.linear(cu(x), cu(W1), cu(b1), act=ops.ACT_SWIGLU)
        y = ops.linear(cu(x), cu(W1), None, act=ops.ACT_RELU, rowbias=cu(rb), rowgroup=grp)
    want = F.silu(F.linear(x.double(), Wg.double(), bg.double())) * F.linear(x.double(), Wx.double(), bx.double())
    _close(u[:, :H], want, 1e-4, what="bf16x6 swiglu epilogue")
    assert (u[:, H:] == 0).all()
    _close(y, F.relu(x.double() @ W1.double().T + rb.double().repeat_interleave(grp, 0)), 1e-4, what="bf16x6 rowbias+relu")


def test_gemm_swiglu_epilogue(ops):
    """fc1 with the SiLU gate in the epilogue: packed weight = alternating 32-row blocks of fc1_g / fc1_x."""
    g = torch.Generator().manual_seed(11)
    M, D, H = 200, 96, 170
    Hp = 192
    x = torch.randn(M, D, generator=g)
    Wg, Wx = torch.randn(H, D, generator=g) / D ** 0.5, torch.randn(H, D, generator=g) / D ** 0.5
    bg, bx = torch.randn(H, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    pad = lambda t: torch.cat([t, torch.zeros((Hp - H,) + tuple(t.shape[1:]))], 0)
    W1 = torch.stack([pad(Wg).view(Hp // 32, 32, D), pad(Wx).view(Hp // 32, 32, D)], 1).reshape(2 * Hp, D)
    b1 = torch.stack([pad(bg).view(Hp // 32, 32), pad(bx).view(Hp // 32, 32)], 1).reshape(2 * Hp)
    for cfg in (0, 1, -1):
        ops._lib.load().psam_gemm_force_config(cfg)
        try:
            u = ops.linear(cu(x), cu(W1), cu(b1), act=ops.ACT_SWIGLU)
        finally:
            ops._lib.load().psam_gemm_force_config(-1)
        assert u.shape == (M, Hp)
        want = F.silu(F.linear(x.double(), Wg.double(), bg.double())) * F.linear(x.double(), Wx.double(), bx.double())
        _close(u[:, :H], want, 1e-4, what=f"swiglu epilogue cfg{cfg}")
        assert (u[:, H:] == 0).all()


@pytest.mark.parametrize("cols", [64, 128, 256, 512, 1000, 1024, 1408, 2730])
def test_layernorm(ops, cols):
    g = torch.Generator().manual_seed(cols)
    x, r = torch.randn(37, cols, generator=g) * 3 + 1, torch.randn(37, cols, generator=g)
    w, b = 1 + 0.1 * torch.randn(cols, generator=g), 0.1 * torch.randn(cols, generator=g)
    _close(ops.layernorm(cu(x), cu(w), cu(b), 1e-6), F.layer_norm(x.double(), (cols,), w.double(), b.double(), 1e-6), 2e-5, what="LN")
    got = ops.layernorm(cu(x), cu(w), cu(b), 1e-5, act=ops.ACT_GELU, residual=cu(r))
    _close(got, F.gelu(F.layer_norm((x + r).double(), (cols,), w.double(), b.double(), 1e-5)), 2e-5, what="LN+res+gelu")


def test_swiglu_ln(ops):
    g = torch.Generator().manual_seed(2)
    H, Hp, M = 170, 192, 50
    gx = torch.randn(M, 2 * Hp, generator=g)
    w, b = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    out = torch.full((M, Hp), 7.0, device="cuda")
    ops.swiglu_ln(cu(gx), Hp, H, cu(w), cu(b), 1e-6, out)
    want = F.layer_norm((F.silu(gx[:, :H]) * gx[:, Hp:Hp + H]).double(), (H,), w.double(), b.double(), 1e-6)
    _close(out[:, :H], want, 2e-5, what="swiglu_ln")
    assert (out[:, H:] == 0).all()


def test_group_max_pos_fourier_addbcast_interp(ops):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(40 * 16, 96, generator=g)
    assert torch.equal(ops.group_max(cu(x), 16).cpu(), x.view(40, 16, 96).max(1).values)
    c = torch.rand(77, 3, generator=g) * 2 - 1
    W, b = torch.randn(128, 3, generator=g), torch.randn(128, generator=g)
    _close(ops.pos_l1(cu(c), cu(W), cu(b)), F.gelu(F.linear(c.double(), W.double(), b.double())), 1e-5, what="pos_l1")
    # fourier PE + label embeddings written into a token buffer at an offset
    Gm = torch.randn(3, 128, generator=g)
    e0, e1 = torch.randn(1, 256, generator=g), torch.randn(1, 256, generator=g)
    Z, P, T = 5, 3, 8
    pts = torch.rand(Z, P, 3, generator=g) * 2 - 1
    lab = torch.randint(0, 2, (Z, P), generator=g)
    sd = {"point_encoder.pe_layer.positional_encoding_gaussian_matrix": Gm, "point_encoder.point_embeddings.0.weight": e0,
          "point_encoder.point_embeddings.1.weight": e1}
    tokens = torch.zeros(Z, T, 256, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.fourier_pe(cu(pts), cu(Gm), tokens.view(-1)[5 * 256:], P, T * 256, labels=cu(lab), emb0=cu(e0), emb1=cu(e1), flag=flag)
    _close(tokens[:, 5:], O.point_encoder(sd, pts, lab), 3e-5, what="point encoder")
    assert (tokens[:, :5] == 0).all() and flag.item() == 0
    pe = torch.empty(Z * P, 256, device="cuda")
    bad = pts.clone(); bad[2, 1, 0] = 1.5
    ops.fourier_pe(cu(bad), cu(Gm), pe, Z * P, 0, flag=flag)
    assert flag.item() == 1
    # add_bcast: repeat + add, and row-vector broadcast
    a, bb = torch.randn(2, 6, 256, generator=g), torch.randn(6, 6, 256, generator=g)
    out = torch.empty(6, 6, 256, device="cuda")
    ops.add_bcast(cu(a), 3, cu(bb), out, 6, 6, 256)
    assert torch.equal(out.cpu(), a.repeat_interleave(3, 0) + bb)
    ops.add_bcast(cu(a), 3, cu(e0), out, 6, 6, 256, sb=0, ldb=0)
    assert torch.equal(out.cpu(), a.repeat_interleave(3, 0) + e0)
    # interpolation
    src = torch.randn(4, 20, 256, generator=g)
    idx = torch.randint(0, 20, (2, 300, 3), generator=g)
    w3 = torch.rand(2, 300, 3, generator=g)
    out = torch.empty(4, 300, 256, device="cuda")
    ops.interp3(cu(src), cu(idx), cu(w3), out, 2)
    want = O.interpolate(src, idx.repeat_interleave(2, 0), w3.repeat_interleave(2, 0))
    _close(out, want, 1e-5, what="interp3")
    # ... with LayerNorm + GELU of the interpolated row (the upscaling MLP's `interp -> Linear -> LN -> GELU`, Linear moved in front)
    gam, bet = 1 + 0.2 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    want_ln = F.gelu(F.layer_norm(want.double(), (256,), gam.double(), bet.double(), 1e-5))
    ops.interp3(cu(src), cu(idx), cu(w3), out, 2, ln=(cu(gam), cu(bet), 1e-5), act=ops.ACT_GELU)
    _close(out, want_ln, 1e-5, what="interp3 + LN + GELU")
    sc = torch.empty(4 * 300, device="cuda")
    ops.interp3(cu(src), cu(idx), cu(w3), out, 2, scale_out=sc, ln=(cu(gam), cu(bet), 1e-5), act=ops.ACT_GELU)
    dec = _unpack_g8(out.view(-1, 256), sc, 256).cpu()
    assert ((dec - want_ln.view(-1, 256)).abs().max() / want_ln.abs().max()).item() < 2e-6, "packed interp3 + LN + GELU"


def test_upscaling_linear_commutes_with_interpolation(ops):
    """Linear(interp(x)) == interp(Linear(x)) to fp32 rounding for the real 3-NN weights (they sum to 1 up to 1 ulp): the identity the
    decoder's upscaling path relies on (mask_decoder.py:53-59,163)."""
    g = torch.Generator().manual_seed(77)
    xyz, _ = _cloud(1, 4096, seed=9)
    centers = xyz[:, torch.randperm(4096, generator=g)[:64]]
    idx, w3 = ops.three_nn(cu(xyz), cu(centers))
    assert ((w3.sum(-1) - 1).abs() < 3e-7).all()
    src = torch.randn(1, 64, 256, generator=g)
    W, b = torch.randn(256, 256, generator=g) / 16, torch.randn(256, generator=g)
    a = torch.empty(1, 4096, 256, device="cuda"); bb = torch.empty(1, 4096, 256, device="cuda")
    with ops.gemm_mode("f32"):
        ops.interp3(cu(src), idx, w3, a, 1)
        y0 = ops.linear(a.view(-1, 256), cu(W), cu(b))
        k1 = ops.linear(cu(src).view(-1, 256), cu(W), cu(b))
        ops.interp3(k1.view(1, 64, 256), idx, w3, bb, 1)
    err = (y0 - bb.view(-1, 256)).abs().max().item() / y0.abs().max().item()
    print(f"\n[Linear o interp vs interp o Linear] max rel diff {err:.2e}")
    assert err < 2e-6


def _sdpa(q, k, v, H, scale):
    B, Lq, D = q.shape
    sp = lambda t: t.reshape(B, t.shape[1], H, D // H).transpose(1, 2).double()
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, -1) @ sp(v)
    return a.transpose(1, 2).reshape(B, Lq, D)


@pytest.mark.parametrize("hd,H,Lq,Lk", [(64, 2, 128, 128), (64, 3, 512, 512), (32, 2, 32, 32), (24, 4, 100, 100), (88, 2, 200, 200),
                                        (16, 2, 70, 333), (64, 1, 5, 64), (48, 1, 129, 65), (128, 1, 64, 96), (96, 1, 33, 31),
                                        (64, 2, 2048, 2048), (88, 1, 2048, 2048)])
def test_flash_attention(ops, hd, H, Lq, Lk):
    g = torch.Generator().manual_seed(hd + Lq)
    B, D = 2, H * hd
    qkv_q = torch.randn(B, Lq, 3 * D, generator=g)          # fused-buffer layout: q | k | v in one row
    qkv_k = torch.randn(B, Lk, 3 * D, generator=g)
    bq, bk = cu(qkv_q).view(B * Lq, 3 * D), cu(qkv_k).view(B * Lk, 3 * D)
    out = torch.empty(B * Lq, D, device="cuda")
    scale = hd ** -0.5 * 2.0  # larger logits exercise the running-max path
    ops.attention(bq[:, :D], bk[:, D:2 * D], bk[:, 2 * D:], out, B, H, Lq, Lk, hd, scale)
    want = _sdpa(qkv_q[..., :D], qkv_k[..., D:2 * D], qkv_k[..., 2 * D:], H, scale)
    _close(out.view(B, Lq, D), want, 2e-5, what=f"flash hd={hd}")


@pytest.mark.parametrize("hd,H,Lq,Lk", [(64, 4, 512, 512), (64, 2, 130, 70), (64, 3, 33, 200), (128, 2, 200, 129), (64, 1, 5, 64),
                                        (64, 2, 2048, 2048), (64, 1, 1000, 2048), (88, 3, 512, 512), (88, 2, 77, 130), (96, 1, 100, 64)])
def test_flash_attention_f16x3(ops, hd, H, Lq, Lk):
    """fp16-split flash attention: same tolerance as the f32-MFMA kernel against an fp64 SDPA; q/k/v with very different
    magnitudes per tensor and per row (the scales are per query row / per 64-key tile)."""
    g = torch.Generator().manual_seed(hd + Lq)
    B, D = 2, H * hd
    q = torch.randn(B, Lq, D, generator=g) * torch.exp(torch.randn(B, Lq, 1, generator=g)) * 37.0
    k = torch.randn(B, Lk, D, generator=g) * 1e-3
    v = torch.randn(B, Lk, D, generator=g) * torch.exp(2 * torch.randn(B, Lk, 1, generator=g)) * 1e4
    scale = hd ** -0.5 * 40.0      # logits of a few units
    out = torch.empty(B * Lq, D, device="cuda")
    with ops.gemm_mode("f16x3"):
        ops.attention(cu(q).view(-1, D), cu(k).view(-1, D), cu(v).view(-1, D), out, B, H, Lq, Lk, hd, scale)
    want = _sdpa(q, k, v, H, scale)
    err = (out.view(B, Lq, D).cpu().double() - want).abs().max().item() / want.abs().max().item()
    out32 = torch.empty_like(out)
    ops.attention(cu(q).view(-1, D), cu(k).view(-1, D), cu(v).view(-1, D), out32, B, H, Lq, Lk, hd, scale)
    err32 = (out32.view(B, Lq, D).cpu().double() - want).abs().max().item() / want.abs().max().item()
    print(f"\n[flash f16x3 hd={hd} Lq={Lq} Lk={Lk}] rel err {err:.2e} (f32 kernel {err32:.2e})")
    assert err < 2e-6 and err < 4 * err32 + 2e-7, (err, err32)


@pytest.mark.parametrize("hd,H,L", [(64, 4, 512), (64, 2, 130), (128, 2, 200), (88, 4, 512), (88, 8, 100)])
def test_flash_attention_f16x3_packed_output(ops, hd, H, L):
    """Packed output of the fp16-split attention (for the output projection): decodes to the fp32 output's hi + lo (22 bits), one
    power-of-two scale per cloud derived from the BOUND of |V|, never overflowing fp16."""
    g = torch.Generator().manual_seed(hd + L)
    B, D = 3, H * hd
    qkv = (torch.randn(B * L, 3 * D, generator=g) * torch.exp(torch.randn(B * L, 1, generator=g))).cuda()
    a_scale = torch.exp2(torch.randint(-3, 12, (B * L,), generator=g).float()).cuda()     # stand-in for the LayerNorm row scales
    vmax = qkv[:, 2 * D:].abs().view(B, L * D).max(1).values
    smin = a_scale.view(B, L).min(1).values
    k2 = 0.25
    k1 = float(((vmax - k2).clamp_min(0) * smin).max().item() * 1.5 + 1.0)               # any k1 with k1 / smin + k2 >= max|V| per cloud
    assert (k1 / smin + k2 >= vmax).all()
    with ops.gemm_mode("f16x3"):
        want = torch.empty(B * L, D, device="cuda")
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], want, B, H, L, L, hd, hd ** -0.5)
        got = torch.empty(B * L, D, device="cuda"); so = torch.empty(B * L, device="cuda")
        ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], got, B, H, L, L, hd, hd ** -0.5, pack=(a_scale, k1, k2, so))
    assert (torch.log2(so) == torch.log2(so).round()).all()
    assert torch.equal(so.view(B, L), so.view(B, L)[:, :1].expand(B, L)), "one scale per cloud"
    assert (want.abs().view(B, L * D).max(1).values * so.view(B, L)[:, 0] < 2.0 ** 15).all()
    # identical bits to packing the fp32 output with the same scales
    ref = ops.pack_rows_g8(want, so)
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))


@pytest.mark.parametrize("H,L,B,spike", [(4, 512, 2, False), (2, 130, 3, False), (1, 64, 1, False), (3, 1000, 1, False), (2, 2048, 1, False), (1, 256, 1, True),
                                          (16, 512, 8, False)])
def test_attention_packed_operands(ops, H, L, B, spike):
    """Attention straight from g8-packed q | k | v rows with one common power-of-two scale (what the qkv GEMM's packing epilogue writes):
    K / V tiles by LDS-DMA, V transposed on read.  Against an fp64 SDPA of the same values, at the tolerance of the other attention
    kernels; ragged sequence lengths; a key that dominates late in the sequence (forces the online-softmax rescale); the scale taken from a
    bound 8x and 1000x above the largest value (a loose a-priori bound must not cost accuracy); output = the packing of the fp32 result."""
    hd, D = 64, H * 64
    g = torch.Generator().manual_seed(H * 1000 + L)
    qkv = torch.randn(B * L, 3 * D, generator=g)
    qkv[:, :D] *= 1.7; qkv[:, D:2 * D] *= 0.6; qkv[:, 2 * D:] *= torch.exp(torch.randn(B * L, 1, generator=g))
    if spike:
        qkv[200, D:2 * D] = qkv[7, :D] * 2.0
    scale = hd ** -0.5
    want = _sdpa(qkv[:, :D].view(B, L, D), qkv[:, D:2 * D].view(B, L, D), qkv[:, 2 * D:].view(B, L, D), H, scale)
    x = cu(qkv)
    vmax = float(qkv[:, 2 * D:].abs().max())
    for slack in (8.0, 1000.0):
        bound = float(qkv.abs().max()) * slack
        e = math.floor(math.log2(bound))
        sc = torch.full((B * L,), 2.0 ** (14 - e), device="cuda")
        assert float(qkv.abs().max()) * float(sc[0]) < 2.0 ** 15
        xp = ops.pack_rows_g8(x, sc)
        out = torch.empty(B * L, D, device="cuda"); so = torch.empty(B * L, device="cuda")
        with ops.gemm_mode("f16x3"):
            ops.attention_packed(xp, sc, out, so, B, H, L, hd, scale, vmax * 1.01)
        assert (so == so[0]).all() and float(torch.log2(so[0])) == round(float(torch.log2(so[0]))) and vmax * float(so[0]) < 2.0 ** 15
        got = _unpack_g8(out, so, D).cpu().view(B, L, D)
        err = (got - want).abs().max().item() / want.abs().max().item()
        print(f"\n[packed-operand attention H={H} L={L} B={B} bound x{slack:g}] rel err {err:.2e}")
        assert err < 2e-6, err


@pytest.mark.parametrize("hd,H,L,B", [(88, 16, 512, 1), (128, 4, 300, 1), (88, 4, 1000, 2)])      # H * hd % 32 == 0: the packed rows have no padding
def test_flash_attention_f16x3_key_split(ops, hd, H, L, B):
    """One cloud through the giant encoder's attention (16 heads x 4 query blocks = 64 workgroups on 256 CUs): the key tiles are split over up to
    four workgroups per (query block, head) and the last arrival combines the partial softmax states in split order inside the kernel.  Same
    accuracy against an fp64 SDPA as the unsplit launch, bitwise repeatable, packed output = packing of the fp32 output; a key that dominates late
    and V magnitudes that change by 2^10 between the splits (each split carries its own running maximum and V-tile scale)."""
    L0 = ops._lib.load()
    g = torch.Generator().manual_seed(hd * 7 + L)
    D = H * hd
    q = torch.randn(B, L, D, generator=g) * torch.exp(torch.randn(B, L, 1, generator=g))
    k = torch.randn(B, L, D, generator=g)
    v = torch.randn(B, L, D, generator=g)
    k[0, L - 30, :hd] = q[0, 7, :hd] * 3.0
    v[:, L // 2:] *= 1000.0
    scale = hd ** -0.5
    want = _sdpa(q, k, v, H, scale)
    qd, kd, vd = (cu(t).view(-1, D) for t in (q, k, v))
    res = {}
    try:
        with ops.gemm_mode("f16x3"):
            for mode in (0, 1):
                L0.psam_attention_f16x3_force_keysplit(mode)
                outs = []
                for rep in range(3):
                    o = torch.empty(B * L, D, device="cuda")
                    ops.attention(qd, kd, vd, o, B, H, L, L, hd, scale)
                    outs.append(o)
                torch.cuda.synchronize()
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), f"keysplit {mode}: not repeatable"
                res[mode] = outs[0]
            # packed output under the split
            a_scale = torch.ones(B * L, device="cuda")
            k1 = float(v.abs().max()) * 1.1
            got = torch.empty(B * L, D, device="cuda"); so = torch.empty(B * L, device="cuda")
            ops.attention(qd, kd, vd, got, B, H, L, L, hd, scale, pack=(a_scale, k1, 0.0, so))
            assert torch.equal(got.view(torch.int32), ops.pack_rows_g8(res[1], so).view(torch.int32))
    finally:
        L0.psam_attention_f16x3_force_keysplit(-1)
    e0 = (res[0].view(B, L, D).cpu().double() - want).abs().max().item() / want.abs().max().item()
    e1 = (res[1].view(B, L, D).cpu().double() - want).abs().max().item() / want.abs().max().item()
    print(f"\n[flash f16x3 key split hd={hd} H={H} L={L} B={B}] rel err unsplit {e0:.2e} split {e1:.2e}; differ {not torch.equal(res[0], res[1])}")
    assert e1 < 2e-6 and e1 < 3 * e0 + 2e-7
    if B * H * ((L + 127) // 128) * 2 <= 256 and L >= 256:
        assert not torch.equal(res[0], res[1]), "the split path did not run"


def test_flash_attention_f16x3_spike(ops):
    g = torch.Generator().manual_seed(9)
    B, H, hd, L = 1, 1, 64, 256
    q, k, v = (torch.randn(B, L, hd, generator=g) for _ in range(3))
    k[0, 200] = q[0, 7] * 4.0
    v[0, 130:] *= 1000.0           # V tile scales change by 2^10 mid-sequence
    out = torch.empty(B * L, hd, device="cuda")
    with ops.gemm_mode("f16x3"):
        ops.attention(cu(q).view(L, hd), cu(k).view(L, hd), cu(v).view(L, hd), out, B, H, L, L, hd, 1.0)
    want = _sdpa(q, k, v, H, 1.0)
    assert ((out.view(B, L, hd).cpu().double() - want).abs().max() / want.abs().max()).item() < 2e-6


def test_flash_attention_running_max_spike(ops):
    """Forces the rescale branch: one key dominates late in the sequence."""
    g = torch.Generator().manual_seed(9)
    B, H, hd, L = 1, 1, 64, 256
    q, k, v = (torch.randn(B, L, hd, generator=g) for _ in range(3))
    k[0, 200] = q[0, 7] * 4.0
    out = torch.empty(B * L, hd, device="cuda")
    ops.attention(cu(q).view(L, hd), cu(k).view(L, hd), cu(v).view(L, hd), out, B, H, L, L, hd, 1.0)
    _close(out.view(B, L, hd), _sdpa(q, k, v, H, 1.0), 2e-5, what="flash spike")


@pytest.mark.parametrize("M,N,K,act,res", [(56, 256, 256, 0, True), (56, 2048, 256, 2, False), (56, 256, 2048, 0, True), (7, 128, 256, 0, False),
                                           (64, 100, 48, 1, True), (1, 16, 16, 0, False), (33, 40, 272, 2, False)])
def test_linear_skinny(ops, M, N, K, act, res):
    """nn.Linear on <= 64 rows (the decoder's token side) through linear_skinny_kernel, vs fp64; ops.linear routes there by itself."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    want = F.linear(x.double(), W.double(), b.double())
    want = F.gelu(want) if act == 1 else (F.relu(want) if act == 2 else want)
    if res:
        want = want + r.double()
    with ops.gemm_mode("f16x3"):
        got = ops.linear(cu(x), cu(W), cu(b), act=act, residual=None if r is None else cu(r))
    _close(got, want, 1e-5, rtol=1e-5, what="linear_skinny")      # fp32 accumulation over up to 2048 terms
    # strided views (a column slice of a wider buffer, as the fused q|k|v buffers are used)
    big = torch.randn(M, K + 32, generator=g)
    with ops.gemm_mode("f32"):
        got2 = ops.linear(cu(big)[:, 16:16 + K], cu(W), cu(b))
    _close(got2, F.linear(big[:, 16:16 + K].double(), W.double(), b.double()), 1e-5, rtol=1e-5, what="linear_skinny strided")


@pytest.mark.parametrize("Z,M,din,dh,dout,T", [(8, 3, 256, 256, 256, 7), (5, 1, 256, 256, 4, 7), (2, 2, 64, 96, 10, 3)])
def test_mlp3_hypernetworks(ops, Z, M, din, dh, dout, T):
    """psam_mlp3 (the decoder's hyper-network MLPs / IoU head in one launch) vs fp64: MLP m reads token 1 + m of every prompt's row block."""
    g = torch.Generator().manual_seed(Z * 100 + M)
    hs = torch.randn(Z, T, din, generator=g)
    layers = [[(torch.randn(o, i, generator=g) / i ** 0.5, torch.randn(o, generator=g) * 0.1) for i, o in ((din, dh), (dh, dh), (dh, dout))] for _ in range(M)]
    mw = ops.Mlp3Weights([[(cu(W), cu(b)) for W, b in mlp] for mlp in layers])
    hsd = cu(hs)
    out = torch.full((Z, M, dout), float("nan"), device="cuda")
    ops.mlp3(hsd[:, 1, :], T * din, din, mw, out, M * dout, dout, Z)
    for m, mlp in enumerate(layers):
        x = hs[:, 1 + m, :].double()
        for li, (W, b) in enumerate(mlp):
            x = F.linear(x, W.double(), b.double())
            if li < 2:
                x = F.relu(x)
        _close(out[:, m, :], x, 2e-6, what=f"mlp3 m={m}")


@pytest.mark.parametrize("hd,H,Lq,Lk", [(32, 8, 7, 7), (16, 8, 7, 512), (16, 8, 512, 7), (32, 4, 300, 12), (16, 8, 6, 2048), (24, 2, 3, 70)])
def test_attention_small(ops, hd, H, Lq, Lk):
    g = torch.Generator().manual_seed(hd * Lk)
    Z, D = 3, H * hd
    q, k, v = torch.randn(Z, Lq, D, generator=g), torch.randn(Z, Lk, D, generator=g), torch.randn(Z, Lk, D, generator=g)
    out = torch.empty(Z * Lq, D, device="cuda")
    ops.attention_small(cu(q).view(-1, D), cu(k).view(-1, D), cu(v).view(-1, D), out, Z, H, Lq, Lk, hd, hd ** -0.5)
    _close(out.view(Z, Lq, D), _sdpa(q, k, v, H, hd ** -0.5), 2e-5, what="attention_small")


@pytest.mark.parametrize("hd,H,Lq,Lk,Z", [(16, 8, 9, 512, 1), (16, 8, 7, 2048, 2), (32, 8, 10, 130, 1), (16, 8, 6, 128, 3), (64, 2, 5, 515, 1)])
def test_attention_small_split_keys(ops, hd, H, Lq, Lk, Z):
    """Few queries against many keys (the decoder's tokens -> patches attention, transformer.py:160-166): one workgroup per query with the keys split
    over its four waves (attention_small_split_kernel) against the one-wave-per-query kernel and against fp64 -- also key counts that leave the last wave
    short (130, 515) and a spike that puts the maximum into one wave's range."""
    L = ops._lib.load()
    g = torch.Generator().manual_seed(hd + Lk)
    D = H * hd
    q, k, v = torch.randn(Z, Lq, D, generator=g), torch.randn(Z, Lk, D, generator=g), torch.randn(Z, Lk, D, generator=g)
    k[0, Lk - 3] = q[0, 1] * 3.0
    out = {}
    try:
        for mode in (0, 1):
            L.psam_attention_small_force_split(mode)
            out[mode] = ops.attention_small(cu(q).view(-1, D), cu(k).view(-1, D), cu(v).view(-1, D), torch.empty(Z * Lq, D, device="cuda"), Z, H, Lq, Lk, hd, hd ** -0.5)
        torch.cuda.synchronize()
    finally:
        L.psam_attention_small_force_split(1)
    want = _sdpa(q, k, v, H, hd ** -0.5)
    _close(out[1].view(Z, Lq, D), want, 2e-5, what="attention_small split")
    _close(out[0].view(Z, Lq, D), want, 2e-5, what="attention_small one wave")
    assert not torch.equal(out[0], out[1]) or Lk < 128      # (another summation order: the split kernel really ran)


@pytest.mark.parametrize("M,K,res", [(7, 256, True), (10, 128, True), (10, 2048, True), (64, 256, False), (1, 2048, True), (9, 256, False), (33, 1024, True)])
def test_linear_skinny_ln(ops, M, K, res):
    """psam_linear_skinny_ln -- `queries = norm(queries + Linear(x))` of the decoder's token side in one launch, the last workgroup normalising the rows
    (transformer.py:153-176) -- against psam_linear_skinny + psam_layernorm and fp64: last-bit agreement for K <= 256 (one K range: same products, same
    order), fp32 round-off for the split K = 2048 of the MLP's lin2; in place on the residual; the same bits launch after launch while another stream
    keeps the chip busy (who arrives last must not matter)."""
    g = torch.Generator().manual_seed(M * 13 + K)
    N = 256
    x, W, b = cu(torch.randn(M, K, generator=g)), cu(torch.randn(N, K, generator=g) / K ** 0.5), cu(torch.randn(N, generator=g))
    r = cu(torch.randn(M, N, generator=g)) if res else None
    lw, lb = cu(1.0 + 0.1 * torch.randn(N, generator=g)), cu(0.1 * torch.randn(N, generator=g))
    assert ops.skinny_ln_supported(M, N, K)
    want2 = ops.layernorm(ops.linear(x, W, b), lw, lb, 1e-6, residual=r)
    got = ops.linear_skinny_ln(x, W, b, lw, lb, 1e-6, residual=r)
    y = F.linear(x.cpu().double(), W.cpu().double(), b.cpu().double()) + (r.cpu().double() if res else 0)
    want = F.layer_norm(y, (N,), lw.cpu().double(), lb.cpu().double(), 1e-6)
    _close(got, want, 2e-5, what="linear_skinny_ln vs fp64")
    _close(got, want2.cpu().double(), 2e-6 if K <= 256 else 1e-5, what="linear_skinny_ln vs two launches")      # (K <= 256: the same products in the same order; the
    # LayerNorm arithmetic is compiled in another translation unit -- contraction may differ in the last bit)
    if res:      # in place: queries = norm(queries + ..)
        q = r.clone()
        ops.linear_skinny_ln(x, W, b, lw, lb, 1e-6, residual=q, out=q)
        assert torch.equal(q, got)
    s2 = torch.cuda.Stream()
    a, bb = torch.randn(2048, 2048, device="cuda"), torch.randn(2048, 2048, device="cuda")
    for rep in range(10):
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            for _ in range(3):
                a @ bb
        again = ops.linear_skinny_ln(x, W, b, lw, lb, 1e-6, residual=r)
        assert torch.equal(again, got), rep
    torch.cuda.synchronize()
    with pytest.raises(ops._lib.PointSamHipError):
        ops.linear_skinny_ln(x, cu(torch.randn(128, K, generator=g)), None, lw, lb, 1e-6)      # N != 256


@pytest.mark.parametrize("M,K,res", [(512, 128, True), (1000, 128, True), (7, 256, False), (2048, 64, True), (33, 512, True), (16, 16, False)])
def test_linear_ln256(ops, M, K, res):
    """psam_linear_ln256 -- `keys = norm4(keys + out_proj(attn))` of the decoder's patch side in one launch, a workgroup per 16 whole rows
    (transformer.py:170-175) -- against fp64 and against Linear + LayerNorm as two launches; ragged row counts; in place on the residual."""
    g = torch.Generator().manual_seed(M + K)
    N = 256
    x, W, b = cu(torch.randn(M, K, generator=g)), cu(torch.randn(N, K, generator=g) / K ** 0.5), cu(torch.randn(N, generator=g))
    r = cu(torch.randn(M, N, generator=g) * 3.0) if res else None
    lw, lb = cu(1.0 + 0.1 * torch.randn(N, generator=g)), cu(0.1 * torch.randn(N, generator=g))
    got = ops.linear_ln256(x, W, b, lw, lb, 1e-6, residual=r)
    y = F.linear(x.cpu().double(), W.cpu().double(), b.cpu().double()) + (r.cpu().double() if res else 0)
    _close(got, F.layer_norm(y, (N,), lw.cpu().double(), lb.cpu().double(), 1e-6), 2e-5, what="linear_ln256 vs fp64")
    with ops.gemm_mode("f32"):
        two = ops.layernorm(ops.linear(x, W, b), lw, lb, 1e-6, residual=r)
    _close(got, two.cpu().double(), 2e-5, what="linear_ln256 vs two launches")
    if res:
        q = r.clone()
        ops.linear_ln256(x, W, b, lw, lb, 1e-6, residual=q, out=q)
        assert torch.equal(q, got)
    assert torch.equal(got, ops.linear_ln256(x, W, b, lw, lb, 1e-6, residual=r))
    with pytest.raises(ops._lib.PointSamHipError):
        ops.linear_ln256(x, cu(torch.randn(128, K, generator=g)), None, lw, lb, 1e-6)      # N != 256


@pytest.mark.parametrize("Z,G,rep,K,Ns", [(1, 512, 1, 256, (128, 128, 128)), (4, 256, 2, 256, (128, 256)), (1, 1000, 1, 128, (512,)), (3, 77, 3, 48, (40, 16, 100)), (1, 2048, 1, 256, (256,))])
def test_linear_rows_multi(ops, Z, G, rep, K, Ns):
    """psam_linear_rows_multi -- the patch side's projections of a decoder layer in one launch (k / q from keys + key_pe, v from keys:
    transformer.py:160-175), exact fp32 products on 32 x 64 tiles -- against fp64: broadcast positional addend over `rep` prompt sets, ragged row and
    column counts, activations, and the single-job form the upscaling's first Linear and the mask encoder use."""
    g = torch.Generator().manual_seed(G + K)
    M = Z * G
    x, pos = torch.randn(M, K, generator=g), torch.randn(Z // rep, G, K, generator=g)
    jobs, want = [], []
    for i, N in enumerate(Ns):
        W, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        add, act = (i % 2 == 0), (0, 2, 1)[i % 3]
        xin = (x.view(Z // rep, rep, G, K) + pos[:, None]).reshape(M, K) if add else x
        y = F.linear(xin.double(), W.double(), b.double())
        want.append(F.gelu(y) if act == 1 else (F.relu(y) if act == 2 else y))
        jobs.append((cu(W), cu(b) if i != 1 else None, cu(pos.view(-1, K)) if add else None, act))
        if i == 1:
            want[-1] = F.linear(xin.double(), W.double())
            want[-1] = F.gelu(want[-1]) if act == 1 else (F.relu(want[-1]) if act == 2 else want[-1])
    got = ops.linear_rows_multi(cu(x), jobs, xadd_rows_per_set=G, xadd_rep=rep)
    for i, (a, w) in enumerate(zip(got, want)):
        _close(a, w, 1e-5, rtol=1e-5, what=f"linear_rows_multi job {i}")


def test_mlp3_pair_equals_two_launches(ops):
    """psam_mlp3_pair (hyper-networks + IoU head in one launch, mask_decoder.py:167-182): the bits of the two psam_mlp3 launches."""
    g = torch.Generator().manual_seed(5)
    Z, T, E, nmt = 3, 9, 256, 4
    hs = cu(torch.randn(Z, T, E, generator=g))
    mk = lambda M, dout: ops.Mlp3Weights([[(cu(torch.randn(o, i, generator=g) / i ** 0.5), cu(torch.randn(o, generator=g) * 0.1)) for i, o in ((E, E), (E, E), (E, dout))]
                                          for _ in range(M)])
    for C, dout in ((3, E), (1, E), (3, E // 2)):
        hw, iw = mk(C, dout), mk(1, nmt)
        h1, i1 = torch.empty(Z, C, dout, device="cuda"), torch.empty(Z, nmt, device="cuda")
        ops.mlp3(hs[:, 1, :], T * E, E, hw, h1, C * dout, dout, Z)
        ops.mlp3(hs, T * E, 0, iw, i1, nmt, 0, Z)
        h2, i2 = torch.full_like(h1, float("nan")), torch.full_like(i1, float("nan"))
        ops.mlp3_pair(hs[:, 1, :], T * E, E, hw, h2, C * dout, dout, hs, T * E, 0, iw, i2, nmt, 0, Z)
        assert torch.equal(h1, h2) and torch.equal(i1, i2)


@pytest.mark.parametrize("Z,G,rep,K", [(2, 512, 1, 256), (4, 128, 2, 256), (1, 300, 1, 128)])
def test_scale_pack_rows_dual(ops, Z, G, rep, K):
    """psam_scale_pack_rows_g8_add_dual: keys + key_pe and keys packed in one pass (transformer.py:160-170) = the bits of psam_scale_pack_rows_g8_add and
    psam_scale_pack_rows_g8."""
    L = ops._lib.load()
    g = torch.Generator().manual_seed(Z * G)
    X = cu(torch.randn(Z * G, K, generator=g) * torch.exp(torch.randn(Z * G, 1, generator=g)))
    pos = cu(torch.randn(Z // rep, G, K, generator=g))
    Kp = ops.packed_cols(K)
    e = lambda: torch.empty(Z * G, Kp, device="cuda")
    s = lambda: torch.empty(Z * G, device="cuda")
    Ps, ss, Px, sx, Pa, sa = e(), s(), e(), s(), e(), s()
    check = ops._lib.check
    check(L.psam_scale_pack_rows_g8_add_dual(X.data_ptr(), K, pos.data_ptr(), K, G, rep, Z * G, K, Ps.data_ptr(), ss.data_ptr(), Px.data_ptr(), sx.data_ptr(), Kp, None), "dual")
    check(L.psam_scale_pack_rows_g8_add(X.data_ptr(), K, pos.data_ptr(), K, G, rep, Z * G, K, Pa.data_ptr(), Kp, sa.data_ptr(), None), "add")
    Pw, sw = ops.scale_pack_rows_g8(X)
    torch.cuda.synchronize()
    i32 = lambda t: t.view(torch.int32)
    assert torch.equal(i32(Ps), i32(Pa)) and torch.equal(ss, sa)
    assert torch.equal(i32(Px), i32(Pw)) and torch.equal(sx, sw)


def test_invalid_arguments_raise(ops):
    with pytest.raises(ops._lib.PointSamHipError):
        ops.knn(torch.zeros(1, 4, 3, device="cuda"), torch.zeros(1, 8, 3, device="cuda"), 9)      # K > N
    with pytest.raises(ops._lib.PointSamHipError):
        ops.linear(torch.zeros(8, 6, device="cuda"), torch.zeros(8, 6, device="cuda"))             # K % 4 != 0
    with pytest.raises(ops._lib.PointSamHipError):
        ops.fps(torch.zeros(1, 8, 3, device="cuda"), 9)                                             # G > N


@pytest.mark.parametrize("N,B,rep", [(400, 1, 3), (3000, 2, 2), (32768, 1, 1)])
def test_border_farthest_and_error_regions(ops, N, B, rep):
    g = torch.Generator().manual_seed(N)
    xyz, _ = _cloud(B, N, seed=N)
    Z = B * rep
    gt = torch.stack([(xyz[z // rep, :, z % 3] > 0.05 * z) for z in range(Z)])                  # [Z,N] bool
    logits = torch.randn(Z, N, generator=g) + torch.where(gt, 0.8, -0.8)
    fn, fp = ops.error_regions(cu(gt), cu(logits))
    pred = logits > 0
    assert torch.equal(fn.cpu().bool(), gt & ~pred) and torch.equal(fp.cpu().bool(), ~gt & pred)
    fn0, fp0 = ops.error_regions(cu(gt), None)
    assert torch.equal(fn0.cpu().bool(), gt) and not fp0.any()
    regions = torch.cat([gt, gt & ~pred, ~gt & pred, torch.zeros(1, N, dtype=torch.bool).expand(Z, N), torch.ones(1, N, dtype=torch.bool).expand(Z, N)])
    # kernel takes Z' = B*rep' regions: run each family separately
    for name, reg in (("gt", gt), ("fn", gt & ~pred), ("fp", ~gt & pred), ("empty", torch.zeros(Z, N, dtype=torch.bool)), ("full", torch.ones(Z, N, dtype=torch.bool))):
        idx, dist = ops.border_farthest(cu(xyz), cu(reg))
        for z in range(Z):
            wi, wd = O.border_farthest(xyz[z // rep], reg[z])
            assert int(idx[z]) == wi and float(dist[z]) == wd, (name, z, int(idx[z]), wi, float(dist[z]), wd)
