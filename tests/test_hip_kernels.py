"""GPU: each HIP kernel (through the C-ABI) against a torch fp32 reference of the same op."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from isegprobe_amd import hip_ops
    assert torch.cuda.is_available()
    return hip_ops


def bf(x):
    return x.to(BF)


def rel_err(a, b):
    return (a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-12)


def test_library_loads(ops):
    from isegprobe_amd import _lib
    assert _lib.lib().isp_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("disks", [True, False])
@pytest.mark.parametrize("shape", [(2, 3, 56, 70), (2, 24, 224, 224), (1, 5, 30, 41)])
def test_click_maps_vs_oracle(ops, disks, shape):
    from oracle.click_maps import click_maps
    B, P, H, W = shape
    rng = np.random.default_rng(0)
    pts = -np.ones((B, 2 * P, 3), np.float32)
    for b in range(B):
        for pol in range(2):
            n = rng.integers(0 if pol else 1, P + 1)
            for i in range(n):
                pts[b, pol * P + i] = (rng.uniform(0, H - 1), rng.uniform(0, W - 1), i)
    pts[0, :1, :2] = np.floor(pts[0, :1, :2])
    ref = click_maps(pts, H, W, 5, 1.0, disks)
    out = ops.click_maps(torch.from_numpy(pts).cuda(), H, W, 5, 1.0, disks).cpu().numpy()
    if disks:
        assert np.array_equal(out, ref)  # bit-exact
    else:
        np.testing.assert_allclose(out, ref, atol=1e-6, rtol=0)


@pytest.mark.parametrize("case", ["int_p1", "int_p3", "frac_p3", "int_p24", "noneg_p3", "frac_p24_224"])
def test_click_maps_vs_reference_golden(ops, golden, case):
    """The torch path of DistMaps (core/model/ops.py:35-77) as the REFERENCE itself computed it
    (tests/golden/click_maps.npz): disks bit-exact incl. fractional clicks, tanh maps to 1e-6."""
    g = golden("click_maps")
    H, W = (int(v) for v in g[case + "_hw"])
    pts = torch.from_numpy(g[case + "_points"]).cuda()
    B = pts.shape[0]
    disks = ops.click_maps(pts, H, W, 5, 1.0, True).cpu().numpy()
    ref = np.unpackbits(g[case + "_disks_bits"])[:B * 2 * H * W].reshape(B, 2, H, W).astype(np.float32)
    assert np.array_equal(disks, ref)
    np.testing.assert_allclose(ops.click_maps(pts, H, W, 5, 1.0, False).cpu().numpy(), g[case + "_tanh"], atol=1e-6, rtol=0)


@pytest.mark.parametrize("via_module", [False, True])
@pytest.mark.parametrize("case", ["p2", "p5", "half"])
def test_click_maps_round_clicks_vs_compiled_cython_golden(ops, golden, case, via_module):
    """The drop-in for the reference's ONLY native entry point: get_dist_maps (core/utils/cython/_get_dist_maps.pyx:18-64)
    + its host loop and post-processing (core/model/ops.py:15-34,72-75) == isp_click_maps_fwd(round_clicks=1) /
    DistMaps(cpu_mode=True).  Golden = the Cython reference compiled and run in the build container
    (tests/golden/dist_maps_bfs.npz; the 'half' case pins round-half-even on .5 coordinates).  The BFS returns squared
    distances; norm_delimeter 1 feeds the disk threshold (bit-exact), norm_delimeter 5 = radius*scale feeds tanh."""
    g = golden("dist_maps_bfs")
    H, W = (int(v) for v in g[case + "_hw"])
    pts = torch.from_numpy(g[case + "_points"][None]).cuda()
    ref_disks = (g[case + "_d1"] <= np.float32(25.0)).astype(np.float32)[None]
    ref_tanh = np.tanh(np.sqrt(g[case + "_d5"]) * np.float32(2))[None]
    if via_module:
        from isegprobe_amd.core.model.ops import DistMaps
        x = torch.zeros(1, 3, H, W, device="cuda")
        disks = DistMaps(norm_radius=5, spatial_scale=1.0, cpu_mode=True, use_disks=True)(x, pts)
        tanh = DistMaps(norm_radius=5, spatial_scale=1.0, cpu_mode=True, use_disks=False)(x, pts)
    else:
        disks = ops.click_maps(pts, H, W, 5, 1.0, True, round_clicks=True)
        tanh = ops.click_maps(pts, H, W, 5, 1.0, False, round_clicks=True)
    assert np.array_equal(disks.cpu().numpy(), ref_disks)
    np.testing.assert_allclose(tanh.cpu().numpy(), ref_tanh, atol=1e-6, rtol=0)
    if case == "half":  # rounding matters here: the un-rounded torch path gives other disks
        assert not np.array_equal(ops.click_maps(pts, H, W, 5, 1.0, True).cpu().numpy(), ref_disks)


def test_normalize(ops):
    x = torch.rand(2, 4, 28, 40, device="cuda")
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    y, prev = ops.normalize(x, mean, std)
    ref = (x[:, :3].cpu() - torch.tensor(mean)[None, :, None, None]) / torch.tensor(std)[None, :, None, None]
    assert torch.equal(y.cpu(), ref)
    assert torch.equal(prev, x[:, 3:])


@pytest.mark.parametrize("D", [64, 384, 404, 1024])
def test_layernorm(ops, D):
    x = torch.randn(1000, D, device="cuda") * 3 + 1
    g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    y = ops.layernorm(x, g, b, 1e-6, out_dtype=torch.float32)
    ref = F.layer_norm(x, (D,), g, b, 1e-6)
    assert (y - ref).abs().max().item() < 2e-5
    y16 = ops.layernorm(x, g, b, 1e-6)
    assert rel_err(y16, ref) < 1e-2
    # cls-drop remap
    T = 9
    xs = torch.randn(4 * (T + 1), D, device="cuda")
    yd = ops.layernorm(xs, g, b, 1e-6, out_dtype=torch.float32, group_out=T, skip=1, rows_out=4 * T)
    refd = F.layer_norm(xs.view(4, T + 1, D)[:, 1:], (D,), g, b, 1e-6).reshape(-1, D)
    assert (yd - refd).abs().max().item() < 2e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 192, 128), (1000, 384, 384), (4100, 1152, 384),
                                   (257, 64, 1216), (129, 132, 64)])
@pytest.mark.parametrize("act", [None, "relu", "gelu"])
def test_gemm_bias_act(ops, M, N, K, act):
    torch.manual_seed(M + N + K)
    A, W = bf(torch.randn(M, K, device="cuda")), bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")
    y = ops.linear(A, W, bias, act)
    ref = A.float() @ W.float().t() + bias
    ref = {None: lambda t: t, "relu": F.relu, "gelu": F.gelu}[act](ref)
    assert rel_err(y, ref) < 1e-2
    if act is None:
        y32 = ops.linear(A, W, bias, None, out_dtype=torch.float32)
        assert (y32 - ref).abs().max().item() < 2e-3


def test_gemm_asymmetric_identity(ops):
    """A = I against an asymmetric W catches transposed / permuted fragment layouts."""
    K = N = 128
    A = bf(torch.eye(K, device="cuda"))
    W = bf((torch.arange(N * K, device="cuda").float().reshape(N, K) % 251) / 16)
    y = ops.linear(A, W, None, None, out_dtype=torch.float32)
    assert torch.equal(y, W.float().t())


def test_gemm_residual_and_tokens(ops):
    from isegprobe_amd import _lib
    M, N, K = 515, 384, 1536
    A, W = bf(torch.randn(M, K, device="cuda")), bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias, gamma = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    x0 = torch.randn(M, N, device="cuda")
    x = x0.clone()
    ops.linear_residual_(x, A, W, bias, gamma)
    ref = x0 + gamma * (A.float() @ W.float().t() + bias)
    assert (x - ref).abs().max().item() < 5e-3
    # token scatter: B=3 images, T=5 tokens each -> rows b*(T+1)+1+t
    Bn, T = 3, 5
    A2 = bf(torch.randn(Bn * T, 64, device="cuda"))
    W2 = bf(torch.randn(N, 64, device="cuda") / 8)
    pos = torch.randn(T + 1, N, device="cuda")
    xt = torch.zeros(Bn * (T + 1), N, device="cuda")
    ops.gemm(A2, W2, ops._epilogue(_lib.EP_TOKENS_F32, xt, N, bias, None, pos, T))
    ref2 = (A2.float() @ W2.float().t() + bias).view(Bn, T, N) + pos[1:]
    assert (xt.view(Bn, T + 1, N)[:, 1:] - ref2).abs().max().item() < 5e-3
    assert xt.view(Bn, T + 1, N)[:, 0].abs().max().item() == 0


@pytest.mark.parametrize("B,H,W,C,N", [(1, 8, 8, 64, 64), (2, 20, 24, 64, 64), (1, 37, 45, 128, 192),
                                       (2, 56, 56, 384, 384),
                                       # one-wave-per-SIMD patch kernel: a single channel block, 128-channel blocks with a
                                       # ragged last block (448), exact 128-channel tiling, interior + border tiles
                                       (2, 33, 47, 64, 192), (1, 40, 40, 448, 448), (1, 16, 16, 128, 128),
                                       (1, 48, 32, 192, 1024), (1, 21, 19, 64, 320)])
def test_conv3x3(ops, B, H, W, C, N):
    torch.manual_seed(C + H)
    x = bf(torch.randn(B, C, H, W, device="cuda"))
    w = bf(torch.randn(N, C, 3, 3, device="cuda") / math.sqrt(9 * C))
    bias = torch.randn(N, device="cuda")
    ref = F.relu(F.conv2d(x.float(), w.float(), bias, padding=1))
    y = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous(), w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous(), bias, "relu")
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-2


@pytest.mark.parametrize("B,H,W,C,N", [(3, 200, 200, 192, 192), (2, 150, 170, 64, 384), (2, 130, 130, 128, 256),
                                       (1, 300, 280, 256, 192)])
def test_conv3x3_tile_stream(ops, B, H, W, C, N):
    """More output tiles than CUs: a workgroup of the patch kernel walks several tiles and its K-step stream runs across
    them (the next tile's first patch / weight tiles are fetched during the last channel block, the epilogue stages through
    the free patch buffer).  Odd (3) and single channel-block counts flip the patch-buffer parity per tile; border tiles
    and both channel-block widths (192, 128) are in there.  Also the fused epilogues on such a launch."""
    torch.manual_seed(C + H)
    x = bf(torch.randn(B, C, H, W, device="cuda"))
    w = bf(torch.randn(N, C, 3, 3, device="cuda") / math.sqrt(9 * C))
    bias = torch.randn(N, device="cuda")
    conv = F.conv2d(x.float(), w.float(), bias, padding=1)
    xn, wt = x.permute(0, 2, 3, 1).contiguous(), w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    y = ops.conv3x3(xn, wt, bias, "relu")
    assert (y.permute(0, 3, 1, 2).float() - F.relu(conv)).abs().max().item() < 3e-2
    assert rel_err(y.permute(0, 3, 1, 2), F.relu(conv)) < 1e-2
    y2 = ops.conv3x3(xn, wt, bias, "relu")
    assert torch.equal(y, y2)
    wcls = torch.randn(N, device="cuda") / math.sqrt(N)
    z = ops.conv3x3_relu_classifier(xn, wt, bias, wcls, 0.25)
    refz = (F.relu(conv) * wcls.view(1, N, 1, 1)).sum(1) + 0.25
    assert (z - refz).abs().max().item() < 2e-2 * refz.abs().max().item() + 1e-2


@pytest.mark.parametrize("B,H,W,C,N", [(2, 56, 56, 384, 384), (1, 37, 45, 128, 192), (3, 200, 200, 192, 192)])
def test_conv3x3_f16(ops, B, H, W, C, N):
    """IEEE-half form of the patch conv (isp_conv3x3_nhwc_f16): plain, folded-affine and classifier-fused epilogues
    against fp32 on the half-rounded operands; its rounding error is ~8x below the bf16 form's on the same data."""
    torch.manual_seed(C + H)
    x = torch.randn(B, C, H, W, device="cuda")
    w = torch.randn(N, C, 3, 3, device="cuda") / math.sqrt(9 * C)
    bias = torch.randn(N, device="cuda")
    xh, wh = x.half(), w.half()
    conv = F.conv2d(xh.float(), wh.float(), bias, padding=1)
    xn, wt = xh.permute(0, 2, 3, 1).contiguous(), wh.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    y = ops.conv3x3(xn, wt, bias, "relu")
    assert y.dtype == torch.float16
    e16 = (y.permute(0, 3, 1, 2).float() - F.relu(conv)).abs().max().item()
    yb = ops.conv3x3(bf(x).permute(0, 2, 3, 1).contiguous(), bf(w).permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous(), bias, "relu")
    eb = (yb.permute(0, 3, 1, 2).float() - F.relu(F.conv2d(x, w, bias, padding=1))).abs().max().item()
    e16_full = (y.permute(0, 3, 1, 2).float() - F.relu(F.conv2d(x, w, bias, padding=1))).abs().max().item()
    print(f"conv3x3 f16: max err {e16:.3g} vs fp32 on half operands, {e16_full:.3g} vs fp32 on fp32 operands (bf16 form: {eb:.3g})")
    assert e16 < 4e-3 and e16_full < eb / 3
    taps = torch.randn(9, N, device="cuda") * 0.3
    yt = ops.conv3x3_folded_affine(xn, wt, bias + taps.sum(0), taps)
    inside = F.conv2d(torch.ones(1, 1, H, W, device="cuda"), torch.eye(9, device="cuda").view(9, 1, 3, 3), padding=1)
    ref = F.relu(conv + torch.einsum("tn,thw->nhw", taps, inside[0]))
    assert (yt.permute(0, 3, 1, 2).float() - ref).abs().max().item() < 4e-3
    wcls = torch.randn(N, device="cuda") / math.sqrt(N)
    z = ops.conv3x3_relu_classifier(xn, wt, bias, wcls, 0.25)
    refz = (F.relu(conv) * wcls.view(1, N, 1, 1)).sum(1) + 0.25
    assert (z - refz).abs().max().item() < 2e-3 * refz.abs().max().item() + 1e-3


@pytest.mark.parametrize("H,W,C,N", [(48, 48, 128, 384), (37, 29, 64, 192), (32, 32, 128, 128)])
def test_conv3x3_fused_epilogues(ops, H, W, C, N):
    """Folded-affine first conv (border-exact tap table) and classifier-fused last conv on interior + border tiles."""
    torch.manual_seed(H + N)
    B = 2
    x = bf(torch.randn(B, C, H, W, device="cuda"))
    w = bf(torch.randn(N, C, 3, 3, device="cuda") / math.sqrt(9 * C))
    wt = w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    xn = x.permute(0, 2, 3, 1).contiguous()
    bias = torch.randn(N, device="cuda")
    # taps[t][n]: constant contributed through tap t wherever that tap lies inside the image
    taps = torch.randn(9, N, device="cuda") * 0.3
    y = ops.conv3x3_folded_affine(xn, wt, bias + taps.sum(0), taps)
    inside = F.conv2d(torch.ones(1, 1, H, W, device="cuda"), torch.eye(9, device="cuda").view(9, 1, 3, 3), padding=1)  # [1,9,H,W]
    const = torch.einsum("tn,thw->nhw", taps, inside[0])
    ref = F.relu(F.conv2d(x.float(), w.float(), bias, padding=1) + const)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-2
    wcls = torch.randn(N, device="cuda") / math.sqrt(N)
    z = ops.conv3x3_relu_classifier(xn, wt, bias, wcls, 0.25)
    refz = (F.relu(F.conv2d(x.float(), w.float(), bias, padding=1)) * wcls.view(1, N, 1, 1)).sum(1) + 0.25
    assert (z - refz).abs().max().item() < 2e-2 * refz.abs().max().item() + 1e-2


def test_gemm_bf16_store_paths(ops):
    """bf16 outputs leave through LDS as 16-byte chunks when the leading dimension allows it; a leading dimension
    that is only a multiple of 4 and partial edge tiles take the direct path.  Same values either way."""
    from isegprobe_amd import _lib
    torch.manual_seed(5)
    for M, N, K, ldo in ((512, 384, 128, 384), (515, 384, 128, 388), (300, 200, 64, 200), (256, 128, 64, 132)):
        A, W = bf(torch.randn(M, K, device="cuda")), bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
        bias = torch.randn(N, device="cuda")
        out = torch.full((M, ldo), 7.0, device="cuda", dtype=BF)
        ops.gemm(A, W, ops._epilogue(_lib.EP_BIAS_RELU_BF16, out, ldo, bias))
        ref = F.relu(A.float() @ W.float().t() + bias)
        assert rel_err(out[:, :N], ref) < 1e-2
        if ldo > N:
            assert torch.equal(out[:, N:], torch.full((M, ldo - N), 7.0, device="cuda", dtype=BF))  # padding untouched


@pytest.mark.parametrize("B,L,heads", [(1, 64, 1), (2, 257, 2), (2, 1025, 6), (1, 130, 3), (1, 33, 2), (1, 96, 1), (1, 97, 1),
                                       (2, 160, 2), (1, 1, 1)])
def test_attention_packed(ops, B, L, heads):
    torch.manual_seed(L)
    D = heads * 64
    qkv = bf(torch.randn(B * L, 3 * D, device="cuda"))
    out = ops.attention_packed_qkv(qkv, B, L, heads, 64 ** -0.5)
    q, k, v = qkv.float().view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    p = ((q * 64 ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
    ref = (p @ v).transpose(1, 2).reshape(B * L, D)
    assert (out.float() - ref).abs().max().item() < 2e-2
    assert rel_err(out, ref) < 2e-2
    # the ViT's form: q already carries scale * log2(e), the scores are base-2 logits
    qkv2 = qkv.clone()
    qkv2[:, :D] = bf(qkv[:, :D].float() * ops.ATTENTION_LOGIT2_SCALE)
    out2 = ops.attention_packed_qkv(qkv2, B, L, heads, None, q_logit2=True)
    q2 = qkv2[:, :D].float().view(B, L, heads, 64).permute(0, 2, 1, 3)
    p2 = (q2 @ k.transpose(-2, -1) * math.log(2.0)).softmax(-1)
    ref2 = (p2 @ v).transpose(1, 2).reshape(B * L, D)
    assert (out2.float() - ref2).abs().max().item() < 2e-2
    assert rel_err(out2, ref2) < 2e-2


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("L", [97, 130, 1025])
def test_attention_tail_rows_read_as_zeros(ops, half, L):
    """The head_dim-64 kernel stages whole 64-key tiles; the rows past Lk of the last tile must come back as ZEROS
    from the buffer range check (their P is 0, but 0 x NaN is NaN).  The packed qkv sits at the END of a NaN-filled
    allocation, so for the last (batch, head) anything read behind the last key is NaN."""
    torch.manual_seed(L)
    B, heads, D = 2, 2, 128
    dt = torch.float16 if half else BF
    n = B * L * 3 * D
    big = torch.full((n + 64 * 3 * D + 4096,), float("nan"), device="cuda", dtype=dt)
    qkv = big[:n].view(B * L, 3 * D)
    qkv.copy_(torch.randn(B * L, 3 * D, device="cuda").to(dt))
    out = ops.attention_packed_qkv(qkv, B, L, heads, None, q_logit2=True)
    assert torch.isfinite(out.float()).all()
    q, k, v = qkv.float().view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = ((q @ k.transpose(-2, -1) * math.log(2.0)).softmax(-1) @ v).transpose(1, 2).reshape(B * L, D)
    assert (out.float() - ref).abs().max().item() < 2e-2
    if not half:
        out2 = ops.attention_packed_qkv(qkv, B, L, heads, 0.125)
        ref2 = ((q * 0.125 @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * L, D)
        assert torch.isfinite(out2.float()).all() and (out2.float() - ref2).abs().max().item() < 2e-2


def _attn_ref(q2, k, v):
    """fp64 softmax over base-2 logits: q2 [B,Lq,H,hd] already carries scale * log2(e)."""
    sc = q2.double().permute(0, 2, 1, 3) @ k.double().permute(0, 2, 3, 1) * math.log(2.0)
    return (sc.softmax(-1) @ v.double().permute(0, 2, 1, 3)).permute(0, 2, 1, 3)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("hd,B,H,Lq,Lk", [(64, 2, 2, 1025, 1025), (64, 1, 3, 512, 192), (64, 1, 1, 300, 130), (64, 2, 1, 256, 128),
                                          (128, 1, 2, 512, 1024), (128, 2, 1, 700, 257), (128, 1, 4, 256, 448), (128, 1, 1, 1030, 193)])
def test_attention_pipelined_kernel(ops, hd, B, H, Lq, Lk, half):
    """csrc/attention_pipe.hip (64 queries per wave as two streams, QK^T of tile t+1 / PV of tile t interleaved with the
    softmax; an experiment that is off by default, called here through its own entry point): full and ragged last key
    tiles, query counts that are / are not multiples of the 256-query workgroup, strided packed operands, both dtypes;
    against an fp64 softmax."""
    torch.manual_seed(hd + Lq + Lk)
    dt = torch.float16 if half else BF
    q = (torch.randn(B, Lq, H, hd, device="cuda") * (hd ** -0.5 * 1.4426950408889634) * 1.5).to(dt)
    kv = torch.randn(B, Lk, 2, H, hd, device="cuda").to(dt)   # K and V interleaved: equal, non-trivial strides
    k, v = kv[:, :, 0], kv[:, :, 1]
    out = ops.attention_pipe(q, k, v)
    ref = _attn_ref(q, k, v)
    err = (out.double() - ref).abs().max().item()
    assert torch.isfinite(out.float()).all() and err < 2e-2, err


@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("spike_key,gain", [(3, 300.0), (70, 30.0), (200, 300.0), (200, 9.0), (319, 60.0), (320, 300.0)])
def test_attention_pipelined_deferred_max(ops, hd, spike_key, gain):
    """The pipelined kernel moves a row's reference maximum only when a tile exceeds it by more than 2^6 -- at the END of
    the iteration that computed the tile, after the previous tile's PV product.  Spikes below, near and far above the
    threshold in the first, a middle and the last (one-key) tile, for a query of either stream of a wave; and a row that is
    ~ -90 in the first tile and ~ 0 later."""
    torch.manual_seed(spike_key)
    B, L, H = 1, 321, 1
    q = torch.randn(B, L, H, hd, device="cuda") * 0.5
    k = torch.randn(B, L, H, hd, device="cuda") * 0.5
    v = torch.randn(B, L, H, hd, device="cuda")
    for qi in (17, 45):  # stream 0 / stream 1 of wave 0
        k[0, spike_key, 0] += q[0, qi, 0] * gain / q[0, qi, 0].square().sum() / 2
    u = torch.zeros(hd, device="cuda")
    u[5] = 1.0
    k[0, :64, 0] += 3 * u
    q[0, 40, 0] = -30 * u
    out = ops.attention_pipe(bf(q), bf(k), bf(v))
    ref = _attn_ref(bf(q), bf(k), bf(v))
    assert (out.double() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("Lk", [257, 1024])
def test_attention_deferred_max_head_dim_128(ops, half, Lk):
    """LoftUp's shape class (many queries, few keys, head_dim 128, base-2-logit queries): routed to the deferred-maximum
    kernel with 8 waves (attention64_kernel<..., 128, 8>) once 256-query blocks fill the chip; ragged and full last key
    tiles, a spike that forces the rescale late in the key sequence; against an fp32 softmax on the GPU."""
    torch.manual_seed(Lk)
    dt = torch.float16 if half else BF
    B, H, hd, Lq = 1, 4, 128, 131072
    q = (torch.randn(B, Lq, H, hd, device="cuda") * (101 ** -0.5 * 1.4426950408889634)).to(dt)
    k = torch.randn(B, Lk, H, hd, device="cuda").to(dt)
    v = torch.randn(B, Lk, H, hd, device="cuda").to(dt)
    k[0, Lk - 3, 1] = (q[0, 777, 1].float() * 40).to(dt)  # score(777, Lk-3) in head 1 far above the running maximum
    out = ops.attention(q, k, v, None, q_logit2=True)
    worst = 0.0
    for h in range(H):
        sc = (q[0, :, h].float() @ k[0, :, h].float().t()) * math.log(2.0)
        ref = sc.softmax(-1) @ v[0, :, h].float()
        worst = max(worst, (out[0, :, h].float() - ref).abs().max().item())
    assert torch.isfinite(out.float()).all() and worst < 2e-2, worst


def test_attention_online_softmax_rescale(ops):
    """A key whose score dwarfs the rest in a LATE tile forces the running-max rescale."""
    B, L, heads = 1, 256, 1
    q = torch.randn(B, L, heads, 64, device="cuda") * 0.1
    k = torch.randn(B, L, heads, 64, device="cuda") * 0.1
    v = torch.randn(B, L, heads, 64, device="cuda")
    k[0, 200] = q[0, 17] * 400  # spike for query 17 in the 4th KV tile
    out = ops.attention(bf(q), bf(k), bf(v), 1.0)
    qf, kf, vf = bf(q).float(), bf(k).float(), bf(v).float()
    p = (qf.permute(0, 2, 1, 3) @ kf.permute(0, 2, 3, 1)).softmax(-1)
    ref = (p @ vf.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    assert (out.float() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("spike_key,gain", [(3, 400.0), (70, 30.0), (200, 400.0), (200, 9.0), (255, 60.0), (256, 400.0)])
def test_attention_deferred_max(ops, spike_key, gain):
    """The head_dim-64 kernel keeps a row's reference maximum until a tile exceeds it by a threshold; spikes below,
    near and far above the threshold, in the first, a middle and the last (one-key) tile, against an fp64 softmax.
    Rows whose scores are all very negative (the first tile sets the reference whatever its sign) are in there too."""
    torch.manual_seed(spike_key)
    B, L, heads = 1, 257, 1
    q = torch.randn(B, L, heads, 64, device="cuda") * 0.5
    k = torch.randn(B, L, heads, 64, device="cuda") * 0.5
    v = torch.randn(B, L, heads, 64, device="cuda")
    k[0, spike_key] = q[0, 17] * gain / q[0, 17].square().sum()  # score(17, spike_key) = gain
    u = torch.zeros(64, device="cuda")
    u[5] = 1.0
    k[0, :64, 0] += 3 * u                                         # query 40: scores ~ -90 in the first tile, ~ 0 later
    q[0, 40, 0] = -30 * u
    out, lse = ops.attention_lse(bf(q), bf(k), bf(v), 1.0)
    qd, kd, vd = bf(q).double(), bf(k).double(), bf(v).double()
    sc = qd.permute(0, 2, 1, 3) @ kd.permute(0, 2, 3, 1)
    ref = (sc.softmax(-1) @ vd.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    assert (out.double() - ref).abs().max().item() < 2e-2
    ref_lse = torch.logsumexp(sc, -1)[0, 0] / math.log(2.0)
    got = lse.reshape(-1)[:L].double()
    assert (got - ref_lse).abs().max().item() < 2e-2 + 2e-3 * ref_lse.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 4, 5, 56, 70, 64), (1, 32, 32, 448, 448, 384)])
def test_bilinear_nhwc(ops, shape):
    B, h, w, H, W, C = shape
    x = bf(torch.randn(B, C, h, w, device="cuda"))
    y = ops.resize_bilinear_nhwc(x.permute(0, 2, 3, 1).contiguous(), H, W)
    ref = F.interpolate(x.float(), (H, W), mode="bilinear", align_corners=True)
    assert (y.permute(0, 3, 1, 2).float() - ref).abs().max().item() < 2e-2


def test_bilinear_nchw_f32_and_identity(ops):
    x = torch.randn(3, 1, 37, 53, device="cuda")
    y = ops.resize_bilinear_nchw_f32(x, 90, 120)
    ref = F.interpolate(x, (90, 120), mode="bilinear", align_corners=True)
    assert (y - ref).abs().max().item() < 1e-5
    assert torch.equal(ops.resize_bilinear_nchw_f32(x, 37, 53), x)  # same-size resize is exact


def test_classifier_and_layout(ops):
    x = bf(torch.randn(2, 20, 24, 384, device="cuda"))
    w = torch.randn(384, device="cuda")
    y = ops.classifier(x, w, 0.3)
    ref = (x.float() * w).sum(-1) + 0.3
    assert (y - ref).abs().max().item() < 1e-3
    n = ops.nhwc_bf16_to_nchw_f32(x)
    assert torch.equal(n, x.float().permute(0, 3, 1, 2))
    f = torch.randn(2, 9, 384, device="cuda")  # token-major view as [B,C,h,w]
    view = f.reshape(2, 3, 3, 384).permute(0, 3, 1, 2)
    back = ops.nchw_f32_to_nhwc_bf16(view)
    assert torch.equal(back, f.reshape(2, 3, 3, 384).to(BF))


@pytest.mark.parametrize("M,D,HID", [(128, 384, 1536), (128 * 5 + 37, 384, 1536), (4100, 384, 1536), (128 * 300, 384, 1536)])
def test_vit_mlp_fused(ops, M, D, HID):
    """x += ls * fc2(GELU(fc1(LayerNorm(x)))) (dinov2/layers/block.py:92-117, mlp.py:34-40): the token-stationary fused
    kernel against torch fp32 and against the three-kernel route it replaces; partial last tile, several tiles per
    workgroup (M > 256 * 128), asymmetric weights (a permuted or transposed operand cannot cancel)."""
    torch.manual_seed(M + D)
    x = torch.randn(M, D, device="cuda") * 2 + 0.3
    nw, nb = torch.randn(D, device="cuda") * 0.3 + 1, torch.randn(D, device="cuda") * 0.2
    w1, b1 = torch.randn(HID, D, device="cuda") / math.sqrt(D), torch.randn(HID, device="cuda") * 0.3
    w2, b2 = torch.randn(D, HID, device="cuda") / math.sqrt(HID), torch.randn(D, device="cuda") * 0.3
    ls = torch.randn(D, device="cuda") * 0.5 + 1
    ref = x + ls * (F.gelu(F.layer_norm(x, (D,), nw, nb, 1e-6) @ w1.t() + b1) @ w2.t() + b2)
    y = ops.vit_mlp_fused_(x.clone(), *ops.vit_mlp_pack(nw, nb, w1, b1, w2, b2, ls), 1e-6)
    h = ops.layernorm(x, nw, nb, 1e-6)
    hid = ops.linear(h, bf(w1), b1, "gelu")
    y3 = x.clone()
    ops.linear_residual_(y3, hid, bf(w2), b2, ls)
    e, e3 = (y - ref).abs().max().item(), (y3 - ref).abs().max().item()
    print(f"fused MLP M={M} D={D}: max err {e:.3g} (three-kernel route {e3:.3g}), ref rms {ref.pow(2).mean().sqrt():.3f}")
    assert e < 3e-2 and e < 2 * e3 + 1e-3


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("B,T", [(3, 200), (2, 1024), (40, 1024), (5, 128)])
def test_vit_mlp_fused_rows(ops, B, T, half):
    """The fused MLP over the patch-token rows of a [B, T+1, D] stream (row 0 of every image = class token, left untouched),
    tiled per image: ragged last tile per image (T = 200), exact tiles, more tiles than workgroups (40 x 8 = 320: several
    tiles per workgroup, the weight stream running across tiles with a different chunk rotation per tile), bf16 and
    IEEE-half operands -- against torch fp32, against the three-kernel route, and batch-invariance of every image."""
    D, HID = 384, 1536
    torch.manual_seed(B * 1000 + T)
    dt = torch.float16 if half else BF
    x = torch.randn(B * (T + 1), D, device="cuda") * 2 + 0.3
    nw, nb = torch.randn(D, device="cuda") * 0.3 + 1, torch.randn(D, device="cuda") * 0.2
    w1, b1 = torch.randn(HID, D, device="cuda") / math.sqrt(D), torch.randn(HID, device="cuda") * 0.3
    w2, b2 = torch.randn(D, HID, device="cuda") / math.sqrt(HID), torch.randn(D, device="cuda") * 0.3
    ls = torch.randn(D, device="cuda") * 0.5 + 1
    ref = x + ls * (F.gelu(F.layer_norm(x, (D,), nw, nb, 1e-6) @ w1.t() + b1) @ w2.t() + b2)
    P = ops.vit_mlp_pack(nw, nb, w1, b1, w2, b2, ls, dtype=dt)
    y = ops.vit_mlp_fused_rows_(x.clone(), *P, 1e-6, B, T + 1, 1, T)
    y3 = x.clone()
    ops.linear_residual_(y3, ops.linear(ops.layernorm(x, nw, nb, 1e-6, out_dtype=dt), w1.to(dt), b1, "gelu"), w2.to(dt), b2, ls)
    yv, xv, rv, y3v = (t.view(B, T + 1, D) for t in (y, x, ref, y3))
    assert torch.equal(yv[:, 0], xv[:, 0])  # class-token rows: not this kernel's
    e, e3 = (yv[:, 1:] - rv[:, 1:]).abs().max().item(), (y3v[:, 1:] - rv[:, 1:]).abs().max().item()
    assert e < (1e-2 if half else 3e-2) and e < 2 * e3 + 1e-3, (e, e3)
    # an image's rows do not depend on where in the batch (or in which batch) the image sits
    b = B // 2
    alone = ops.vit_mlp_fused_rows_(xv[b].clone().view(T + 1, D), *P, 1e-6, 1, T + 1, 1, T)
    assert torch.equal(alone.view(T + 1, D)[1:], yv[b, 1:])


def test_sigmoid_gelu_far_from_zero(ops):
    """The kernels' GELU is x * sigmoid(x * quartic(x^2)) with a negative leading coefficient: without the clamp of x^2 it
    returns 0 for x > 11.1 and x for x < -11.1.  Hidden pre-activations of +-40 through the fused ViT MLP and the
    FeatUp-JBU range projection (both use it), against the erf form."""
    torch.manual_seed(3)
    M, D, HID = 256, 384, 1536
    x = torch.randn(M, D, device="cuda")
    nw, nb = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    w1, b1 = torch.randn(HID, D, device="cuda") / math.sqrt(D), torch.randn(HID, device="cuda") * 0.3
    b1[::3] += 40.0
    b1[1::3] -= 40.0
    w2, b2 = torch.randn(D, HID, device="cuda") / math.sqrt(HID) * 0.05, torch.zeros(D, device="cuda")
    ls = torch.ones(D, device="cuda")
    pre = bf(F.layer_norm(x, (D,), nw, nb, 1e-6)).float() @ bf(w1).float().t() + b1
    ref = x + bf(F.gelu(pre)).float() @ bf(w2).float().t()
    y = ops.vit_mlp_fused_(x.clone(), *ops.vit_mlp_pack(nw, nb, w1, b1, w2, b2, ls), 1e-6)
    assert pre.abs().max().item() > 30
    assert (y - ref).abs().max().item() < 5e-2  # (an unclamped quartic is off by ~40 * |w2| * 512 terms)
    g = torch.randn(1, 3, 32, 32, device="cuda")
    w0, b0 = torch.randn(32, 3, device="cuda"), torch.randn(32, device="cuda")
    b0[::2] += 40.0
    b0[1::2] -= 40.0
    w3, b3 = torch.randn(32, 32, device="cuda") * 0.05, torch.randn(32, device="cuda")
    proj = ops.jbu_range_proj(g, w0, b0, w3, b3)
    hid = F.gelu(torch.einsum("bchw,nc->bhwn", g, w0) + b0)
    refp = hid @ w3.t() + b3
    assert (proj - refp).abs().max().item() < 2e-2 * refp.abs().max().item()


def test_f16_stream_kernels(ops):
    """IEEE-half forms used by LoftUp's inference stream: LayerNorm (bf16 / f16 / f32 in, f16 out), the dense GEMM with its
    bias, bias + GELU and residual epilogues, head_dim-128 / 256 attention -- each against fp32 on the half-rounded
    operands, and each clearly tighter than its bf16 twin."""
    torch.manual_seed(12)
    M, K, N = 1000, 448, 512
    x = torch.randn(M, K, device="cuda") * 3 + 0.5
    g, b = torch.randn(K, device="cuda") * 0.3 + 1, torch.randn(K, device="cuda") * 0.2
    ref = F.layer_norm(x, (K,), g, b, 1e-5)
    for xin in (x, x.half(), bf(x)):
        y = ops.layernorm(xin.contiguous(), g, b, 1e-5, out_dtype=torch.float16)
        assert y.dtype == torch.float16
        tol = 2e-3 if xin.dtype != BF else 3e-2
        assert (y.float() - F.layer_norm(xin.float(), (K,), g, b, 1e-5)).abs().max().item() < tol
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / math.sqrt(K)
    bias = torch.randn(N, device="cuda")
    ah, wh = a.half(), w.half()
    exact = a @ w.t() + bias
    for act, fn in ((None, lambda t: t), ("gelu", F.gelu)):
        y16 = ops.linear(ah, wh, bias, act)
        yb = ops.linear(bf(a), bf(w), bias, act)
        assert y16.dtype == torch.float16
        e16, eb = (y16.float() - fn(exact)).abs().max().item(), (yb.float() - fn(exact)).abs().max().item()
        assert (y16.float() - fn(ah.float() @ wh.float().t() + bias)).abs().max().item() < 4e-3
        assert e16 < eb / 3, (e16, eb)
    res = torch.randn(M, N, device="cuda")
    yr = ops.linear_axpy_res(ah, wh, bias, res.half(), 0.5)
    assert (yr.float() - (res.half().float() + 0.5 * (ah.float() @ wh.float().t() + bias))).abs().max().item() < 6e-3
    for hd, real in ((128, 101), (256, 197)):
        B, Lq, Lk, H = 2, 300, 260, 2
        def mk(L):
            t = torch.randn(B, L, H, hd, device="cuda")
            t[..., real:] = 0
            return t
        q, k, v = mk(Lq), mk(Lk), mk(Lk)
        o16 = ops.attention(q.half(), k.half(), v.half(), real ** -0.5)
        ob = ops.attention(bf(q), bf(k), bf(v), real ** -0.5)
        p = ((q.permute(0, 2, 1, 3) @ k.permute(0, 2, 3, 1)) * real ** -0.5).softmax(-1)
        refo = (p @ v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
        e16, eb = (o16.float() - refo).abs().max().item(), (ob.float() - refo).abs().max().item()
        assert o16.dtype == torch.float16 and e16 < 3e-3 and e16 < eb / 2, (hd, e16, eb)


def test_f16_trunk_kernels(ops):
    """The ViT trunk's half-precision inference stream: packed-qkv attention on IEEE half (base-2 logit form), the fp32
    residual update from half operands, and the saturating half GELU epilogue (no inf where bf16 would still be finite)."""
    torch.manual_seed(21)
    B, L, heads = 2, 517, 3
    D = heads * 64
    qkv = torch.randn(B * L, 3 * D, device="cuda")
    qkv[:, :D] *= ops.ATTENTION_LOGIT2_SCALE
    qh = qkv.half()
    out = ops.attention_packed_qkv(qh, B, L, heads, None, q_logit2=True)
    assert out.dtype == torch.float16
    q, k, v = qh.float().view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = ((q @ k.transpose(-2, -1)) * math.log(2.0)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B * L, D)
    e16 = (out.float() - ref).abs().max().item()
    outb = ops.attention_packed_qkv(bf(qkv), B, L, heads, None, q_logit2=True)
    qb, kb, vb = bf(qkv).float().view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    refb = (((qb @ kb.transpose(-2, -1)) * math.log(2.0)).softmax(-1) @ vb).transpose(1, 2).reshape(B * L, D)
    eb = (outb.float() - refb).abs().max().item()
    assert e16 < 2e-3 and e16 < eb / 2, (e16, eb)
    M, K, N = 700, 384, 384
    x = torch.randn(M, N, device="cuda")
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / math.sqrt(K)
    bias, gamma = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    y = ops.linear_residual_(x.clone(), a.half(), w.half(), bias, gamma)
    assert (y - (x + gamma * (a.half().float() @ w.half().float().t() + bias))).abs().max().item() < 2e-3
    big = torch.full((64, 64), 300.0, device="cuda")
    h = ops.linear(big.half(), torch.full((64, 64), 200.0, device="cuda").half(), None, "gelu")  # 64 * 300 * 200 = 3.8e6
    assert torch.isfinite(h.float()).all() and h.float().max().item() == 65504.0


@pytest.mark.parametrize("M", [300, 4096 + 77])
@pytest.mark.parametrize("act", [None, "gelu"])
def test_layernorm_folded_into_gemm(ops, M, act):
    """LoftUp's half stream (csrc/gemm.hip, EpAxpyResStats / EpLnFold): the residual GEMM emits per-row partial sums of what
    it stores, the next GEMM multiplies the RAW rows by W diag(g) and applies rstd * (acc - mean * s) + (c + W b) in its
    epilogue -- against LayerNorm + Linear (+ GELU) in fp32 on the same 16-bit-rounded rows.  404 real channels padded to
    448 (zero columns), ragged M."""
    torch.manual_seed(M)
    c, cp, Kin, N = 404, 448, 512, 384
    H16 = torch.float16
    A = torch.randn(M, Kin, device="cuda").to(H16)
    W0 = torch.zeros(cp, Kin, device="cuda")
    W0[:c] = torch.randn(c, Kin, device="cuda") / math.sqrt(Kin)
    b0 = torch.zeros(cp, device="cuda")
    b0[:c] = torch.randn(c, device="cuda")
    res = torch.zeros(M, cp, device="cuda")
    res[:, :c] = torch.randn(M, c, device="cuda") * 2 + 0.7            # (a mean far from zero: the correction term matters)
    x, stats = ops.linear_axpy_res_stats(A, W0.to(H16), b0, res.to(H16), 1.0)
    x_ref = res.to(H16).float() + A.float() @ W0.to(H16).float().t() + b0
    assert (x.float() - x_ref).abs().max().item() < 2e-2 and torch.equal(x[:, c:], torch.zeros(M, cp - c, device="cuda", dtype=H16))
    xs = x.float()
    assert torch.allclose(stats.sum(0)[:, 0], xs.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(stats.sum(0)[:, 1], xs.pow(2).sum(1), rtol=1e-5, atol=1e-3)
    g, beta = 1 + 0.2 * torch.randn(c, device="cuda"), 0.3 * torch.randn(c, device="cuda")
    W1 = torch.zeros(N, cp, device="cuda")
    W1[:, :c] = torch.randn(N, c, device="cuda") / math.sqrt(c)
    b1 = torch.randn(N, device="cuda")
    wf = W1.clone()
    wf[:, :c] *= g
    wh = wf.to(H16)
    y = ops.linear_lnfold(x, stats, wh, wh.float().sum(1).contiguous(), (b1 + W1[:, :c] @ beta).contiguous(), c, 1e-5, act)
    ref = F.layer_norm(xs[:, :c], (c,), g, beta, 1e-5) @ W1[:, :c].t() + b1
    ref = F.gelu(ref) if act == "gelu" else ref
    err = (y.float() - ref).abs().max().item()
    print(f"lnfold M={M} act={act}: max err {err:.3g} (ref max {ref.abs().max().item():.3g})")
    assert err < 2e-2


@pytest.mark.parametrize("M,N", [(70000, 384), (1000, 384), (333, 128), (4097, 448), (129, 500), (2050, 64)])
def test_lnfold_gemm_with_output_layernorm(ops, M, N):
    """LoftUp's tail as ONE GEMM (csrc/gemm.hip, EpLnFoldLayerNorm): LayerNorm_D -> Linear -> LayerNorm_N, the second
    LayerNorm from the accumulators of a tile that spans the output row (per-wave partial sums through LDS) -- against the
    three ops in fp32 on the same 16-bit-rounded input rows.  Full-row tile configurations for N <= 128 / 384 / 448 / 512,
    ragged M and N, a row mean far from zero on both sides."""
    torch.manual_seed(M + N)
    c, cp, Kin = 404, 448, 448
    H16 = torch.float16
    A = torch.randn(M, Kin, device="cuda").to(H16)
    W0 = torch.zeros(cp, Kin, device="cuda")
    W0[:c] = torch.randn(c, Kin, device="cuda") / math.sqrt(Kin)
    b0 = torch.zeros(cp, device="cuda")
    b0[:c] = torch.randn(c, device="cuda")
    res = torch.zeros(M, cp, device="cuda")
    res[:, :c] = torch.randn(M, c, device="cuda") * 2 + 0.7
    x, stats = ops.linear_axpy_res_stats(A, W0.to(H16), b0, res.to(H16), 1.0)
    xs = x.float()
    g, beta = 1 + 0.2 * torch.randn(c, device="cuda"), 0.3 * torch.randn(c, device="cuda")
    W1 = torch.zeros(N, cp, device="cuda")
    W1[:, :c] = torch.randn(N, c, device="cuda") / math.sqrt(c)
    b1 = torch.randn(N, device="cuda") + 1.5
    g2, beta2 = 1 + 0.2 * torch.randn(N, device="cuda"), 0.3 * torch.randn(N, device="cuda")
    wf = W1.clone()
    wf[:, :c] *= g
    wh = wf.to(H16)
    y = ops.linear_lnfold_layernorm(x, stats, wh, wh.float().sum(1).contiguous(), (b1 + W1[:, :c] @ beta).contiguous(), c, 1e-5,
                                    g2, beta2, 1e-6)
    mid = F.layer_norm(xs[:, :c], (c,), g, beta, 1e-5) @ W1[:, :c].t() + b1
    ref = F.layer_norm(mid, (N,), g2, beta2, 1e-6)
    err = (y.float() - ref).abs().max().item()
    two = ops.layernorm(ops.linear_lnfold(x, stats, wh, wh.float().sum(1).contiguous(), (b1 + W1[:, :c] @ beta).contiguous(), c, 1e-5),
                        g2, beta2, 1e-6, out_dtype=H16)
    print(f"lnfold+layernorm M={M} N={N}: max err {err:.3g} (two-kernel route {(two.float() - ref).abs().max().item():.3g}; ref max {ref.abs().max().item():.3g})")
    assert y.shape == (M, N) and torch.isfinite(y.float()).all()
    assert err < 2e-2


@pytest.mark.parametrize("B,H,W,C,N", [(2, 40, 56, 64, 448), (1, 33, 17, 128, 448), (1, 16, 16, 64, 384), (1, 21, 50, 64, 128)])
def test_conv3x3_relu_with_row_statistics(ops, B, H, W, C, N):
    """csrc/gemm.hip, EpBiasActStats: the half-precision 3x3 conv + ReLU whose epilogue also emits per-pixel partial sums of
    the values it stores (LoftUp's second convolution -> the LN-folded query projection).  Output bit-identical to the plain
    conv; the partials summed over their slots equal the row sums / sums of squares of the stored map; 128- and 192-channel
    block kernels, interior and ragged tiles (N = 448 = 3.5 blocks of 128: the last wave column group writes zeros)."""
    torch.manual_seed(B * H + W)
    H16 = torch.float16
    x = torch.randn(B, H, W, C, device="cuda").to(H16)
    w = (torch.randn(N, 9 * C, device="cuda") / math.sqrt(9 * C)).to(H16)
    b = torch.randn(N, device="cuda")
    y0 = ops.conv3x3(x, w, b, "relu")
    y, st = ops.conv3x3_relu_stats(x, w, b)
    assert torch.equal(y, y0)
    ys = y.float().view(-1, N)
    assert st.shape[1:] == (B * H * W, 2) and torch.isfinite(st).all()
    assert torch.allclose(st.sum(0)[:, 0], ys.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st.sum(0)[:, 1], ys.pow(2).sum(1), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("B,H,W,p,chans,Kpad", [(2, 56, 84, 14, (3, 1, 2), 1216), (1, 448, 448, 14, (3, 0, 2), 1024), (2, 48, 32, 16, (3, 1, 0), 1024),
                                                (1, 28, 42, 14, (0, 1, 2), 640), (1, 15, 10, 5, (3, 1, 2), 152)])
def test_patchify_matrix(ops, B, H, W, p, chans, Kpad):
    """isp_patchify_fwd: row t of sample b = [img | prev | click maps] patches flattened (c, i, j) as a Conv2d weight is, bf16,
    zero padded to Kpad -- both kernels (a token row per block through LDS; the per-token gather for shapes it does not take:
    p * p % 4 != 0) against torch unfold, bit for bit."""
    torch.manual_seed(H + W)
    n_img, n_prev, n_maps = chans
    mk = lambda n: torch.randn(B, n, H, W, device="cuda") if n else None
    img, prev, maps = mk(n_img), mk(n_prev), mk(n_maps)
    A = ops.patchify(img, prev, maps, p, Kpad)
    x = torch.cat([t for t in (img, prev, maps) if t is not None], 1)
    ref = torch.nn.functional.unfold(x, kernel_size=p, stride=p).transpose(1, 2).reshape(B * (H // p) * (W // p), -1)  # [tokens, C*p*p], (c, i, j) order
    assert A.shape == (B * (H // p) * (W // p), Kpad) and A.dtype == torch.bfloat16
    assert torch.equal(A[:, :ref.shape[1]], ref.to(torch.bfloat16))
    assert torch.equal(A[:, ref.shape[1]:], torch.zeros_like(A[:, ref.shape[1]:]))


# ---- first head conv through the bilinear resize (csrc/conv_bilinear.hip)
def _conv_of_bilinear_ref(x_nhwc, wconv, bias, H, W):
    """iseg_probe_model.py:120-129 + conv_heads.py:59-73 in torch fp32: resize, then conv + ReLU."""
    y = F.interpolate(x_nhwc.float().permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True)
    return F.relu(F.conv2d(y, wconv, bias, padding=1)).permute(0, 2, 3, 1)


@pytest.mark.parametrize("geom", [
    (2, 4, 4, 56, 56, 64, 64),        # tiny fixture geometry: partial 16 x 16 tiles, every tile touches a border
    (1, 8, 8, 56, 56, 128, 64),       # LiFT's 2h x 2w map at the tiny size (x7)
    (2, 16, 16, 224, 224, 64, 128),   # x14.9, interior tiles
    (1, 32, 24, 224, 168, 64, 64),    # x7.2 / x7.3, non-square
    (1, 9, 7, 75, 61, 64, 64),        # odd sizes, ragged tiles on both axes
])
def test_conv3x3_of_bilinear_vs_torch(ops, geom):
    B, h, w, H, W, C, N = geom
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, h, w, C, generator=g).cuda()
    wconv = (torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    bias = (0.2 * torch.randn(N, generator=g)).cuda()
    assert ops.conv3x3_of_bilinear_supported(h, w, H, W, N)
    ref = _conv_of_bilinear_ref(x, wconv, bias, H, W)
    wz = wconv.permute(2, 3, 0, 1).reshape(9 * N, C)
    # fp32 tap planes: the blend itself is exact fp32 arithmetic -> 1e-5 of the torch result
    z32 = (x.view(-1, C) @ wz.t()).contiguous()
    out32 = ops.conv3x3_of_bilinear_blend(z32, bias, B, h, w, H, W, N, relu=True, out_dtype=torch.float32)
    assert (out32 - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # product route: half operands, half tap planes, half output
    xh, wh = x.half(), wz.half().contiguous()
    refh = _conv_of_bilinear_ref(xh, wh.float().view(3, 3, N, C).permute(2, 3, 0, 1).contiguous(), bias, H, W)
    z16 = ops.linear(xh.view(-1, C), wh)
    assert z16.dtype == torch.float16 and z16.shape == (B * h * w, 9 * N)
    for odt in (torch.float16, torch.bfloat16):
        out = ops.conv3x3_of_bilinear_blend(z16, bias, B, h, w, H, W, N, relu=True, out_dtype=odt)
        tol = (2e-3 if odt == torch.float16 else 1e-2) * max(1.0, refh.abs().max().item())
        assert (out.float() - refh).abs().max().item() <= tol
    # no activation, no bias
    lin = ops.conv3x3_of_bilinear_blend(z32, None, B, h, w, H, W, N, relu=False, out_dtype=torch.float32)
    y = F.interpolate(x.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True)
    assert (lin - F.conv2d(y, wconv, None, padding=1).permute(0, 2, 3, 1)).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv3x3_of_bilinear_rejects_small_factors(ops):
    from isegprobe_amd._lib import IspError
    assert not ops.conv3x3_of_bilinear_supported(32, 32, 64, 64, 64)      # x2: footprint of a tile exceeds 5 x 5
    assert not ops.conv3x3_of_bilinear_supported(512, 512, 448, 448, 64)  # down-scaling (FeatUp's 512 -> 448)
    assert not ops.conv3x3_of_bilinear_supported(4, 4, 56, 56, 48)        # channel block
    z = torch.zeros(32 * 32, 9 * 64, device="cuda", dtype=torch.float16)
    with pytest.raises(IspError):
        ops.conv3x3_of_bilinear_blend(z, None, 1, 32, 32, 64, 64, 64)


@pytest.mark.parametrize("B,L,heads", [(2, 17, 2), (1, 257, 6), (2, 1025, 6), (3, 129, 1)])
def test_attention_packed_f32_vs_torch(ops, B, L, heads):
    """The checking mode's exact-fp32 attention (csrc/attention_f32.hip, attention.py:54-71) against torch in fp64: ragged
    last key tile (L = 17, 129, 257, 1025), several query blocks per (batch, head), a score spike that moves the running maximum
    late in the key sequence."""
    torch.manual_seed(L)
    D = heads * 64
    qkv = torch.randn(B * L, 3 * D, device="cuda")
    qkv[L - 2, D:D + 64] *= 6.0  # one key of (batch 0, head 0) far above the rest, in the last tile
    out = ops.attention_packed_qkv_f32(qkv, B, L, heads, 64 ** -0.5)
    q, k, v = (qkv.double().view(B, L, 3, heads, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax((q * 64 ** -0.5) @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * L, D)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-6 * max(1.0, ref.abs().max().item()), err
