"""GPU: backward kernels against torch autograd (fp32 reference of the same op)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from isegprobe_amd import hip_ops
    return hip_ops


def rel(a, b):
    return (a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-12)


@pytest.mark.parametrize("M,N,J", [(64, 128, 128), (1000, 64, 640), (6272, 128, 192), (4096, 384, 384)])
def test_tn_gemm_dense(ops, M, N, J):
    torch.manual_seed(M)
    P = torch.randn(M, N, device="cuda").to(BF)
    Q = torch.randn(M, J, device="cuda").to(BF)
    out = torch.zeros(N, J, device="cuda")
    ops.tn_gemm_atomic(P, Q, out)
    ref = P.float().t() @ Q.float()
    assert rel(out, ref) < 5e-3


def test_tn_gemm_asymmetric_layout(ops):
    """P = shifted identity-like pattern against an asymmetric Q: catches transposed fragments."""
    M, N, J = 128, 128, 128
    P = torch.zeros(M, N, device="cuda")
    P[torch.arange(M), (torch.arange(M) * 7 + 3) % N] = 1.0
    Q = ((torch.arange(M * J, device="cuda").float().reshape(M, J) % 253) / 8).to(BF)
    out = torch.zeros(N, J, device="cuda")
    ops.tn_gemm_atomic(P.to(BF), Q, out)
    assert torch.equal(out, P.t() @ Q.float())


@pytest.mark.parametrize("B,H,W,C,N", [(2, 20, 24, 64, 64), (1, 37, 45, 128, 192), (2, 56, 56, 128, 128)])
def test_conv_wgrad_dgrad_vs_autograd(ops, B, H, W, C, N):
    torch.manual_seed(C + H)
    x = torch.randn(B, C, H, W, device="cuda").to(BF).float().requires_grad_(True)
    w = (torch.randn(N, C, 3, 3, device="cuda") / math.sqrt(9 * C)).to(BF).float().requires_grad_(True)
    bias = torch.randn(N, device="cuda", requires_grad=True)
    y = F.relu(F.conv2d(x, w, bias, padding=1))
    gy = torch.randn_like(y).to(BF).float()
    y.backward(gy)
    # HIP: mask, bias grad, weight grad (9 tap GEMMs), data grad (conv with rotated weights)
    x_n = x.detach().permute(0, 2, 3, 1).contiguous().to(BF)
    y_n = y.detach().permute(0, 2, 3, 1).contiguous().to(BF)
    gy_n = gy.permute(0, 2, 3, 1).contiguous().to(BF)
    g, db = ops.relu_mask_colsum(gy_n, y_n)
    assert rel(db, bias.grad) < 5e-3
    dw = torch.zeros(N, 9 * C, device="cuda")
    M = B * H * W
    for t in range(9):
        ops.tn_gemm_atomic(g.view(M, N), x_n.view(M, C), dw[:, t * C:(t + 1) * C], shift=(H, W, t // 3 - 1, t % 3 - 1))
    dw_ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9 * C)
    assert rel(dw, dw_ref) < 1e-2
    w_rot = w.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * N).to(BF).contiguous()  # [C][ky][kx][N]
    dx = ops.conv3x3(g, w_rot, None, None)
    assert rel(dx.permute(0, 3, 1, 2), x.grad) < 1e-2


def test_classifier_bwd(ops):
    M, C = 5000, 128
    x = F.relu(torch.randn(M, C, device="cuda")).to(BF)
    w = torch.randn(C, device="cuda")
    gl = torch.randn(M, device="cuda")
    dx, dw, db = ops.classifier_bwd(gl, x, w)
    ref_dx = (x.float() > 0) * gl[:, None] * w[None, :]
    assert rel(dx, ref_dx) < 1e-2
    assert rel(dw, (gl[:, None] * x.float()).sum(0)) < 1e-3
    assert abs(db.item() - gl.sum().item()) < 1e-2


@pytest.mark.parametrize("shape", [(2, 4, 5, 56, 70, 64), (1, 16, 16, 224, 224, 128)])
def test_bilinear_bwd(ops, shape):
    B, h, w, H, W, C = shape
    x = torch.randn(B, C, h, w, device="cuda", requires_grad=True)
    y = F.interpolate(x, (H, W), mode="bilinear", align_corners=True)
    gy = torch.randn_like(y).to(BF).float()
    y.backward(gy)
    din = ops.resize_bilinear_nhwc_bwd(gy.permute(0, 2, 3, 1).contiguous().to(BF), h, w)
    assert rel(din.permute(0, 3, 1, 2), x.grad) < 1e-2
