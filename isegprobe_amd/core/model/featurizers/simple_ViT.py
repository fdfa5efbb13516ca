"""Simple ViT click encoder (reference core/model/featurizers/simple_ViT.py:18-155, adapted there
from lucidrains/vit-pytorch), the ``embed_coords_type == "simple_vit"`` alternative to the conv
PatchEmbed: (p1 p2 c)-ordered patches -> LN -> Linear -> LN -> + 2-D sincos pos-emb -> pre-norm
transformer (bias-free qkv / out projections, dim_head 64) -> LN.

Parameter container with the reference's state-dict layout; forward = HIP launches (patchify,
LayerNorm, bf16 GEMMs with fused epilogues, fused attention).  The patch flattening order differs
from the kernel's (c, p1, p2): the first LayerNorm's affine and the Linear's input columns are
permuted at pack time instead (LayerNorm statistics are permutation invariant).
As a *trainable* click encoder (models/sbd/dinov2/simple-vit_noup.py) every parameter gets its gradient from
``_SimpleViTFn``: the activation chain of the frozen-trunk backward (attention / LayerNorm / GELU backward, transposed
GEMMs) plus the weight gradients -- pixel-reduction GEMMs for the Linear layers, column reductions for biases and
LayerNorm affines."""
import torch
import torch.nn as nn

from .... import hip_ops as ops
from .._tensor import BF16, PackedCache


def _pad64(n):
    return (n + 63) // 64 * 64


def posemb_sincos_2d(h, w, dim, temperature: int = 10000, dtype=torch.float32):
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    assert (dim % 4) == 0, "feature dimension must be multiple of 4 for sincos emb"
    omega = 1.0 / (temperature ** (torch.arange(dim // 4) / (dim // 4 - 1)))
    y, x = y.flatten()[:, None] * omega[None, :], x.flatten()[:, None] * omega[None, :]
    return torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).type(dtype)


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.heads = heads
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, dim_head * heads * 3, bias=False)
        self.to_out = nn.Linear(dim_head * heads, dim, bias=False)


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim))


class _Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([nn.ModuleList([_Attention(dim, heads, dim_head), _FeedForward(dim, mlp_dim)])
                                     for _ in range(depth)])


class SimpleViTFeaturizer(nn.Module):
    def __init__(self, *, image_size, patch_size, dim: int, depth: int, heads: int, mlp_dim: int, channels: int = 3,
                 dim_head: int = 64) -> None:
        super().__init__()
        pair = lambda t: tuple(t) if isinstance(t, (tuple, list)) else (t, t)
        (ih, iw), (ph, pw) = pair(image_size), pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        if ph != pw or dim_head != 64:
            raise NotImplementedError("square patches and dim_head 64 only")
        patch_dim = channels * ph * pw
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim),
                                                nn.LayerNorm(dim))
        self.pos_embedding = posemb_sincos_2d(h=ih // ph, w=iw // pw, dim=dim)
        self.transformer = _Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.image_hw, self.patch_hw = (ih, iw), (ph, pw)
        self.channels, self.dim, self.heads = channels, dim, heads
        self._packed = PackedCache()
        self._pos = {}

    def packed(self):
        def build():
            C, p = self.channels, self.patch_hw[0]
            K = C * p * p
            # kernel order k = c*p*p + i*p + j ; reference order r = (i*p + j)*C + c
            c, i, j = torch.meshgrid(torch.arange(C), torch.arange(p), torch.arange(p), indexing="ij")
            ref_of_kernel = ((i * p + j) * C + c).reshape(-1)  # reference column feeding kernel column k
            ln0, lin, ln1 = self.to_patch_embedding[1], self.to_patch_embedding[2], self.to_patch_embedding[3]
            dev = lin.weight.device
            f = lambda t: t.detach().float().contiguous()
            Kp = _pad64(K)
            w = torch.zeros(self.dim, Kp, device=dev)
            w[:, :K] = lin.weight.detach().float()[:, ref_of_kernel.to(dev)]
            P = dict(K=K, Kp=Kp, ln0_w=f(ln0.weight[ref_of_kernel.to(dev)]), ln0_b=f(ln0.bias[ref_of_kernel.to(dev)]),
                     lin_w=w.to(BF16).contiguous(), lin_b=f(lin.bias), ln1_w=f(ln1.weight), ln1_b=f(ln1.bias), layers=[])
            for att, ff in self.transformer.layers:
                hid = ff.net[1].weight.shape[0]
                P["layers"].append(dict(
                    n_w=f(att.norm.weight), n_b=f(att.norm.bias),
                    qkv_w=att.to_qkv.weight.detach().to(BF16).contiguous(), out_w=att.to_out.weight.detach().to(BF16).contiguous(),
                    f_nw=f(ff.net[0].weight), f_nb=f(ff.net[0].bias),
                    f1_w=ff.net[1].weight.detach().to(BF16).contiguous(), f1_b=f(ff.net[1].bias),
                    f2_w=ff.net[3].weight.detach().to(BF16).contiguous(), f2_b=f(ff.net[3].bias)))
            P["tn_w"], P["tn_b"] = f(self.transformer.norm.weight), f(self.transformer.norm.bias)
            return P
        return self._packed.get(self._packed.tensors_of(self.parameters), build)

    # ---- parameters in a fixed order (the order _SimpleViTFn returns their gradients in)
    def _param_list(self):
        ln0, lin, ln1 = self.to_patch_embedding[1], self.to_patch_embedding[2], self.to_patch_embedding[3]
        ps = [ln0.weight, ln0.bias, lin.weight, lin.bias, ln1.weight, ln1.bias]
        for att, ff in self.transformer.layers:
            ps += [att.norm.weight, att.norm.bias, att.to_qkv.weight, att.to_out.weight, ff.net[0].weight, ff.net[0].bias,
                   ff.net[1].weight, ff.net[1].bias, ff.net[3].weight, ff.net[3].bias]
        return ps + [self.transformer.norm.weight, self.transformer.norm.bias]

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _SimpleViTFn.apply(img, self, *self._param_list())
        return self._run(img, None)

    def _run(self, img, save):
        P = self.packed()
        B, C, H, W = img.shape
        p = self.patch_hw[0]
        T = (H // p) * (W // p)
        eps = 1e-5
        A = ops.patchify(img.float().contiguous(), None, None, p, P["Kp"])             # [B*T, Kp] bf16, (c,i,j) order
        A_ln = ops.layernorm(A, P["ln0_w"], P["ln0_b"], eps, D=P["K"], ld_out=P["Kp"])  # LN over the 588 real columns
        y = ops.linear(A_ln, P["lin_w"], P["lin_b"], None, out_dtype=torch.float32)
        x = ops.layernorm(y, P["ln1_w"], P["ln1_b"], eps, out_dtype=torch.float32)
        key = (H // p, W // p, str(img.device))
        if key not in self._pos:
            self._pos[key] = posemb_sincos_2d(H // p, W // p, self.dim).to(img.device).contiguous()
        ops.token_add_(x, self._pos[key].unsqueeze(0).expand(B, -1, -1).contiguous(), B, T, has_cls=False)
        if save is not None:
            save.update(A=A, A_ln=A_ln, y=y, layers=[], geom=(B, T))
        for L in P["layers"]:
            x_in = x.clone() if save is not None else None
            a = ops.layernorm(x, L["n_w"], L["n_b"], eps)
            qkv = ops.linear(a, L["qkv_w"], None)                                      # [B*T, 3*heads*64], (3, h, d) packed
            if save is None:
                att = ops.attention_packed_qkv(qkv, B, T, self.heads, 64 ** -0.5)
            else:
                att, lse = ops.attention_packed_qkv_lse(qkv, B, T, self.heads, 64 ** -0.5)
            ops.linear_residual_(x, att, L["out_w"], None, None)
            a2 = ops.layernorm(x, L["f_nw"], L["f_nb"], eps)
            if save is None:
                f1 = ops.linear(a2, L["f1_w"], L["f1_b"], "gelu")
            else:
                x_mid = x.clone()
                f1, pre = ops.linear_gelu_save(a2, L["f1_w"], L["f1_b"])
                save["layers"].append(dict(x_in=x_in, a=a, qkv=qkv, att=att, lse=lse, x_mid=x_mid, a2=a2, pre=pre, hid=f1))
            ops.linear_residual_(x, f1, L["f2_w"], L["f2_b"], None)
        out = ops.layernorm(x, P["tn_w"], P["tn_b"], eps, out_dtype=torch.float32)
        if save is not None:
            save["x_fin"] = x
        return out.view(B, T, self.dim)

    def _backward(self, saved, g_out):
        """Gradients of every parameter (in _param_list order) given d loss / d tokens [B,T,dim] fp32."""
        P = self.packed()
        B, T = saved["geom"]
        eps, heads = 1e-5, self.heads
        t = lambda w: w.float().t().contiguous().to(BF16)  # data-gradient copy of a [N,K] weight
        gy = g_out.reshape(B * T, self.dim).to(BF16).contiguous()
        d_tn = ops.layernorm_wgrad(saved["x_fin"], gy, eps)
        gx, g16 = ops.layernorm_bwd(saved["x_fin"], gy, P["tn_w"], eps)
        layer_grads = []
        for L, S in zip(reversed(P["layers"]), reversed(saved["layers"])):
            d_f2w, d_f2b = ops.linear_wgrad(g16, S["hid"])
            g_pre = ops.linear_mul_dgelu(g16, t(L["f2_w"]), S["pre"])
            d_f1w, d_f1b = ops.linear_wgrad(g_pre, S["a2"])
            g_a2 = ops.linear(g_pre, t(L["f1_w"]))
            d_fn = ops.layernorm_wgrad(S["x_mid"], g_a2, eps)
            gx, g16 = ops.layernorm_bwd(S["x_mid"], g_a2, L["f_nw"], eps, gx=gx)
            d_outw, _ = ops.linear_wgrad(g16, S["att"], want_bias=False)
            g_att = ops.linear(g16, t(L["out_w"]))
            g_qkv = ops.attention_packed_qkv_bwd(S["qkv"], S["att"], g_att, S["lse"], B, T, heads, 64 ** -0.5)
            d_qkvw, _ = ops.linear_wgrad(g_qkv, S["a"], want_bias=False)
            g_a = ops.linear(g_qkv, t(L["qkv_w"]))
            d_n = ops.layernorm_wgrad(S["x_in"], g_a, eps)
            gx, g16 = ops.layernorm_bwd(S["x_in"], g_a, L["n_w"], eps, gx=gx)
            layer_grads.append([d_n[0], d_n[1], d_qkvw, d_outw, d_fn[0], d_fn[1], d_f1w, d_f1b, d_f2w, d_f2b])
        # patch embedding: x0 = LN1(y) + pos, y = LN0(A) lin_w^T + lin_b   (kernel column order k = c*p*p + i*p + j)
        d_ln1 = ops.layernorm_wgrad(saved["y"], g16, eps)
        _, g_y = ops.layernorm_bwd(saved["y"], g16, P["ln1_w"], eps)
        d_linw_k, d_linb = ops.linear_wgrad(g_y, saved["A_ln"])            # [dim, Kp]
        g_Aln = ops.linear(g_y, t(P["lin_w"]))                            # [M, Kp]
        d_ln0_k = ops.layernorm_wgrad(saved["A"], g_Aln, eps, D=P["K"])
        # back to the reference's (p1 p2 c) column order: reference column r sits at kernel column kernel_of_ref[r]
        C, p = self.channels, self.patch_hw[0]
        i, j, c = torch.meshgrid(torch.arange(p), torch.arange(p), torch.arange(C), indexing="ij")
        kernel_of_ref = (c * p * p + i * p + j).reshape(-1).to(gy.device)
        grads = [d_ln0_k[0][kernel_of_ref], d_ln0_k[1][kernel_of_ref], d_linw_k[:, kernel_of_ref].contiguous(), d_linb,
                 d_ln1[0], d_ln1[1]]
        for lg in reversed(layer_grads):
            grads += lg
        return grads + [d_tn[0], d_tn[1]]

    def reshape_feats_to_patches(self, feats: torch.Tensor) -> torch.Tensor:
        B, _, c = feats.shape
        return feats.transpose(1, 2).reshape(B, c, self.image_hw[0] // self.patch_hw[0], self.image_hw[1] // self.patch_hw[1])


class _SimpleViTFn(torch.autograd.Function):
    """The whole click encoder as one autograd node (img carries no gradient)."""

    @staticmethod
    def forward(ctx, img, module, *params):
        saved = {}
        out = module._run(img.detach(), saved)
        ctx.module, ctx.saved, ctx.shapes = module, saved, [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, g_out):
        grads = ctx.module._backward(ctx.saved, g_out.float().contiguous())
        ctx.saved = None
        return (None, None) + tuple(g.reshape(s) for g, s in zip(grads, ctx.shapes))
