"""Simple ViT click encoder (reference core/model/featurizers/simple_ViT.py:18-155, adapted there
from lucidrains/vit-pytorch), the ``embed_coords_type == "simple_vit"`` alternative to the conv
PatchEmbed: (p1 p2 c)-ordered patches -> LN -> Linear -> LN -> + 2-D sincos pos-emb -> pre-norm
transformer (bias-free qkv / out projections, dim_head 64) -> LN.

Parameter container with the reference's state-dict layout; forward = HIP launches (patchify,
LayerNorm, bf16 GEMMs with fused epilogues, fused attention).  The patch flattening order differs
from the kernel's (c, p1, p2): the first LayerNorm's affine and the Linear's input columns are
permuted at pack time instead (LayerNorm statistics are permutation invariant).
Inference only: as a *trainable* click encoder it needs a transformer backward, which is not built."""
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

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("training the simple_vit click encoder needs a transformer backward (not built); "
                                      "wrap the call in torch.no_grad()")
        P = self.packed()
        B, C, H, W = img.shape
        p = self.patch_hw[0]
        T = (H // p) * (W // p)
        eps = 1e-5
        A = ops.patchify(img.float().contiguous(), None, None, p, P["Kp"])             # [B*T, Kp] bf16, (c,i,j) order
        A = ops.layernorm(A, P["ln0_w"], P["ln0_b"], eps, D=P["K"], ld_out=P["Kp"])    # LN over the 588 real columns
        y = ops.linear(A, P["lin_w"], P["lin_b"], None, out_dtype=torch.float32)
        x = ops.layernorm(y, P["ln1_w"], P["ln1_b"], eps, out_dtype=torch.float32)
        key = (H // p, W // p, str(img.device))
        if key not in self._pos:
            self._pos[key] = posemb_sincos_2d(H // p, W // p, self.dim).to(img.device).contiguous()
        ops.token_add_(x, self._pos[key].unsqueeze(0).expand(B, -1, -1).contiguous(), B, T, has_cls=False)
        for L in P["layers"]:
            a = ops.layernorm(x, L["n_w"], L["n_b"], eps)
            qkv = ops.linear(a, L["qkv_w"], None)                                      # [B*T, 3*heads*64], (3, h, d) packed
            att = ops.attention_packed_qkv(qkv, B, T, self.heads, 64 ** -0.5)
            ops.linear_residual_(x, att, L["out_w"], None, None)
            f1 = ops.linear(ops.layernorm(x, L["f_nw"], L["f_nb"], eps), L["f1_w"], L["f1_b"], "gelu")
            ops.linear_residual_(x, f1, L["f2_w"], L["f2_b"], None)
        out = ops.layernorm(x, P["tn_w"], P["tn_b"], eps, out_dtype=torch.float32)
        return out.view(B, T, self.dim)

    def reshape_feats_to_patches(self, feats: torch.Tensor) -> torch.Tensor:
        B, _, c = feats.shape
        return feats.transpose(1, 2).reshape(B, c, self.image_hw[0] // self.patch_hw[0], self.image_hw[1] // self.patch_hw[1])
