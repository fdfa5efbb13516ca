"""LoftUp upsampler (reference core/model/upsamplers/LoftUp.py + loftup/loftup.py:16-177,
loftup/layers.py): Fourier image features -> 2 x (3x3 conv + BN + ReLU) at full resolution ->
2-layer cross-attention transformer (queries = every pixel, keys/values = LR tokens + sine PE)
-> 1x1 conv + channel LayerNorm.

``LoftUp`` / ``ChannelNorm`` / ``UpsamplerwithChannelNorm`` below are parameter containers with
the reference's state-dict layout (``upsampler.upsampler.*``, ``upsampler.channelnorm.norm.*``
under ``LoftUpUpsampler``).  The forward pass is HIP launches only:

  * channel dims are zero-padded to multiples of 64 (203 -> 256, 404 -> 448) and the 4 heads'
    dim 101 to 128, so every projection is a bf16 MFMA GEMM and the padding is exact;
  * eval-mode BatchNorm is folded into the 3x3 conv weights at pack time;
  * the cross-attention is the fused flash-style kernel (csrc/attention.hip, head_dim 128): the
    reference's nn.MultiheadAttention materialises a [B*4, H*W, h*w] weight tensor (3.3 GB per
    image and layer at 448^2, layers.py:182-198) -- here it never exists;
  * MinMaxScaler statistics are batch-global (per-shard under data parallelism, as in the
    reference's DDP).
Train-mode BatchNorm (the reference's ``net.train()``, trainer.py:214, also flips this frozen module's two
BatchNorm2d layers, loftup.py:58,63): with the module in training mode the two 3x3 convs run unfolded and are
followed by the batch-statistics BatchNorm kernels (running statistics updated as torch does).  Both layers see
the image only, so no gradient passes through them.
"""
import math
import os

import torch
import torch.nn as nn

from .... import hip_ops as ops
from ...utils.log import logger
from .._tensor import BF16, PackedCache, nchw_view, to_nhwc_bf16, pack_serial
from .._guidance_cache import GuidanceCache
from . import BaseUpsampler


LOFTUP_F16 = os.environ.get("ISEGPROBE_LOFTUP_F16", "1") != "0"  # IEEE-half inference stream (see _run)
LOFTUP_LNFOLD = os.environ.get("ISEGPROBE_LOFTUP_LNFOLD", "1") != "0"  # LayerNorms folded into the consuming GEMMs (half stream)
LOFTUP_Q0_STATS = os.environ.get("ISEGPROBE_LOFTUP_Q0_STATS", "1") != "0"  # row statistics from the second convolution's epilogue
LOFTUP_TAIL_FUSED = os.environ.get("ISEGPROBE_LOFTUP_TAIL_FUSED", "1") != "0"  # final channel LayerNorm in the 1x1 conv's epilogue


def _pad64(n):
    return (n + 63) // 64 * 64


# ------------------------------------------------------------------ parameter containers
class ChannelNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)


class _ChannelLN(nn.Module):  # loftup/layers.py:38-58 (eps 1e-6)
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps


class _Implicit(nn.Module):  # ImplicitFeaturizer with learn_bias=True
    def __init__(self, color_feats, n_freqs):
        super().__init__()
        self.n_freqs = n_freqs
        self.dim_multiplier = 5 if color_feats else 2
        self.biases = nn.Parameter(torch.randn(2, self.dim_multiplier, n_freqs))


class _CrossAttn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm_q = nn.LayerNorm(dim)
        self.norm_kv = nn.LayerNorm(dim)
        self.attention = nn.MultiheadAttention(embed_dim=dim, num_heads=heads)


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden), nn.GELU(), nn.Identity(),
                                 nn.Linear(hidden, dim), nn.Identity())


class _CATransformer(nn.Module):
    def __init__(self, dim, depth, heads, mlp_dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([nn.ModuleList([_CrossAttn(dim, heads), _FeedForward(dim, mlp_dim)])
                                     for _ in range(depth)])


class LoftUp(nn.Module):
    def __init__(self, dim, color_feats=True, n_freqs=20, num_heads=4, num_layers=2, num_conv_layers=1, lr_size=16,
                 lr_pe_type="sine"):
        super().__init__()
        if not color_feats or lr_pe_type != "sine":
            raise NotImplementedError("only the reference's configuration (colour feats, sine LR PE) is built")
        self.dim, self.n_freqs, self.num_heads, self.num_layers = dim, n_freqs, num_heads, num_layers
        start_dim = 5 * n_freqs * 2 + 3
        self.lr_pe_type = lr_pe_type
        self.lr_pe = _Implicit(False, 5)
        self.lr_pe_dim = 2 * 5 * 2
        c = dim + self.lr_pe_dim
        self.fourier_feat = nn.Sequential(nn.Identity(), _Implicit(True, n_freqs))
        self.first_conv = nn.Sequential(ChannelNorm(start_dim), nn.Conv2d(start_dim, c, 3, padding=1),
                                        nn.BatchNorm2d(c), nn.ReLU(inplace=True), nn.Conv2d(c, c, 3, padding=1),
                                        nn.BatchNorm2d(c), nn.ReLU(inplace=True))
        self.final_conv = nn.Sequential(nn.Conv2d(c, dim, kernel_size=1), _ChannelLN(dim))
        self.ca_transformer = _CATransformer(c, num_layers, num_heads, dim)


class UpsamplerwithChannelNorm(nn.Module):
    def __init__(self, upsampler, channelnorm):
        super().__init__()
        self.upsampler = upsampler
        self.channelnorm = channelnorm


def load_loftup_checkpoint(upsampler_path, n_dim, lr_pe_type="sine", lr_size=16):
    """loftup.py:152-177: split a LoftUp checkpoint into the ChannelNorm (``model.1.*``) and the
    upsampler (``upsampler.*``) parts.  With no readable path the random init is kept."""
    module = UpsamplerwithChannelNorm(LoftUp(n_dim, lr_pe_type=lr_pe_type, lr_size=16), ChannelNorm(n_dim))
    if upsampler_path and os.path.exists(str(upsampler_path)):
        ckpt = torch.load(upsampler_path, map_location="cpu")["state_dict"]
        module.channelnorm.load_state_dict({k.replace("model.1.", ""): v for k, v in ckpt.items() if "model.1" in k})
        module.upsampler.load_state_dict({k.replace("upsampler.", "", 1): v for k, v in ckpt.items()
                                          if k.startswith("upsampler")})
        logger.info(f"Loaded LoftUp checkpoint: {upsampler_path}")
    else:
        logger.info("LoftUpUpsampler: no checkpoint at upsampler_path, keeping random init")
    for p in module.parameters():
        p.requires_grad = False
    return module


# ------------------------------------------------------------------ the plugin
class LoftUpUpsampler(BaseUpsampler):
    def __init__(self, upsampler_path: str = None, n_dim: int = 384, lr_pe_type: str = "sine", lr_size: int = 16) -> None:
        super().__init__()
        self.upsampler = load_loftup_checkpoint(upsampler_path, n_dim, lr_pe_type, lr_size)
        self._packed = PackedCache()
        self._packed_train, self._packed_half = PackedCache(), PackedCache()
        self._gcache = GuidanceCache()
        self._pe_cache = {}

    def _hdp(self):
        lu = self.upsampler.upsampler
        hd = (lu.dim + lu.lr_pe_dim) // lu.num_heads
        return 64 if hd <= 64 else (128 if hd <= 128 else 256)

    def _bn_train(self):
        return self.upsampler.upsampler.first_conv[2].training

    def _cp(self):
        """Padded channel count of the pixel stream (n_dim + 20 PE channels, to a multiple of 64)."""
        lu = self.upsampler.upsampler
        return _pad64(lu.dim + lu.lr_pe_dim)

    # ---- weight packing (bf16, padded, BN folded unless the module is in training mode)
    def packed(self, train=False, half=False):
        """``half``: kernel-layout weights in IEEE half (from the fp32 parameters) for the half-precision inference stream."""
        wd = ops.F16 if half else BF16

        def build():
            lu, cn = self.upsampler.upsampler, self.upsampler.channelnorm
            dev = cn.norm.weight.device
            C, heads = lu.dim, lu.num_heads
            c = C + lu.lr_pe_dim
            cp, fin, fin_p = _pad64(c), 10 * lu.n_freqs + 3, _pad64(10 * lu.n_freqs + 3)
            hd = c // heads
            hdp = 64 if hd <= 64 else (128 if hd <= 128 else 256)  # n_dim 384 -> 101 -> 128; n_dim 768 -> 197 -> 256
            if c % heads or hd > 256:
                raise NotImplementedError("head dim > 256")
            f32 = lambda t: t.detach().float().contiguous()

            def padded(w, rows, cols):  # [r, k] -> zero-padded bf16 [rows, cols]
                out = torch.zeros(rows, cols, device=dev, dtype=torch.float32)
                out[:w.shape[0], :w.shape[1]] = w
                return out.to(wd).contiguous()

            def padvec(v, n):
                out = torch.zeros(n, device=dev, dtype=torch.float32)
                out[:v.shape[0]] = v
                return out

            def conv_bn(conv, bn, cin_p, cout_p):  # fold eval BatchNorm, [N,C,3,3] -> [Np, 9*Cp]
                if train:  # raw conv; its BatchNorm runs on batch statistics as a separate op
                    w, b = conv.weight.detach().float(), conv.bias.detach().float()
                else:
                    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
                    w = conv.weight.detach().float() * s[:, None, None, None]
                    b = (conv.bias.detach().float() - bn.running_mean.float()) * s + bn.bias.detach().float()
                wt = torch.zeros(cout_p, 3, 3, cin_p, device=dev)
                wt[:w.shape[0], :, :, :w.shape[1]] = w.permute(0, 2, 3, 1)
                return wt.reshape(cout_p, 9 * cin_p).to(wd).contiguous(), padvec(b, cout_p)

            def head_rows(w, b):  # [heads*hd, k] rows -> [heads*hdp, cp]; bias likewise
                wo = torch.zeros(heads, hdp, cp, device=dev)
                wo[:, :hd, :c] = w.detach().float().reshape(heads, hd, c)
                bo = torch.zeros(heads, hdp, device=dev)
                bo[:, :hd] = b.detach().float().reshape(heads, hd)
                return wo.reshape(heads * hdp, cp).to(wd).contiguous(), bo.reshape(-1).contiguous()

            P = dict(C=C, c=c, cp=cp, fin=fin, fin_p=fin_p, heads=heads, hd=hd, hdp=hdp,
                     cn_w=f32(cn.norm.weight), cn_b=f32(cn.norm.bias), cn_eps=cn.norm.eps,
                     freqs=torch.exp(torch.linspace(-2, 10, lu.n_freqs)).to(dev),
                     bias_sin=f32(lu.fourier_feat[1].biases[0].reshape(-1)),
                     bias_cos=f32(lu.fourier_feat[1].biases[1].reshape(-1)),
                     fc_cn_w=f32(lu.first_conv[0].norm.weight), fc_cn_b=f32(lu.first_conv[0].norm.bias),
                     fc_cn_eps=lu.first_conv[0].norm.eps)
            P["conv1_w"], P["conv1_b"] = conv_bn(lu.first_conv[1], lu.first_conv[2], fin_p, cp)
            P["conv2_w"], P["conv2_b"] = conv_bn(lu.first_conv[4], lu.first_conv[5], cp, cp)
            for name, bn in (("bn1", lu.first_conv[2]), ("bn2", lu.first_conv[5])):  # affine, zero on padded channels
                P[name + "_g"], P[name + "_bt"] = padvec(f32(bn.weight), cp), padvec(f32(bn.bias), cp)
            layers = []
            for ca, ff in lu.ca_transformer.layers:
                E = c
                ipw, ipb = ca.attention.in_proj_weight, ca.attention.in_proj_bias
                L = dict(nq_w=f32(ca.norm_q.weight), nq_b=f32(ca.norm_q.bias), nq_eps=ca.norm_q.eps,
                         nkv_w=f32(ca.norm_kv.weight), nkv_b=f32(ca.norm_kv.bias), nkv_eps=ca.norm_kv.eps)
                L["wq"], L["bq"] = head_rows(ipw[:E], ipb[:E])
                # inference copy of the query projection carrying softmax scale x log2(e) (nn.MultiheadAttention scales q by
                # head_dim ** -0.5 after the projection: same product, rounded once): Q K^T are then base-2 logits, the form the
                # software-pipelined attention kernel takes (csrc/attention_pipe.hip)
                qs = (c // heads) ** -0.5 * 1.4426950408889634
                L["wq2"], L["bq2"] = head_rows(ipw[:E].detach().float() * qs, ipb[:E].detach().float() * qs)
                L["wk"], L["bk"] = head_rows(ipw[E:2 * E], ipb[E:2 * E])
                L["wv"], L["bv"] = head_rows(ipw[2 * E:], ipb[2 * E:])
                wo = torch.zeros(cp, heads, hdp, device=dev)  # out_proj: input index = head*hd + d
                wo[:c, :, :hd] = ca.attention.out_proj.weight.detach().float().reshape(c, heads, hd)
                L["wo"], L["bo"] = wo.reshape(cp, heads * hdp).to(wd).contiguous(), padvec(f32(ca.attention.out_proj.bias), cp)
                L["ff_nw"], L["ff_nb"], L["ff_eps"] = f32(ff.net[0].weight), f32(ff.net[0].bias), ff.net[0].eps
                hid = ff.net[1].weight.shape[0]
                hid_p = _pad64(hid)
                L["ff1_w"], L["ff1_b"] = padded(ff.net[1].weight.detach().float(), hid_p, cp), padvec(f32(ff.net[1].bias), hid_p)
                L["ff2_w"], L["ff2_b"] = padded(ff.net[4].weight.detach().float(), cp, hid_p), padvec(f32(ff.net[4].bias), cp)
                if half:  # LayerNorm folded into the consuming GEMM (csrc/gemm.hip, EpLnFold): W diag(g), its row sums, c + W b
                    def fold(w_pad, b_pad, ln_w, ln_b):
                        wf = w_pad.float().clone()
                        wf[:, :c] *= ln_w.detach().float()[None, :]
                        wh = wf.to(ops.F16).contiguous()
                        return wh, wh.float().sum(1).contiguous(), (b_pad + w_pad.float()[:, :c] @ ln_b.detach().float()).contiguous()
                    L["ff1_fold"] = fold(L["ff1_w"], L["ff1_b"], ff.net[0].weight, ff.net[0].bias)
                    L["wq2_fold"] = fold(L["wq2"], L["bq2"], ca.norm_q.weight, ca.norm_q.bias)
                layers.append(L)
            P["layers"] = layers
            nrm = lu.ca_transformer.norm
            P["tn_w"], P["tn_b"], P["tn_eps"] = f32(nrm.weight), f32(nrm.bias), nrm.eps
            fc = lu.final_conv[0]
            P["fin_w"], P["fin_b"] = padded(fc.weight.detach().float().flatten(1), _pad64(C), cp), padvec(f32(fc.bias), _pad64(C))
            if half:
                wf = P["fin_w"].float().clone()
                wf[:, :c] *= nrm.weight.detach().float()[None, :]
                wh = wf.to(ops.F16).contiguous()
                P["fin_fold"] = (wh, wh.float().sum(1).contiguous(), (P["fin_b"] + P["fin_w"].float()[:, :c] @ nrm.bias.detach().float()).contiguous())
            P["fln_w"], P["fln_b"], P["fln_eps"] = f32(lu.final_conv[1].weight), f32(lu.final_conv[1].bias), lu.final_conv[1].eps
            self._pe_cache.clear()
            return P
        if train:  # raw weights: independent of the running statistics the train-mode forward keeps updating
            return self._packed_train.get(self._packed_train.tensors_of(lambda: list(self.upsampler.parameters())), build)
        cache = self._packed_half if half else self._packed
        params = cache.tensors_of(lambda: list(self.upsampler.parameters()) + [b for n, b in self.upsampler.named_buffers() if "running" in n])
        return cache.get(params, build)

    def _lr_pe(self, h, w, device, dtype=BF16):
        """Sine PE of the LR grid (ImplicitFeaturizer(color_feats=False, n_freqs=5), layers.py:107-158):
        depends only on (h, w) and the learnt biases -> a [h*w, 20] table, cached."""
        key = (h, w, str(device), dtype)
        if key not in self._pe_cache:
            with torch.no_grad():
                lu = self.upsampler.upsampler
                gh, gw = torch.linspace(-1, 1, h, device=device), torch.linspace(-1, 1, w, device=device)
                grid = torch.stack(torch.meshgrid(gh, gw, indexing="ij"))  # [2,h,w]
                freqs = torch.exp(torch.linspace(-2, 10, 5, device=device)).reshape(5, 1, 1, 1)
                feats = grid.unsqueeze(0) * freqs  # [F,2,h,w]
                bias = lu.lr_pe.biases.detach().float().to(device)
                s = torch.sin(feats + bias[0].reshape(5, 2, 1, 1)).reshape(10, h, w)
                c = torch.cos(feats + bias[1].reshape(5, 2, 1, 1)).reshape(10, h, w)
                self._pe_cache[key] = torch.cat([s, c], 0).permute(1, 2, 0).reshape(h * w, 20).to(dtype).contiguous()
        return self._pe_cache[key]

    def forward(self, source: torch.Tensor, guidance: torch.Tensor) -> torch.Tensor:
        src = to_nhwc_bf16(source, keep_f16=LOFTUP_F16 and not self._bn_train() and ops.conv_takes_f16(self._cp()))  # (half tokens only for the half stream)
        if torch.is_grad_enabled() and src.requires_grad:
            # training with clicks injected before the upsampler (the reference's default): activation
            # gradients w.r.t. the LR features flow back through the K/V side of both cross-attention layers
            return nchw_view(_LoftUpFn.apply(src, guidance, self))
        return nchw_view(self._run(src, guidance, None))

    def _run(self, src, guidance, save):
        """The whole upsampler on NHWC bf16 LR features; `save` (a dict) collects what the backward needs."""
        train = self._bn_train()
        # inference runs the whole stream -- tokens, Fourier features, both convolutions, both cross-attention + feed-forward
        # layers, the final projection and LayerNorms -- on IEEE half: its maps are LayerNorm-bounded, and the twelve bf16
        # roundings between the ViT's tokens and the head were 2.5e-3 of the 2.7e-3 rms logit error of S/14 + LoftUp
        half = save is None and not train and LOFTUP_F16 and ops.conv_takes_f16(self._cp())  # (else bf16: no f16 conv for that width)
        dt = ops.F16 if half else BF16
        P = self.packed(train, half)
        B, h, w, C = src.shape
        guidance = guidance.float().contiguous()
        H, W = guidance.shape[2:]
        c, cp, heads, hdp = P["c"], P["cp"], P["heads"], P["hdp"]
        M, T = B * H * W, h * w
        # ---- LR tokens: ChannelNorm(source) ++ sine PE, zero-padded to cp
        kv = torch.zeros(B, T, cp, device=src.device, dtype=dt)
        kv[:, :, :C] = ops.layernorm(src.view(-1, C), P["cn_w"], P["cn_b"], P["cn_eps"], out_dtype=dt).view(B, T, C)
        kv[:, :, C:c] = self._lr_pe(h, w, src.device, dt)
        kv = kv.view(B * T, cp)
        # ---- queries: Fourier features -> ChannelNorm -> 2 x (conv3x3 + folded BN + ReLU).  Guidance-only: the
        # click loop reuses them (and the first layer's query projection) while the image is unchanged.
        def image_queries():
            mm = ops.minmax_nchw(guidance)
            f = ops.loftup_fourier_cn(guidance, mm, P["freqs"], P["bias_sin"], P["bias_cos"], P["fc_cn_w"], P["fc_cn_b"],
                                      P["fin_p"], P["fc_cn_eps"], out_dtype=dt)
            if train:
                from .LiFT import LiFTUpsampler
                fc = self.upsampler.upsampler.first_conv
                f = LiFTUpsampler._bn_forward(ops.conv3x3(f, P["conv1_w"], P["conv1_b"], None), fc[2], P["bn1_g"], P["bn1_bt"])[0]
                return LiFTUpsampler._bn_forward(ops.conv3x3(f, P["conv2_w"], P["conv2_b"], None), fc[5], P["bn2_g"],
                                                 P["bn2_bt"])[0].view(M, cp)
            f = ops.conv3x3(f, P["conv1_w"], P["conv1_b"], "relu")
            if fold and LOFTUP_Q0_STATS:  # the second convolution also emits the row statistics the first query projection's folded LayerNorm needs
                y, st = ops.conv3x3_relu_stats(f, P["conv2_w"], P["conv2_b"])
                return y.view(M, cp), st
            return ops.conv3x3(f, P["conv2_w"], P["conv2_b"], "relu").view(M, cp)
        # Half stream with <= 448 padded channels: the LayerNorms in front of the feed-forward, of the query projections and
        # of the final 1x1 conv are FOLDED into those GEMMs (weights carry the gain, a per-row correction in the epilogue),
        # with the row statistics emitted by the kernel that produced the rows (the residual GEMMs, the second convolution) --
        # no pass over the [B*H*W, 448] pixel map is left between the GEMMs (csrc/gemm.hip: EpAxpyResStats / EpBiasActStats /
        # EpLnFold)
        fold = half and cp <= 448 and LOFTUP_LNFOLD
        x = image_queries() if train else self._gcache.get(guidance, pack_serial(P), "x0", image_queries)
        stats = None  # row statistics of the current x
        if fold and LOFTUP_Q0_STATS:
            x, stats = x
        scale = P["hd"] ** -0.5
        if save is not None:
            save.update(kv=kv, layers=[], geom=(B, h, w, C, H, W), train=train)
        for li, L in enumerate(P["layers"]):
            def project_q(x=x, L=L, stats=stats):
                if fold and stats is not None:
                    wf, sf, bf = L["wq2_fold"]
                    return ops.linear_lnfold(x, stats, wf, sf, bf, c, L["nq_eps"]).view(B, H * W, heads, hdp)
                qn = ops.layernorm(x, L["nq_w"], L["nq_b"], L["nq_eps"], D=c, ld_out=cp, out_dtype=dt)
                if save is None:  # inference: base-2-logit queries (scale folded into the projection)
                    return ops.linear(qn, L["wq2"], L["bq2"]).view(B, H * W, heads, hdp)
                return ops.linear(qn, L["wq"], L["bq"]).view(B, H * W, heads, hdp)
            # the first layer's queries see the image only (x is still x0)
            q = (self._gcache.get(guidance, pack_serial(P), "q0" if save is None else "q0_grad", project_q) if (li == 0 and not train)
                 else project_q())  # (the inference projection carries the softmax scale: its own cache slot)
            kn = ops.layernorm(kv, L["nkv_w"], L["nkv_b"], L["nkv_eps"], D=c, ld_out=cp, out_dtype=dt)
            k = ops.linear(kn, L["wk"], L["bk"]).view(B, T, heads, hdp)
            v = ops.linear(kn, L["wv"], L["bv"]).view(B, T, heads, hdp)
            if save is None:
                a = ops.attention(q, k, v, None, q_logit2=True).view(M, heads * hdp)
            else:
                a, lse = ops.attention_lse(q, k, v, scale)
                a = a.view(M, heads * hdp)
            if fold:
                x_mid, st_mid = ops.linear_axpy_res_stats(a, L["wo"], L["bo"], x, 1.0)   # cross-attention + residual (+ row statistics)
                wf, sf, bf = L["ff1_fold"]
                f = ops.linear_lnfold(x_mid, st_mid, wf, sf, bf, c, L["ff_eps"], "gelu")  # LayerNorm + Linear + GELU
                x, stats = ops.linear_axpy_res_stats(f, L["ff2_w"], L["ff2_b"], x_mid, 1.0)  # feed-forward + residual
                continue
            x_mid = ops.linear_axpy_res(a, L["wo"], L["bo"], x, 1.0)        # cross-attention + residual
            f = ops.layernorm(x_mid, L["ff_nw"], L["ff_nb"], L["ff_eps"], D=c, ld_out=cp, out_dtype=dt)
            if save is None:
                f = ops.linear(f, L["ff1_w"], L["ff1_b"], "gelu")
            else:
                f, pre = ops.linear_gelu_save(f, L["ff1_w"], L["ff1_b"])
                save["layers"].append(dict(x_in=x, q=q, k=k, v=v, a=a, lse=lse, x_mid=x_mid, pre=pre))
            x = ops.linear_axpy_res(f, L["ff2_w"], L["ff2_b"], x_mid, 1.0)  # feed-forward + residual
        if fold and stats is not None:
            wf, sf, bf = P["fin_fold"]
            if C <= 512 and LOFTUP_TAIL_FUSED:  # LayerNorm + 1x1 conv c -> C + channel LayerNorm: one GEMM, the map is written once
                return ops.linear_lnfold_layernorm(x, stats, wf, sf, bf, c, P["tn_eps"], P["fln_w"], P["fln_b"], P["fln_eps"]).view(B, H, W, C)
            y = ops.linear_lnfold(x, stats, wf, sf, bf, c, P["tn_eps"])     # LayerNorm + 1x1 conv c -> C
        else:
            xn = ops.layernorm(x, P["tn_w"], P["tn_b"], P["tn_eps"], D=c, ld_out=cp, out_dtype=dt)
            y = ops.linear(xn, P["fin_w"], P["fin_b"])                       # 1x1 conv c -> C
        out = ops.layernorm(y, P["fln_w"], P["fln_b"], P["fln_eps"], D=C, ld_out=C, out_dtype=dt)  # channel LayerNorm
        if save is not None:
            save.update(x_fin=x, y=y)
        return out.view(B, H, W, C)

    def _bwd_weights(self, train=False):
        """Transposed (data-gradient) copies of the frozen projection weights, cached with the packed set."""
        P = self.packed(train)
        if "bwd" not in P:
            t = lambda w: w.float().t().contiguous().to(BF16)
            P["bwd"] = dict(fin=t(P["fin_w"]),
                            layers=[dict(ff2=t(L["ff2_w"]), ff1=t(L["ff1_w"]), wo=t(L["wo"]), wq=t(L["wq"]), wk=t(L["wk"]),
                                         wv=t(L["wv"])) for L in P["layers"]])
        return P["bwd"]

    def _backward(self, src, saved, g_out):
        """d out / d src applied to g_out [B,H,W,C] bf16 (all weights frozen).  Mirrors loftup.py:100-138 under
        autograd: channel-LN -> 1x1 conv -> LN -> 2 x [FF, cross-attention (dK, dV; dQ only where the queries depend
        on the LR features, i.e. not in the first layer)] -> K/V LayerNorm -> ChannelNorm of the source."""
        P, Wt = self.packed(saved["train"]), self._bwd_weights(saved["train"])
        B, h, w, C, H, W = saved["geom"]
        c, cp, heads, hdp = P["c"], P["cp"], P["heads"], P["hdp"]
        M, T = B * H * W, h * w
        scale = P["hd"] ** -0.5
        _, gy16 = ops.layernorm_bwd(saved["y"], g_out.reshape(M, C).contiguous(), P["fln_w"], P["fln_eps"], D=C)
        g_xt = ops.linear(gy16, Wt["fin"])
        gx, gx16 = ops.layernorm_bwd(saved["x_fin"], g_xt, P["tn_w"], P["tn_eps"], D=c)
        gkv = gkv16 = None
        n = len(P["layers"])
        for i in range(n - 1, -1, -1):
            L, Wl, S = P["layers"][i], Wt["layers"][i], saved["layers"][i]
            g_pre = ops.linear_mul_dgelu(gx16, Wl["ff2"], S["pre"])
            g_f = ops.linear(g_pre, Wl["ff1"])
            gx, gx16 = ops.layernorm_bwd(S["x_mid"], g_f, L["ff_nw"], L["ff_eps"], gx=gx, D=c)
            g_a = ops.linear(gx16, Wl["wo"]).view(B, H * W, heads, hdp)
            dq, dk, dv = ops.attention_bwd(S["q"], S["k"], S["v"], S["a"].view(B, H * W, heads, hdp), g_a, S["lse"],
                                           scale, want_dq=i > 0)
            g_kn = ops.linear(dk.view(B * T, heads * hdp), Wl["wk"])
            g_kn = ops.linear_axpy_res(dv.view(B * T, heads * hdp), Wl["wv"], None, g_kn, 1.0)
            gkv, gkv16 = ops.layernorm_bwd(saved["kv"], g_kn, L["nkv_w"], L["nkv_eps"], gx=gkv, D=c, want_bf16=i == 0)
            if i > 0:  # the first layer's queries come from the image only
                g_qn = ops.linear(dq.view(M, heads * hdp), Wl["wq"])
                gx, gx16 = ops.layernorm_bwd(S["x_in"], g_qn, L["nq_w"], L["nq_eps"], gx=gx, D=c)
        _, g_src = ops.layernorm_bwd(src.reshape(B * T, C), gkv16[:, :C], P["cn_w"], P["cn_eps"], D=C)
        return g_src.view(B, h, w, C)


class _LoftUpFn(torch.autograd.Function):
    """LoftUpUpsampler as one autograd node: gradient w.r.t. the LR features only (frozen weights, guidance
    carries no gradient)."""

    @staticmethod
    def forward(ctx, src, guidance, module):
        saved = {}
        out = module._run(src.detach(), guidance.detach(), saved)
        ctx.module, ctx.saved, ctx.src = module, saved, src.detach()
        return out

    @staticmethod
    def backward(ctx, g_out):
        g = ctx.module._backward(ctx.src, ctx.saved, g_out.contiguous())
        ctx.saved = None
        return g, None, None
