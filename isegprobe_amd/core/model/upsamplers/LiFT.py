"""LiFT x2 learned upsampler (reference core/model/upsamplers/LiFT.py:12-146).

``LiFT`` below is a parameter container with the reference's state-dict layout
(``lift.{image_convs_1, image_convs_2, up1.up, up1.conv_1.double_conv, outc}.*`` under
``LiFTUpsampler``).  Forward (eval mode: BatchNorm folded into the conv weights at pack time; module in training
mode -- the reference's net.train(), trainer.py:214 -- : raw convs + batch-statistics BatchNorm kernels, running
statistics updated, and the backward goes through those statistics):
  image pyramid   3->32 s2, 32->32 s2 (+BN+ReLU), adaptive max pool to (2h,2w), 32->32 s2   csrc/lift.hip
  ConvTranspose2d(C+32 -> (C+32)/2, k2, s2)   ONE bf16 GEMM with the four (dy,dx) taps as output
                                              column blocks, then a pixel shuffle (a strided copy)
  DoubleConv      2 x (3x3 conv + BN + ReLU)  implicit-GEMM conv engine (channels padded to x64)
  outc            1x1 conv                    GEMM
The reference's loader does ``torch.load(lift_path)`` then ``.to("cuda")`` (LiFT.py:125-136); with
no readable path the random init is kept."""
import os

import torch
import torch.nn as nn

from .... import hip_ops as ops
from ...utils.log import logger
from .._tensor import BF16, PackedCache, nchw_view, to_nhwc_bf16, pack_serial
from .._guidance_cache import GuidanceCache
from . import BaseUpsampler


def _pad64(n):
    return (n + 63) // 64 * 64


class _DoubleConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.double_conv = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout),
                                         nn.ReLU(inplace=True), nn.Conv2d(cout, cout, 3, padding=1, bias=False),
                                         nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
        self.conv_1 = _DoubleConv(cin // 2 + 32, cout // 2)


class LiFT(nn.Module):
    def __init__(self, in_channels, patch_size, pre_shape=False, post_shape=False):
        super().__init__()
        if pre_shape or post_shape:
            raise NotImplementedError("token-shaped LiFT I/O is not used by the probe")
        if patch_size not in (8, 14, 16):
            raise ValueError(f"patch size {patch_size} not currently supported")
        self.patch_size = patch_size
        self.up1 = _Up(in_channels + 32, in_channels)
        self.outc = nn.Conv2d(in_channels // 2, in_channels, kernel_size=1)
        self.image_convs_1 = nn.Sequential(nn.Conv2d(3, 32, 3, padding=1, stride=2), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                                           nn.Conv2d(32, 32, 3, padding=1, stride=2), nn.BatchNorm2d(32), nn.ReLU(inplace=True))
        self.image_convs_2 = nn.Sequential(nn.Conv2d(32, 32, 3, padding=1, stride=2), nn.BatchNorm2d(32), nn.ReLU(inplace=True))


def _fold(conv, bn):
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
    w = conv.weight.detach().float() * s[:, None, None, None]
    b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(s)
    return w, (b0 - bn.running_mean.float()) * s + bn.bias.detach().float()


def _nofold(conv, bn):
    """Train mode: the conv as it stands (its BatchNorm runs on batch statistics as a separate op)."""
    w = conv.weight.detach().float()
    return w, conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)


class LiFTUpsampler(BaseUpsampler):
    def __init__(self, lift_path: str = None, n_dim: int = 384, patch: int = 14):
        super().__init__()
        self.lift = LiFT(n_dim, patch)
        if lift_path and os.path.exists(str(lift_path)):
            sd = torch.load(lift_path, map_location="cpu")
            self.lift.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()})
            logger.info("Loaded LiFT module from: " + str(lift_path))
        else:
            logger.info("LiFTUpsampler: no checkpoint at lift_path, keeping random init")
        self._packed = PackedCache()
        self._packed_train = PackedCache()
        self._gcache = GuidanceCache()

    def _bn_train(self):
        return self.lift.image_convs_1[1].training

    def packed(self, train=False):
        _fold = globals()["_nofold" if train else "_fold"]

        def build():
            L = self.lift
            dev = L.outc.weight.device
            C = L.outc.weight.shape[0]
            cu_in, cu_out = C + 32, (C + 32) // 2
            cat_c = cu_out + 32
            half = C // 2
            P = dict(C=C, cu_out=cu_out, cat_p=_pad64(cat_c), half_p=_pad64(half), cu_in_p=_pad64(cu_in))
            for name, conv, bn in (("ic1a", L.image_convs_1[0], L.image_convs_1[1]),
                                   ("ic1b", L.image_convs_1[3], L.image_convs_1[4]),
                                   ("ic2", L.image_convs_2[0], L.image_convs_2[1])):
                w, b = _fold(conv, bn)
                P[name + "_w"], P[name + "_b"] = w.permute(0, 2, 3, 1).contiguous(), b.contiguous()
                P[name + "_g"], P[name + "_bt"] = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
            # ConvTranspose2d weight [cin, cout, 2, 2] -> GEMM Wt [4*cout_p4, cin_p], row = (dy*2+dx)*cout + n
            wt = L.up1.up.weight.detach().float()
            g = torch.zeros(4 * cu_out, P["cu_in_p"], device=dev)
            g[:, :cu_in] = wt.permute(2, 3, 1, 0).reshape(4 * cu_out, cu_in)
            P["up_w"], P["up_b"] = g.to(BF16).contiguous(), L.up1.up.bias.detach().float().repeat(4).contiguous()
            dc = L.up1.conv_1.double_conv

            def conv_pack(conv, bn, cin_p, cout_p):
                w, b = _fold(conv, bn)
                o = torch.zeros(cout_p, 3, 3, cin_p, device=dev)
                o[:w.shape[0], :, :, :w.shape[1]] = w.permute(0, 2, 3, 1)
                bo = torch.zeros(cout_p, device=dev)
                bo[:b.shape[0]] = b
                return o.reshape(cout_p, 9 * cin_p).to(BF16).contiguous(), bo
            P["dc1_w"], P["dc1_b"] = conv_pack(dc[0], dc[1], P["cat_p"], P["half_p"])
            P["dc2_w"], P["dc2_b"] = conv_pack(dc[3], dc[4], P["half_p"], P["half_p"])
            for name, bn in (("dc1", dc[1]), ("dc2", dc[4])):  # BatchNorm affine, zero on the padded channels
                gm, bt = torch.zeros(P["half_p"], device=dev), torch.zeros(P["half_p"], device=dev)
                gm[:half], bt[:half] = bn.weight.detach().float(), bn.bias.detach().float()
                P[name + "_g"], P[name + "_bt"] = gm, bt
            ow = torch.zeros(_pad64(C), P["half_p"], device=dev)
            ow[:C, :half] = L.outc.weight.detach().float().flatten(1)
            ob = torch.zeros(_pad64(C), device=dev)
            ob[:C] = L.outc.bias.detach().float()
            P["out_w"], P["out_b"] = ow.to(BF16).contiguous(), ob
            return P
        if train:  # raw weights: independent of the running statistics the train-mode forward keeps updating
            return self._packed_train.get(self._packed_train.tensors_of(lambda: list(self.lift.parameters())), build)
        params = self._packed.tensors_of(lambda: list(self.lift.parameters()) + [b for n, b in self.lift.named_buffers() if "running" in n])
        return self._packed.get(params, build)

    @staticmethod
    def _bn_forward(raw, bn, gamma, beta):
        """Batch-statistics BatchNorm2d + ReLU on a raw conv output; updates the module's running statistics as torch
        does.  Returns (y, sums) -- sums feed the backward."""
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not built")
        y, sums, new = ops.bn_train(raw, gamma, beta, bn.eps, True, (bn.running_mean, bn.running_var), bn.momentum)
        with torch.no_grad():
            bn.running_mean.copy_(new[0])
            bn.running_var.copy_(new[1])
            bn.num_batches_tracked += 1
        return y, sums

    def forward(self, source, guidance):
        """LiFT(imgs=guidance, x=source) (LiFT.py:106-122, :145-146) -> [B, C, 2h, 2w]."""
        x = to_nhwc_bf16(source)
        if torch.is_grad_enabled() and x.requires_grad:  # training with clicks injected before the upsampler
            return nchw_view(_LiFTFn.apply(x, guidance, self))
        return nchw_view(self._run(x, guidance, None))

    def _run(self, x, guidance, save):
        train = self._bn_train()
        P = self.packed(train)
        L = self.lift
        B, h, w, C = x.shape
        g = guidance.float().contiguous()

        def conv_s2(a, name, bn):
            if not train:
                return ops.conv3x3_s2_c32(a, P[name + "_w"], P[name + "_b"])
            return self._bn_forward(ops.conv3x3_s2_c32(a, P[name + "_w"], P[name + "_b"], relu=False), bn,
                                    P[name + "_g"], P[name + "_bt"])[0]

        def pyramid():  # image-only (LiFT.py:109-111): reused across clicks while the image is unchanged
            a = conv_s2(g, "ic1a", L.image_convs_1[1])
            a = conv_s2(a, "ic1b", L.image_convs_1[4])
            a = ops.adaptive_max_pool_nhwc(a, 2 * h, 2 * w)                   # [B,2h,2w,32]
            return a, conv_s2(a, "ic2", L.image_convs_2[1])                    # [B,h,w,32]
        i1, i2 = pyramid() if train else self._gcache.get(g, pack_serial(P), ("pyr", h, w), pyramid)
        xin = torch.zeros(B, h, w, P["cu_in_p"], device=x.device, dtype=BF16)  # cat([x, imgs_2]) + zero pad
        xin[..., :C] = x
        xin[..., C:C + 32] = i2
        n = P["cu_out"]
        up = ops.linear(xin.view(-1, P["cu_in_p"]), P["up_w"], P["up_b"])     # [B*h*w, 4*n]: taps as column blocks
        cat = torch.zeros(B, 2 * h, 2 * w, P["cat_p"], device=x.device, dtype=BF16)
        cat.view(B, h, 2, w, 2, P["cat_p"])[..., :n] = up.view(B, h, w, 2, 2, n).permute(0, 1, 3, 2, 4, 5)  # pixel shuffle
        cat[..., n:n + 32] = i1
        if train:
            dc = L.up1.conv_1.double_conv
            r1 = ops.conv3x3(cat, P["dc1_w"], None, None)
            y1, s1 = self._bn_forward(r1, dc[1], P["dc1_g"], P["dc1_bt"])
            r2 = ops.conv3x3(y1, P["dc2_w"], None, None)
            y2, s2 = self._bn_forward(r2, dc[4], P["dc2_g"], P["dc2_bt"])
            if save is not None:
                save.update(bn=dict(r1=r1, s1=s1, r2=r2, s2=s2, eps1=dc[1].eps, eps2=dc[4].eps))
        else:
            y1 = ops.conv3x3(cat, P["dc1_w"], P["dc1_b"], "relu")
            y2 = ops.conv3x3(y1, P["dc2_w"], P["dc2_b"], "relu")
        out = ops.linear(y2.view(-1, P["half_p"]), P["out_w"], P["out_b"])
        if save is not None:
            save.update(y1=y1, y2=y2, train=train)
        return out.view(B, 2 * h, 2 * w, -1)[..., :C]

    def _bwd_weights(self, train=False):
        P = self.packed(train)
        if "bwd" not in P:
            def rot(wt, cin_p):  # forward [Np][ky][kx][Cp] -> data-gradient conv weights [Cp][2-ky][2-kx][Np]
                npad = wt.shape[0]
                return wt.view(npad, 3, 3, cin_p).flip(1, 2).permute(3, 1, 2, 0).reshape(cin_p, 9 * npad).contiguous()
            t = lambda w: w.float().t().contiguous().to(BF16)
            P["bwd"] = dict(out=t(P["out_w"]), dc2=rot(P["dc2_w"], P["half_p"]), dc1=rot(P["dc1_w"], P["cat_p"]),
                            up=t(P["up_w"]))
        return P["bwd"]

    def _backward(self, saved, g_out, h, w):
        """d out / d source applied to g_out [B,2h,2w,C] bf16 (frozen weights; eval BatchNorm is folded, so the
        chain is 1x1 conv^T -> [ReLU mask, 3x3 conv^T] x 2 -> pixel un-shuffle -> ConvTranspose^T; LiFT.py:30-44,113-122)."""
        train = saved["train"]
        P, Wt = self.packed(train), self._bwd_weights(train)
        B, C, n = g_out.shape[0], P["C"], P["cu_out"]
        M = B * 4 * h * w
        cpad = P["out_w"].shape[0]
        if cpad != C:
            gp = torch.zeros(M, cpad, device=g_out.device, dtype=BF16)
            gp[:, :C] = g_out.reshape(M, C)
        else:
            gp = g_out.reshape(M, C).contiguous()
        from .._autograd import _imposed_mask  # (test hook: ReLU masks imposed from the fp32 oracle; None outside the tests)
        m2 = _imposed_mask(saved["y2"])
        y2 = saved["y2"] if m2 is None else m2
        if train:  # through the batch statistics: d BN_train / d x, the ReLU masks applied inside
            bn = saved["bn"]
            g2 = ops.bn_train_bwd(ops.linear(gp, Wt["out"]), bn["r2"].view(M, -1), y2.view(M, -1), bn["s2"], P["dc2_g"], bn["eps2"])
            g1 = ops.conv3x3(g2.view(B, 2 * h, 2 * w, -1), Wt["dc2"], None, None)
            m1 = _imposed_mask(saved["y1"])
            y1 = saved["y1"] if m1 is None else m1
            g1 = ops.bn_train_bwd(g1.view(M, -1), bn["r1"].view(M, -1), y1.view(M, -1), bn["s1"], P["dc1_g"], bn["eps1"])
        else:
            g2, _ = ops.relu_mask_colsum(ops.linear(gp, Wt["out"]), y2.view(M, -1), want_colsum=False)
            g1 = ops.conv3x3(g2.view(B, 2 * h, 2 * w, -1), Wt["dc2"], None, None)
            m1 = _imposed_mask(saved["y1"])
            y1 = saved["y1"] if m1 is None else m1
            g1, _ = ops.relu_mask_colsum(g1.view(M, -1), y1.view(M, -1), want_colsum=False)
        g_cat = ops.conv3x3(g1.view(B, 2 * h, 2 * w, -1), Wt["dc1"], None, None)          # [B,2h,2w,cat_p]
        g_up = g_cat.view(B, h, 2, w, 2, -1)[..., :n].permute(0, 1, 3, 2, 4, 5).reshape(B * h * w, 4 * n).contiguous()
        g_xin = ops.linear(g_up, Wt["up"])                                                 # [B*h*w, cu_in_p]
        return g_xin.view(B, h, w, -1)[..., :C].contiguous()


class _LiFTFn(torch.autograd.Function):
    """LiFTUpsampler as one autograd node: gradient w.r.t. the LR features only (frozen weights)."""

    @staticmethod
    def forward(ctx, src, guidance, module):
        saved = {}
        out = module._run(src.detach(), guidance.detach(), saved)
        ctx.module, ctx.saved, ctx.hw = module, saved, src.shape[1:3]
        return out.contiguous()

    @staticmethod
    def backward(ctx, g_out):
        g = ctx.module._backward(ctx.saved, g_out.contiguous(), *ctx.hw)
        ctx.saved = None
        return g, None, None
