"""CPU restatement (pure PyTorch, fp32) of the VCA-GAN model layer and GAN step.

TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path: it
is imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg, never by the product package.  It restates the algorithm of
the reference's src/models (file:line citations on every class) with the same
sub-module attribute names, so state_dict keys and shapes are identical and a
reference checkpoint loads with strict=True.  It is pinned against outputs of
the imported reference by oracle/make_goldens.py + tests/test_oracle_vs_golden.py.

Deliberate, documented differences from the reference:
  * nothing calls .cuda(); the decoder noise is an optional argument (when
    omitted it is drawn from the CPU default generator exactly as
    generator.py:248 does, then moved to the input's device);
  * the ResNet head pools with the top-left 4x4 window like nn.AvgPool2d(4)
    (resnet.py:82) when the map is >= 4 wide and with a global mean when it is
    smaller (the reference raises there; 96x96 / 48x48 clips need it).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)
LOG_EPS = math.log(1e-5)


# --------------------------------------------------------------------------- #
# visual front-end: src/models/resnet.py + src/models/visual_front.py
# --------------------------------------------------------------------------- #
class BasicBlock(nn.Module):
    """resnet.py:25-66. 3x3 conv-BN-act, 3x3 conv-BN, (+1x1 s2 conv-BN shortcut), add, act."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type='relu'):
        super().__init__()
        assert relu_type in ('relu', 'prelu')
        mk_act = (lambda: nn.ReLU()) if relu_type == 'relu' else (lambda: nn.PReLU(planes))
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1 = mk_act()
        self.relu2 = mk_act()
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.bn2(self.conv2(self.relu1(self.bn1(self.conv1(x)))))
        sc = x if self.downsample is None else self.downsample(x)
        return self.relu2(y + sc)


class ResNet(nn.Module):
    """resnet.py:69-123 with block=BasicBlock, layers=[2,2,2,2]."""

    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), relu_type='relu'):
        super().__init__()
        widths, strides = (64, 128, 256, 512), (1, 2, 2, 2)
        cin = 64
        for li, (w, s, n) in enumerate(zip(widths, strides, layers), 1):
            blocks = []
            for bi in range(n):
                st = s if bi == 0 else 1
                ds = None
                if st != 1 or cin != w:       # resnet.py:10-14, :103-106
                    ds = nn.Sequential(nn.Conv2d(cin, w, 1, st, bias=False), nn.BatchNorm2d(w))
                blocks.append(block(cin, w, st, ds, relu_type=relu_type))
                cin = w
            setattr(self, 'layer%d' % li, nn.Sequential(*blocks))
        for m in self.modules():              # resnet.py:85-93
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    @staticmethod
    def head_pool(x):
        """nn.AvgPool2d(4) (resnet.py:82) for maps >= 4 wide; global mean below (extension)."""
        if x.shape[-1] >= 4 and x.shape[-2] >= 4:
            return F.avg_pool2d(x, 4)
        return x.mean((2, 3), keepdim=True)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.head_pool(x).flatten(1)


class Visual_front(nn.Module):
    """visual_front.py:4-37."""

    def __init__(self, in_channels=1):
        super().__init__()
        self.in_channels = in_channels
        self.frontend = nn.Sequential(
            nn.Conv3d(in_channels, 64, (5, 7, 7), (1, 2, 2), (2, 3, 3), bias=False),
            nn.BatchNorm3d(64), nn.PReLU(64),
            nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)))
        self.resnet = ResNet(BasicBlock, [2, 2, 2, 2], relu_type='prelu')
        self.dropout = nn.Dropout(0.3)
        self.sentence_encoder = nn.GRU(512, 512, 2, bidirectional=True, dropout=0.3)
        self.fc = nn.Linear(1024, 512)

    def forward(self, x):                                  # x: B,1,T,H,W
        f = self.frontend(x)
        B, C, T, H, W = f.shape
        f = self.resnet(f.transpose(1, 2).reshape(B * T, C, H, W))   # B*T,512
        f = self.dropout(f).view(B, T, -1)
        seq = f.transpose(0, 1).contiguous()               # T,B,512
        out, _ = self.sentence_encoder(seq)
        sent = self.fc(out).permute(1, 2, 0).contiguous()  # B,512,T
        return seq.transpose(0, 1), sent


# --------------------------------------------------------------------------- #
# generator / discriminators: src/models/generator.py
# --------------------------------------------------------------------------- #
def _lrelu(x):
    return F.leaky_relu(x, 0.2)


class ResBlk1D(nn.Module):
    """generator.py:8-49 (only normalize=False, downsample=False is ever built: :181-183)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.conv1 = nn.Conv1d(dim_in, dim_in, 5, 1, 2)
        self.conv2 = nn.Conv1d(dim_in, dim_out, 5, 1, 2)
        self.learned_sc = dim_in != dim_out
        if self.learned_sc:
            self.conv1x1 = nn.Conv1d(dim_in, dim_out, 1, 1, 0, bias=False)

    def forward(self, x):
        r = self.conv2(_lrelu(self.conv1(_lrelu(x))))
        s = self.conv1x1(x) if self.learned_sc else x
        return (s + r) / SQRT2


class ResBlk(nn.Module):
    """generator.py:51-92 (built only with normalize=False, downsample=True: :283)."""

    def __init__(self, dim_in, dim_out, downsample=True):
        super().__init__()
        self.downsample = downsample
        self.conv1 = nn.Conv2d(dim_in, dim_in, 5, 1, 2)
        self.conv2 = nn.Conv2d(dim_in, dim_out, 5, 1, 2)
        self.learned_sc = dim_in != dim_out
        if self.learned_sc:
            self.conv1x1 = nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=False)

    def forward(self, x):
        s = self.conv1x1(x) if self.learned_sc else x
        r = self.conv1(_lrelu(x))
        if self.downsample:
            s, r = F.avg_pool2d(s, 2), F.avg_pool2d(r, 2)
        r = self.conv2(_lrelu(r))
        return (s + r) / SQRT2


class GenResBlk(nn.Module):
    """generator.py:94-131."""

    def __init__(self, dim_in, dim_out, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.conv1 = nn.Conv2d(dim_in, dim_out, 5, 1, 2)
        self.conv2 = nn.Conv2d(dim_out, dim_out, 5, 1, 2)
        self.norm1 = nn.BatchNorm2d(dim_in)
        self.norm2 = nn.BatchNorm2d(dim_out)
        self.learned_sc = dim_in != dim_out
        if self.learned_sc:
            self.conv1x1 = nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=False)

    def forward(self, x):
        up = (lambda t: F.interpolate(t, scale_factor=2, mode='nearest')) if self.upsample else (lambda t: t)
        r = self.conv1(up(_lrelu(self.norm1(x))))
        r = self.conv2(_lrelu(self.norm2(r)))
        s = up(x)
        if self.learned_sc:
            s = self.conv1x1(s)
        return (r + s) / SQRT2


class AVAttention(nn.Module):
    """generator.py:142-171: single-head visual-context attention."""

    def __init__(self, out_dim):
        super().__init__()
        self.out_dim = out_dim
        self.k = nn.Linear(512, out_dim)
        self.v = nn.Linear(512, out_dim)
        self.q = nn.Linear(2560, out_dim)
        self.mel = nn.Linear(out_dim, 20 * 64)

    def forward(self, ph, g, lens):
        B, C, Fq, T = g.shape
        q = self.q(g.reshape(B, C * Fq, T).transpose(1, 2))          # B,T',D  (feature = c*F+f)
        k = self.k(ph)                                               # B,S,D
        att = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(self.out_dim)
        S = att.shape[2]
        lens_t = torch.as_tensor([int(l) for l in lens], device=att.device)
        pad = torch.arange(S, device=att.device)[None, None, :] >= lens_t[:, None, None]
        att = att.masked_fill(pad, float('-inf')).softmax(2)
        out = self.mel(torch.bmm(att, self.v(ph)))                   # B,T',1280 (feature = f*C'+c')
        return out.view(B, T, Fq, -1).permute(0, 3, 2, 1)


class Postnet(nn.Module):
    """generator.py:173-192."""

    def __init__(self):
        super().__init__()
        self.postnet = nn.Sequential(
            nn.Conv1d(80, 128, 7, 1, 3), nn.BatchNorm1d(128), nn.LeakyReLU(0.2),
            ResBlk1D(128, 256), ResBlk1D(256, 256), ResBlk1D(256, 256),
            nn.Conv1d(256, 321, 1, 1, 0, bias=False))

    def forward(self, x):
        return self.postnet(x.squeeze(1)).unsqueeze(1)


def _to_mel(c):
    return nn.Sequential(nn.BatchNorm2d(c), nn.LeakyReLU(0.2), nn.Conv2d(c, 1, 1, 1, 0), nn.Tanh())


class Decoder(nn.Module):
    """generator.py:194-265."""

    def __init__(self):
        super().__init__()
        self.decode, self.g1, self.g2, self.g3 = (nn.ModuleList() for _ in range(4))
        self.att1 = AVAttention(256)
        self.attconv1 = nn.Conv2d(128 + 64, 128, 5, 1, 2)
        self.att2 = AVAttention(256)
        self.attconv2 = nn.Conv2d(64 + 32, 64, 5, 1, 2)
        self.to_mel1, self.to_mel2, self.to_mel3 = _to_mel(128), _to_mel(64), _to_mel(32)
        for a, b in ((640, 512), (512, 256), (256, 256)):
            self.decode.append(GenResBlk(a, b))
        for a, b in ((256, 128), (128, 128), (128, 128)):
            self.g1.append(GenResBlk(a, b))
        for i, (a, b) in enumerate(((128, 64), (64, 64), (64, 64))):
            self.g2.append(GenResBlk(a, b, upsample=(i == 0)))
        for i, (a, b) in enumerate(((64, 32), (32, 32), (32, 32))):
            self.g3.append(GenResBlk(a, b, upsample=(i == 0)))

    def forward(self, s, x, lens, noise=None):
        # s: B,512,T sentence ; x: B,T,512 phonemes
        B, T = x.shape[0], x.shape[1]
        ctx = s.transpose(1, 2)
        if noise is None:
            noise = torch.randn([B, 128, 20, T]).to(x.device)         # CPU generator, generator.py:248
        h = torch.cat([x.transpose(1, 2).unsqueeze(2).expand(B, 512, 20, T), noise], 1)
        for blk in list(self.decode) + list(self.g1):
            h = blk(h)
        g1 = h
        h = self.attconv1(torch.cat([h, self.att1(ctx, g1, lens)], 1))
        for blk in self.g2:
            h = blk(h)
        g2 = h
        h = self.attconv2(torch.cat([h, self.att2(ctx, g2, lens)], 1))
        for blk in self.g3:
            h = blk(h)
        return self.to_mel1(g1), self.to_mel2(g2), self.to_mel3(h)


def final_length(vid_length):
    """generator.py:368-371."""
    return (vid_length // 2) // 2


class Avgpool(nn.Module):
    """generator.py:137-140 (parameter-free; occupies an index in the Sequential heads)."""

    def forward(self, x):
        return x.mean([2, 3])


class Discriminator(nn.Module):
    """generator.py:267-317."""

    def __init__(self, num_class=1, max_conv_dim=512, phase='1'):
        super().__init__()
        reps = {'1': 2, '2': 3}.get(phase, 4)
        d = 32
        blocks = [nn.Conv2d(1, d, 5, 1, 2)]
        for _ in range(reps):
            d_out = min(2 * d, max_conv_dim)
            blocks.append(ResBlk(d, d_out, downsample=True))
            d = d_out
        self.main = nn.Sequential(*blocks)
        self.uncond = nn.Sequential(nn.LeakyReLU(0.2), nn.Conv2d(d, d, 5, 1, 0), nn.LeakyReLU(0.2),
                                    Avgpool(), nn.Linear(d, num_class))
        self.cond = nn.Sequential(nn.LeakyReLU(0.2), nn.Conv2d(d + 512, d, 5, 1, 2), nn.LeakyReLU(0.2),
                                  nn.Conv2d(d, d, 5, 1, 0), nn.LeakyReLU(0.2),
                                  Avgpool(), nn.Linear(d, num_class))

    def forward(self, x, c, vid_max_length):
        f_len = final_length(vid_max_length)
        cvec = c.mean(2)[:, :, None, None].expand(-1, -1, 5, f_len)
        h = self.main(x)
        u = self.uncond(h)
        o = self.cond(torch.cat([h, cvec], 1))
        return u.view(u.size(0), -1), o.view(o.size(0), -1)


class sync_Discriminator(nn.Module):
    """generator.py:319-361."""

    def __init__(self, temp=1.0):
        super().__init__()
        self.frontend = nn.Sequential(
            nn.Conv2d(1, 128, 3, 2, 1), nn.BatchNorm2d(128), nn.PReLU(128),
            nn.Conv2d(128, 256, 3, 2, 1), nn.BatchNorm2d(256), nn.PReLU(256))
        self.Res_block = nn.Sequential(BasicBlock(256, 256))
        self.Linear = nn.Linear(256 * 20, 512)
        self.temp = temp

    def forward(self, v_feat, aud, gen=False):
        a = self.Res_block(self.frontend(aud))
        b, c, f, t = a.shape
        a = self.Linear(a.reshape(b, c * f, t).transpose(1, 2))        # B,S,512
        if gen:
            return 5.0 - F.cosine_similarity(v_feat, a, 2).abs().mean(1)
        sim = torch.bmm(F.normalize(v_feat, dim=2), F.normalize(a, dim=2).transpose(1, 2)) / self.temp
        d_va = torch.diagonal(F.log_softmax(sim, 2), dim1=-2, dim2=-1).mean(1)
        d_av = torch.diagonal(F.log_softmax(sim, 1), dim1=-2, dim2=-1).mean(1)
        return -0.5 * (d_va + d_av)


def gan_loss(inputs, label=None):
    """generator.py:363-366."""
    return F.softplus(-inputs if label else inputs).mean()


# --------------------------------------------------------------------------- #
# GAN step body: train.py:166-237 (GRID) / train_LRS.py:179-243 (LRS)
# --------------------------------------------------------------------------- #
def denormalize(m):
    """src/data/vid_aud_grid.py:238-240."""
    return (m + 1.0) * (-LOG_EPS / 2.0) + LOG_EPS


def build_models(temp=1.0):
    """train.py:70-76."""
    return dict(v_front=Visual_front(1), gen=Decoder(), post=Postnet(),
                dis1=Discriminator(phase='1'), dis2=Discriminator(phase='2'),
                dis3=Discriminator(phase='3'), s_dis=sync_Discriminator(temp=temp))


G_NAMES = ('v_front', 'gen', 'post')
D_NAMES = ('dis1', 'dis2', 'dis3', 's_dis')


def build_optimizers(models, lr=1e-4, weight_decay=1e-5, amsgrad=True):
    """train.py:78-83 (GRID: amsgrad=True) / train_LRS.py:94-98 (amsgrad=False)."""
    g = torch.optim.Adam([{'params': models[n].parameters()} for n in G_NAMES],
                         lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)
    d = torch.optim.Adam([{'params': models[n].parameters()} for n in D_NAMES],
                         lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)
    return g, d


def gan_step(models, optimizers, batch, variant='grid', noise=None, ctc=None, on_d_backward=None):
    """One full G+D step.  batch = dict(mel (B,1,80,4T), spec (B,1,321,4T), vid (B,1,T,H,W), vid_len).

    on_d_backward: None or a callable(models, phon) run right after dis_loss.backward() and before
    d_optimizer.step() (train.py:210-211): the only moment at which the discriminators hold the gradient their
    optimizer consumes and v_front holds the sync-loss contribution alone; phon.grad is d dis_loss / d phon.

    ctc: None or dict(front, back, weight): the multi-task extension (no counterpart in the reference's training
    scripts; its ASR scorer, ASR_model/GRID/src/models, is used frozen and in eval mode): gen_loss gains
    weight * F.ctc_loss(log_softmax(back(front(g3))), batch['text'], vid_len, batch['text_len'], blank=0,
    reduction='mean', zero_infinity=True).

    Returns a dict of python floats plus the generator outputs (detached).
    """
    m = models
    g_opt, d_opt = optimizers
    mel, spec, vid, vid_len = batch['mel'], batch['spec'], batch['vid'], batch['vid_len']
    for n in G_NAMES:
        m[n].zero_grad()
    mel1 = F.interpolate(mel, scale_factor=0.25, mode='bilinear', align_corners=False)
    mel2 = F.interpolate(mel, scale_factor=0.5, mode='bilinear', align_corners=False)
    phon, sent = m['v_front'](vid)
    g1, g2, g3 = m['gen'](sent, phon, vid_len, noise=noise)
    reals = [t.detach().clone().requires_grad_(True) for t in (mel1, mel2, mel)]
    T = phon.size(1)
    sd = sent.detach()

    # ---- D phase ----
    dis = (m['dis1'], m['dis2'], m['dis3'])
    real_lo = [d(r, sd, T) for d, r in zip(dis, reals)]
    sync_loss = m['s_dis'](phon, reals[2]).mean()
    r1 = []
    for (u, _), r in zip(real_lo, reals):
        gr = torch.autograd.grad(u.sum(), r, create_graph=True)[0]
        r1.append((gr.reshape(gr.size(0), -1).norm(2, dim=1) ** 2).mean())
    fake_lo = [d(g.detach(), sd, T) for d, g in zip(dis, (g1, g2, g3))]
    real_loss = (sum(gan_loss(u, True) for u, _ in real_lo) + sum(gan_loss(c, True) for _, c in real_lo)) / 3 \
        + sum(r1) / 3
    fake_loss = (sum(gan_loss(u, False) for u, _ in fake_lo) + sum(gan_loss(c, False) for _, c in fake_lo)) / 3
    sync_w = 1.0 if variant == 'grid' else 0.5
    dis_loss = real_loss + fake_loss + sync_w * sync_loss
    d_opt.zero_grad()
    if on_d_backward is not None:
        phon.retain_grad()
    dis_loss.backward(retain_graph=True)
    if on_d_backward is not None:
        on_d_backward(m, phon)
        phon.grad = None
    d_opt.step()

    # ---- G phase ----
    gs = m['post'](g3)
    gen_lo = [d(g, sd, T) for d, g in zip(dis, (g1, g2, g3))]
    g_sync = m['s_dis'](phon.detach(), g3, gen=True).mean()
    g_adv = (sum(gan_loss(u, True) for u, _ in gen_lo) + sum(gan_loss(c, True) for _, c in gen_lo)) / 3
    tgt = [r.detach() for r in reals]
    if variant == 'grid':
        g_loss = g_adv + g_sync
        recon = sum(F.l1_loss(denormalize(g), denormalize(t)) for g, t in zip((g1, g2, g3), tgt)) / 3 \
            + F.l1_loss(gs, spec)
        gen_loss = g_loss + 50.0 * recon
    else:
        g_loss = g_adv
        recon = sum(F.l1_loss(g, t) for g, t in zip((g1, g2, g3), tgt)) / 3 + F.l1_loss(gs, spec)
        gen_loss = g_loss + 50.0 * recon + g_sync
    ctc_loss = None
    if ctc is not None:
        lp = F.log_softmax(ctc['back'](ctc['front'](g3)), 2).transpose(0, 1)          # (T, B, 28)
        frames = torch.as_tensor([int(v) for v in vid_len]).clamp(max=lp.size(0))
        ctc_loss = F.ctc_loss(lp, batch['text'], frames, torch.as_tensor(batch['text_len']), blank=0,
                              reduction='mean', zero_infinity=True)
        gen_loss = gen_loss + float(ctc['weight']) * ctc_loss
    for n in D_NAMES + ('gen', 'post'):
        m[n].zero_grad()
    gen_loss.backward()
    g_opt.step()
    f = lambda t: t.detach().item()
    out = dict(dis_loss=f(dis_loss), gen_loss=f(gen_loss), sync_loss=f(sync_loss),
               g_sync_loss=f(g_sync), recon_loss=f(recon), g_loss=f(g_loss),
               grad_p=[f(x) for x in r1],
               g1=g1.detach(), g2=g2.detach(), g3=g3.detach(), gs=gs.detach())
    if ctc_loss is not None:
        out['ctc_loss'] = f(ctc_loss)
    return out
