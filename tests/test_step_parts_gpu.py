"""Step-level pieces on the GPU against plain torch: the fused loss heads (train.py:200-231), the Adam step under
graph replay with a changing learning rate (train.py:82-86), the on-device decoder noise (generator.py:248) and the
dropout seeding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


@gpu
def test_loss_heads_match_torch():
    from vcagan_amd import ops
    g = torch.Generator(device='cpu').manual_seed(3)
    u = torch.randn(32, 1, generator=g)
    c = (torch.randn(32, 1, generator=g) * 30)                 # crosses softplus' threshold of 20 on both sides
    a = torch.randn(2, 1, 80, 300, generator=g)
    b = torch.randn(2, 1, 80, 300, generator=g)
    r = torch.randn(4, 24000, generator=g)
    s = torch.randn(32, generator=g)
    xs = [t.clone().requires_grad_(True) for t in (u, c, a, r, s)]
    want = torch.stack([
        F.softplus(-xs[0]).mean() / 3 + F.softplus(xs[1]).mean() / 3 + 50.0 * 0.7 * F.l1_loss(xs[2], b),
        0.7 * F.l1_loss(xs[2], b),
        xs[3].pow(2).sum(1).mean() + 0.5 * xs[4].mean(),
        xs[4].mean()])
    w = torch.tensor([1.0, 0.25, -2.0, 0.5])
    (want * w).sum().backward()
    ys = [t.detach().cuda().requires_grad_(True) for t in (u, c, a, r, s)]
    heads = [(ops.HEAD_SOFTPLUS, 0, None, -1.0, 1.0 / 32, ((0, 1.0 / 3),)),
             (ops.HEAD_SOFTPLUS, 1, None, 1.0, 1.0 / 32, ((0, 1.0 / 3),)),
             (ops.HEAD_L1, 2, 5, 1.0, 0.7 / a.numel(), ((0, 50.0), (1, 1.0))),
             (ops.HEAD_SQSUM, 3, None, 1.0, 1.0 / 4, ((2, 1.0),)),
             (ops.HEAD_SUM, 4, None, 1.0, 1.0 / 32, ((2, 0.5), (3, 1.0)))]
    got = ops.loss_heads(heads, 4, ys + [b.cuda()])
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=1e-6)
    got.backward(w.cuda())
    for x, y in zip(xs, ys):
        np.testing.assert_allclose(y.grad.cpu().numpy(), x.grad.numpy(), rtol=2e-5, atol=1e-9)


@gpu
def test_adam_follows_lr_under_graph_replay():
    """A captured optimizer step reads the learning rate from a device word: changing param_groups[0]['lr'] between
    replays changes the update (MultiStepLR, train.py:85-86), exactly as torch.optim.Adam does eagerly."""
    from vcagan_amd import ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 32).cuda()
    ref = torch.nn.Linear(64, 32)
    ref.load_state_dict({k: v.cpu() for k, v in lin.state_dict().items()})
    flat = ops.flatten_module(lin)
    opt = ops.FusedAdam([flat], lr=1e-3, weight_decay=1e-5, amsgrad=True)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-5, amsgrad=True)
    grads = [torch.randn(flat[1].numel()) for _ in range(4)]
    static_g = torch.zeros_like(flat[1])

    def one():
        flat[1].copy_(static_g)
        opt.step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    static_g.copy_(grads[0])
    with torch.cuda.stream(side):
        one()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):         # recorded, not executed
        one()
    lrs = [1e-3, 1e-3, 1e-4, 1e-5]
    for i in (1, 2, 3):                   # replays, the last two with a decayed learning rate
        opt.param_groups[0]['lr'] = lrs[i]
        opt.sync_lr()
        static_g.copy_(grads[i])
        graph.replay()
    torch.cuda.synchronize()
    for i in range(4):
        ropt.param_groups[0]['lr'] = lrs[i]
        off = 0
        for p in ref.parameters():
            n = p.numel()
            p.grad = grads[i][off:off + n].view(p.shape).clone()
            off += (n + 3) // 4 * 4
        ropt.step()
    for p, q in zip(lin.parameters(), ref.parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), rtol=1e-5, atol=2e-7)


@gpu
def test_device_noise_statistics_and_replay():
    from vcagan_amd import ops
    dev = torch.device('cuda', 0)
    ops.rng_advance(dev)
    x = ops.randn((32, 20, 75, 128), dev)
    y = ops.randn((32, 20, 75, 128), dev)
    assert abs(float(x.mean())) < 2e-3 and abs(float(x.std()) - 1.0) < 2e-3
    assert abs(float((x * y).mean())) < 2e-3                      # two call sites: independent streams
    k = float(((x - x.mean()) ** 4).mean() / x.var() ** 2)
    assert abs(k - 3.0) < 0.02                                    # normal kurtosis
    assert float(x.abs().max()) < 6.5
    ops.rng_advance(dev)
    x2 = ops.randn((32, 20, 75, 128), dev)
    assert abs(float((x * x2).mean())) < 2e-3                     # the next step draws new values


@gpu
def test_graphed_step_noise_contract():
    """A step captured with on-device noise draws new noise on every replay; one captured with a noise tensor
    refuses a call without it (generator.py:248 draws fresh noise every step: never a silent reuse)."""
    from oracle import fixtures
    from vcagan_amd import ops, train_step
    ops.set_precision('bf16x3')
    b = {k: (v.cuda() if k != 'vid_len' else v) for k, v in fixtures.make_batch(2, 20, 112, 5).items()}
    m = fixtures.fill_models(train_step.build_models(), 300)
    for k in m:
        m[k].cuda().train()
    step = train_step.GanStep(m, train_step.build_optimizers(m, lr=0.0), 'grid')     # lr 0: weights stay put
    gs = train_step.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], 'device', warmup=1)
    a = gs(b['mel'], b['spec'], b['vid'])['g3'].clone()
    c = gs(b['mel'], b['spec'], b['vid'])['g3'].clone()
    assert float((a - c).abs().max()) > 1e-2                       # same weights and inputs, different noise
    with pytest.raises(AssertionError):
        gs(b['mel'], b['spec'], b['vid'], None, b['noise'])
    gt = train_step.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'], warmup=1)
    with pytest.raises(AssertionError):
        gt(b['mel'], b['spec'], b['vid'])
    a = gt(b['mel'], b['spec'], b['vid'], None, b['noise'])['g3'].clone()
    c = gt(b['mel'], b['spec'], b['vid'], None, b['noise'])['g3'].clone()
    assert float((a - c).abs().max()) < 3e-3                       # same noise: equal up to atomic-add order


@gpu
def test_dropout_seed_depends_on_rank():
    from vcagan_amd import ops
    dev = torch.device('cuda', 0)
    x = torch.ones(1 << 16, device=dev)
    masks = []
    for rank in (0, 1, 0):
        ops.set_seed(1234, rank)
        ops.rng_advance(dev)
        step = ops._rng_step[dev]
        step.zero_()
        ops._rng_calls[0] = 0
        masks.append(ops.dropout(x, 0.3, True) > 0)
    assert bool((masks[0] == masks[2]).all())
    agree = float((masks[0] == masks[1]).float().mean())
    assert abs(agree - (0.7 * 0.7 + 0.3 * 0.3)) < 0.02            # independent masks agree with probability 0.58
    ops.set_seed(0x5EED)
