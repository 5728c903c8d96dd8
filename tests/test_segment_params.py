"""mappo._SegmentParams (CPU): per-mini-batch aliases of shared parameters with their own gradient storage -- the mechanism behind
MAPPO._train_grouped (one autograd graph for all mini-batches of an epoch, separate gradients for the per-mini-batch clip)."""
import torch
import torch.nn as nn

from distributed_multi_agent_reinforcement_learning_amd.mappo import _SegmentParams


class _Net(nn.Module):
    def __init__(self, enc):
        super().__init__()
        self.shared_net = enc                       # the same encoder instance in both networks, like the DHGN of actor and critic
        self.head = nn.Linear(6, 3)

    def forward(self, x):
        return self.head(torch.relu(self.shared_net(x)))


def test_segment_params_deliver_per_segment_gradients_and_restore_the_modules():
    torch.manual_seed(0)
    enc = nn.Linear(5, 6)
    a, c = _Net(enc), _Net(enc)
    params = list(enc.parameters()) + list(a.head.parameters()) + list(c.head.parameters())
    segs = _SegmentParams((a, c), params, 3)
    assert len(segs.owners) == len(params)           # the shared encoder's parameters are swapped once
    xs = [torch.randn(7, 5) for _ in range(3)]
    losses = []
    for k, x in enumerate(xs):
        with segs.use(k):
            assert a.shared_net.weight is segs.sets[k][0] and c.head.bias is segs.sets[k][5]
            losses.append(a(x).square().sum() + c(x).abs().sum())
        assert a.shared_net.weight is params[0] and isinstance(a.shared_net.weight, nn.Parameter)
    segs.zero()
    torch.autograd.backward(losses)
    assert all(p.grad is None for p in params)       # nothing leaked into the real parameters
    for k, x in enumerate(xs):                       # each segment's gradient equals the plain per-mini-batch backward
        for p in params:
            p.grad = None
        (a(x).square().sum() + c(x).abs().sum()).backward()
        for g_seg, p in zip(segs.grads(k), params):
            assert torch.allclose(g_seg, p.grad, rtol=1e-6, atol=1e-7)
    # the aliases ARE the weights: an in-place optimiser step is seen by the next forward, a replaced storage is detected
    with torch.no_grad():
        params[0].add_(1.0)
    assert torch.equal(segs.sets[1][0], params[0]) and not segs.stale()
    enc.weight.data = enc.weight.data.clone()
    assert segs.stale()
    # gradients accumulate across backward passes until zero()
    flat0 = segs.flat.clone()
    with segs.use(0):
        a(xs[0]).sum().backward()
    assert not torch.equal(segs.flat[0], flat0[0]) and torch.equal(segs.flat[1], flat0[1])
    segs.zero()
    assert not segs.flat.any()
