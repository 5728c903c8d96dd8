"""Block parity of the alternative encoder `GnnExtractor` against vectors captured from the reference's
obstacle_differ_3hop.GnnExtractor (tests/golden/gen/make_goldens_gnn.py): construction stream, forward, gradients."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN


def _run(tag, device):
    from distributed_multi_agent_reinforcement_learning_amd.model import GnnExtractor
    z = np.load(os.path.join(GOLDEN, "gnn_extractor.npz"))
    A, K, F, M, O = [int(v) for v in z[f"{tag}_dims"]]
    is_sn = tag == "ortho"
    torch.manual_seed(21 if is_sn else 20)
    net = GnnExtractor(F, M, O, n_hops=1, is_sn=is_sn)
    keys = [k[len(tag) + 3:] for k in z.files if k.startswith(f"{tag}_w_")]
    assert list(net.state_dict().keys()) == keys
    for k, v in net.state_dict().items():   # same initialisers in the same order: same weights from the same seed
        assert np.allclose(v.numpy(), z[f"{tag}_w_{k}"], rtol=1e-5, atol=1e-6), k
    net.load_state_dict({k: torch.as_tensor(z[f"{tag}_w_{k}"]) for k in keys})
    net = net.to(device)
    obs = torch.tensor(z[f"{tag}_obs"], device=device, requires_grad=True)
    last = torch.tensor(z[f"{tag}_last"], device=device, requires_grad=True)
    y = net(obs, last, torch.tensor(z[f"{tag}_adj"], device=device))
    assert np.allclose(y.detach().cpu().numpy(), z[f"{tag}_y"], rtol=1e-5, atol=1e-5)
    (y * torch.tensor(z[f"{tag}_gy"], device=device)).sum().backward()
    assert np.allclose(obs.grad.cpu().numpy(), z[f"{tag}_gobs"], rtol=1e-4, atol=1e-5)
    assert np.allclose(last.grad.cpu().numpy(), z[f"{tag}_glast"], rtol=1e-4, atol=1e-5)
    for k, p in net.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), z[f"{tag}_g_{k}"], rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("tag", ["plain", "ortho"])
def test_gnn_extractor_block_parity_cpu(tag):
    _run(tag, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "ortho"])
def test_gnn_extractor_block_parity_gpu(tag):
    _run(tag, "cuda")


def test_gnn_encoder_is_selectable_and_equals_the_block_on_its_observation_contract():
    """`algo.encoder: gnn_extractor`: actor and critic each get a GnnEncoder (the block's parameters under shared_net.*); its forward on
    the pursuit-evasion observation tensors equals GnnExtractor.forward on the explicitly assembled (obs, last_comm, adj), for the
    actor's adjacency and the critic's ones, per-episode obstacle lists (q_div) and row chunking included."""
    from distributed_multi_agent_reinforcement_learning_amd.model import GnnEncoder, GnnExtractor, build_actor_critic
    from tests.helpers import product_cfg
    cfg = product_cfg(4, 20, 20, T=6, depth=2, blocks=2, variance=4, **{"algo.encoder": "gnn_extractor", "map.num_max_obstacle": 24})
    torch.manual_seed(5)
    actor, critic = build_actor_critic(cfg, "cpu")
    assert isinstance(actor.shared_net, GnnEncoder) and isinstance(critic.shared_net, GnnEncoder) and actor.shared_net is not critic.shared_net
    assert [k for k in actor.state_dict() if k.startswith("shared_net.")] == ["shared_net.one_hop.0.weight", "shared_net.one_hop.0.bias",
                                                                            "shared_net.one_hop.2.weight", "shared_net.one_hop.2.bias",
                                                                            "shared_net.bottleneck.0.weight", "shared_net.bottleneck.0.bias"]
    with pytest.raises(ValueError, match="algo.depth"):
        build_actor_critic(product_cfg(4, 20, 20, depth=1, **{"algo.encoder": "gnn_extractor"}), "cpu")
    with pytest.raises(ValueError, match="algo.encoder"):
        build_actor_critic(product_cfg(4, 20, 20, depth=1, **{"algo.encoder": "transformer"}), "cpu")
    n, T, P, O, E = 3, 6, 4, 24, 128
    R = n * T
    g = torch.Generator().manual_seed(0)
    p, e = torch.rand(R, P, 4, generator=g) * 20, torch.rand(R, 1, 4, generator=g) * 20
    o = torch.rand(n, O, 4, generator=g) * 20
    adj_p, adj_e = (torch.rand(R, P, P, generator=g) < 0.6).float(), (torch.rand(R, P, 1, generator=g) < 0.4).float()
    adj_o = (torch.rand(R, P, O, generator=g) < 0.2).float()
    hist = [torch.randn(R, P, E, generator=g), torch.randn(R, P, E, generator=g)]
    enc = actor.shared_net
    for is_critic in (False, True):
        y = enc(p, e, o, adj_p, adj_e, adj_o, hist, is_critic, None, T)
        oq = o.repeat_interleave(T, dim=0)
        q = torch.cat([p, e, oq], 1)
        kind = torch.zeros(P + 1 + O, 3); kind[:P, 0] = 1; kind[P, 1] = 1; kind[P + 1:, 2] = 1
        obs = torch.cat([q[:, None] - p[:, :, None], kind.expand(R, P, P + 1 + O, 3)], -1)
        adj = torch.cat([adj_p, adj_e, adj_o], -1)
        ref = GnnExtractor.forward(enc, obs, torch.cat([hist[1], hist[0]], -1), torch.ones_like(adj) if is_critic else adj)
        assert y.shape == (R, P, E) and torch.allclose(y, ref, rtol=1e-5, atol=1e-6)
        enc.CHUNK_PAIRS = 2 * T * P * (P + 1 + O)          # two episodes per chunk
        assert torch.allclose(enc(p, e, o, adj_p, adj_e, adj_o, hist, is_critic, None, T), ref, rtol=1e-5, atol=1e-6)
        del enc.CHUNK_PAIRS
