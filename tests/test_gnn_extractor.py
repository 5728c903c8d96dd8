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
