"""Golden vectors for the alternative encoder of SURVEY 8f row 4: obstacle_differ_3hop.GnnExtractor
(reference obstacle_differ_3hop/mappo_parallel.py:34-70), block level -- its environment is not in the reference, so the
module is pinned on seeded random inputs: state_dict, forward output, input / parameter gradients.
Usage:  python tests/golden/gen/make_goldens_gnn.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refload  # noqa: E402

refload.activate()
OUT = os.path.dirname(HERE)


def main():
    from obstacle_differ_3hop.mappo_parallel import GnnExtractor
    out = {}
    for tag, is_sn, lead in (("plain", False, ()), ("ortho", True, (3, 4))):   # the reference supports adj of 2 or 4 dims only (:62-65)
        torch.manual_seed(21 if is_sn else 20)
        A, K, F, M, O = 4, 9, 6, 32, 16
        net = GnnExtractor(F, M, O, n_hops=1, is_sn=is_sn)
        obs = torch.randn(*lead, A, K, F, requires_grad=True)
        adj = (torch.rand(*lead, A, K) < 0.5).float()
        adj[..., 0, :] = 0          # an all-zero row
        last = torch.randn(*lead, A, 2 * O, requires_grad=True)
        y = net(obs, last, adj)
        g = torch.randn_like(y)
        (y * g).sum().backward()
        for k, v in net.state_dict().items():
            out[f"{tag}_w_{k}"] = v.numpy().copy()
        for k, p in net.named_parameters():
            out[f"{tag}_g_{k}"] = p.grad.numpy().copy()
        out.update({f"{tag}_obs": obs.detach().numpy(), f"{tag}_adj": adj.numpy(), f"{tag}_last": last.detach().numpy(), f"{tag}_y": y.detach().numpy(),
                    f"{tag}_gy": g.numpy(), f"{tag}_gobs": obs.grad.numpy(), f"{tag}_glast": last.grad.numpy(),
                    f"{tag}_dims": np.asarray([A, K, F, M, O], np.int64)})
        print(tag, "keys", list(net.state_dict().keys()), "y", tuple(y.shape))
    np.savez_compressed(os.path.join(OUT, "gnn_extractor.npz"), **out)


if __name__ == "__main__":
    main()
