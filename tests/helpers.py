"""Shared test helpers: golden loading and the oracle replay of an environment trace."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def trace_files(pattern="env_trace_*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_trace(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    seed, P, W, H, blocks, variance, T = [int(v) for v in d["meta"]]
    d.update(seed=seed, P=P, W=W, H=H, blocks=blocks, variance=variance, T=T)
    n_obs = int(d["n_obs"])
    d["o_adj"] = np.unpackbits(d["o_adj"], axis=-1)[..., :n_obs]
    if "raser" in d:
        d["raser"] = np.unpackbits(d["raser"], axis=-1)[..., :n_obs].reshape(W, H, n_obs)
    d["tape"] = d["drawn_targets"][1:]
    return d


def padded_tape(d, tape_len):
    tape = np.zeros((tape_len, 2), np.int32)
    k = min(len(d["tape"]), tape_len)
    tape[:k] = d["tape"][:k]
    return tape


def product_cfg(P, W, H, T=150, depth=1, blocks=5, variance=10, **extra):
    from distributed_multi_agent_reinforcement_learning_amd.config import load_config
    ov = {"env.num_defender": P, "map.map_size": [W, H], "map.center": [W // 2, H // 2], "env.max_steps": T,
          "algo.depth": depth, "map.num_obstacle_block": blocks, "map.variance": variance, "algo.use_reward_norm": True}
    ov.update(extra)
    return load_config(**ov)


def init_from_traces(traces, O=176, tape_len=16):
    """Stack golden traces of equal config into the host-init dict BatchedEnv.load takes."""
    N = len(traces)
    d0 = traces[0]
    W, H, P = d0["W"], d0["H"], d0["P"]
    init = dict(grid=np.zeros((N, W, H), np.uint8), obs_xy=np.zeros((N, O, 2), np.int32), n_obs=np.zeros(N, np.int32),
                defenders=np.zeros((N, P, 4)), evader=np.zeros((N, 4)), target=np.zeros((N, 2), np.int32),
                tape=np.zeros((N, tape_len, 2), np.int32))
    for n, d in enumerate(traces):
        k = int(d["n_obs"])
        init["grid"][n] = d["grid"]; init["obs_xy"][n, :k] = d["obs_xy"]; init["n_obs"][n] = k
        init["defenders"][n] = d["defenders0"]; init["evader"][n] = d["evader0"]; init["target"][n] = d["target0"]
        init["tape"][n] = padded_tape(d, tape_len)
    return init


def oracle_envs_from_init(init, P, W, H, T, O=176, tape_len=16):
    from oracle import pe_oracle
    cfg = pe_oracle.make_config(W=W, H=H, P=P, O=O, max_steps=T, tape_len=tape_len)
    envs = []
    for n in range(len(init["n_obs"])):
        e = pe_oracle.OracleEnv(cfg)
        k = int(init["n_obs"][n])
        e.load(init["grid"][n], init["obs_xy"][n, :k], init["defenders"][n], init["evader"][n], init["target"][n], init["tape"][n])
        envs.append(e)
    return cfg, envs


def random_init(N, P, W, H, blocks, variance, seed, O=176, tape_len=16):
    """Host initial conditions from the oracle's reset restatement (seeded per environment)."""
    import random
    from oracle import reset_oracle
    init = dict(grid=np.zeros((N, W, H), np.uint8), obs_xy=np.zeros((N, O, 2), np.int32), n_obs=np.zeros(N, np.int32),
                defenders=np.zeros((N, P, 4)), evader=np.zeros((N, 4)), target=np.zeros((N, 2), np.int32),
                tape=np.zeros((N, tape_len, 2), np.int32))
    for n in range(N):
        random.seed(seed + n); np.random.seed(seed + n)
        r = reset_oracle.reset_oracle(W, H, P, blocks, [W // 2, H // 2], variance, tape_len=tape_len)
        k = len(r["obs_xy"])
        init["grid"][n] = r["grid"]; init["obs_xy"][n, :k] = r["obs_xy"]; init["n_obs"][n] = k
        init["defenders"][n] = r["defenders"]; init["evader"][n] = r["evader"]; init["target"][n] = r["target"]
        init["tape"][n] = r["tape"]
    return init


# ---------------------------------------------------------------- model goldens
def load_model_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    seed, P, W, H, blocks, variance, depth, T, n_epi, mb, E = [int(v) for v in d["meta"]]
    d.update(seed=seed, P=P, W=W, H=H, blocks=blocks, variance=variance, depth=depth, T=T, n_epi=n_epi, mb=mb, E=E)
    d["buf_o_adj"] = np.unpackbits(d["buf_o_adj"], axis=-1)[..., :176].astype(np.float32)
    return d


def golden_cfg(d, **extra):
    cfg = product_cfg(d["P"], d["W"], d["H"], T=d["T"], depth=d["depth"], blocks=d["blocks"], variance=d["variance"],
                      **{"algo.sample_epi_num": d["n_epi"], "algo.max_train_steps": 100000, "algo.embedding_dim": d["E"],
                         "algo.rnn_hidden_dim": d["E"], **extra})
    return cfg


def golden_models(d, device="cpu", from_seed=False):
    """Product modules carrying the reference's initial weights.  from_seed=True re-creates them from the torch seed
    (bit-identical only on the CPU model that made the fixture: orthogonal_ uses LAPACK); otherwise they are loaded
    from the fixture."""
    import torch
    from distributed_multi_agent_reinforcement_learning_amd.model import build_actor_critic
    cfg = golden_cfg(d)
    torch.manual_seed(d["seed"])
    actor, critic = build_actor_critic(cfg, "cpu")
    if not from_seed:
        load_golden_weights(d, actor, critic)
    return cfg, actor.to(device), critic.to(device)


def load_golden_weights(d, actor, critic):
    import torch
    sd_a = {k[len("w_actor_"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("w_actor_")}
    actor.load_state_dict(sd_a)
    sd_c = {k[len("w_critic_"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("w_critic_")}
    missing = critic.load_state_dict(sd_c, strict=False)
    assert all(k.startswith("shared_net.") for k in missing.missing_keys) and not missing.unexpected_keys


def sharpen(d, actor):
    import torch
    with torch.no_grad():
        params = dict(actor.named_parameters())
        for k, f in zip(d["sharpen_keys"], d["sharpen_factors"]):
            params[str(k)].mul_(float(f))


def digest(t):
    import torch
    a = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, np.float64).ravel()
    head = np.zeros(8); tail = np.zeros(8)
    head[:min(8, a.size)] = a[:8]
    tail[:min(8, a.size)] = a[-8:]
    return np.concatenate([[a.size, a.sum(), np.abs(a).sum(), (a * a).sum()], head, tail])


def buffer_tensors(d):
    import torch
    keys = ("p_state", "e_state", "o_state", "p_adj", "e_adj", "o_adj", "actor_historical_embedding",
            "critic_historical_embedding", "v_n", "a_n", "a_logprob_n", "r", "active")
    return {k: torch.as_tensor(np.ascontiguousarray(d["buf_" + k]), dtype=torch.float32) for k in keys}
