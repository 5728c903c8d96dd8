"""The plain-torch model/algorithm oracle (oracle/model_oracle.py) and the product's module construction against
golden vectors captured from the reference MAPPO (tests/golden/gen/make_goldens_model.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import model_oracle as mo
from tests.helpers import buffer_tensors, digest, golden_models, load_model_golden, sharpen

NAMES = ["model_p4_20x20_d1", "model_p8_40x40_d3"]


def close(a, b, tol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) <= tol * (1.0 + np.max(np.abs(b)))


def clone_sd(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("name", NAMES)
def test_module_construction_reproduces_reference_weights(name):
    d = load_model_golden(name)
    cfg, actor, critic = golden_models(d, from_seed=True)
    assert list(actor.state_dict().keys()) == [str(k) for k in d["actor_keys"]]
    assert list(critic.state_dict().keys()) == [str(k) for k in d["critic_keys"]]
    for m, dg, pre in ((actor, d["actor_init_digest"], "w_actor_"), (critic, d["critic_init_digest"], "w_critic_")):
        for (k, v), ref in zip(m.state_dict().items(), dg):
            # same generator stream and initialisers as the reference; orthogonal_ (LAPACK QR) may differ in the last
            # bits on another CPU model, everything else is exact
            assert np.allclose(digest(v.float()), ref, rtol=1e-5, atol=1e-6), k
            assert v.shape == tuple(d[("w_actor_" if k.startswith("shared_net.") else pre) + k].shape)
    assert [n for n, _ in actor.named_parameters()] == [str(k) for k in d["actor_keys"]]
    assert actor.shared_net is critic.shared_net


@pytest.mark.parametrize("name", NAMES)
def test_rollout_model_side_with_shared_history_quirk(name):
    d = load_model_golden(name)
    cfg, actor, critic = golden_models(d)
    sharpen(d, actor)
    buf = buffer_tensors(d)
    with torch.no_grad():
        logp, v, ea, ec = mo.rollout_from_observations(clone_sd(actor), clone_sd(critic), buf, d["depth"], n_obs=d["init_n_obs"])
    T, dep = d["T"], d["depth"]
    assert close(logp, d["buf_a_logprob_n"], 2e-5)
    assert close(v[:, :T], d["buf_v_n"][:, :T], 2e-5)
    assert close(ea, d["buf_actor_historical_embedding"][:, dep:], 2e-5)
    assert close(ec, d["buf_critic_historical_embedding"][:, dep:], 2e-5)


@pytest.mark.parametrize("name", NAMES)
def test_sequence_mode_logprob_entropy_values(name):
    d = load_model_golden(name)
    cfg, actor, critic = golden_models(d)
    sharpen(d, actor)
    buf = buffer_tensors(d)
    with torch.no_grad():
        prob = mo.sequence_forward(clone_sd(actor), buf, "actor_historical_embedding", False, d["depth"])
        logp, ent = mo.categorical_logprob_entropy(prob, buf["a_n"])
        vals = mo.sequence_forward(clone_sd(critic), buf, "critic_historical_embedding", True, d["depth"])
    assert close(logp, d["m1_logp"], 2e-5) and close(ent, d["m1_entropy"], 2e-5) and close(vals, d["m1_values"], 2e-5)
    # the training data path does not reproduce the rollout's own numbers (shared history + padded ones-adjacency)
    assert not close(vals, d["buf_v_n"][:, :d["T"]], 1e-4)


@pytest.mark.parametrize("name", NAMES)
def test_gae_and_advantage_normalisation(name):
    d = load_model_golden(name)
    buf = buffer_tensors(d)
    adv, vt = mo.gae(buf["r"], buf["v_n"], buf["active"], 0.99, 0.95)
    assert close(adv, d["gae_adv"], 1e-5) and close(vt, d["gae_v_target"], 1e-6)


@pytest.mark.parametrize("name", NAMES)
def test_train_losses_gradients_and_adam_step(name):
    d = load_model_golden(name)
    cfg, actor, critic = golden_models(d)
    sharpen(d, actor)
    buf = buffer_tensors(d)
    sd_a, sd_c = clone_sd(actor), clone_sd(critic)
    objC, objA, ga, gc, adv, vt = mo.train(sd_a, sd_c, buf, d["depth"], d["mb"], 0.99, 0.95, cfg.algo.epsilon, cfg.algo.entropy_coef)
    assert abs(objC - float(d["objC"])) <= 1e-4 * (1 + abs(float(d["objC"])))
    assert abs(objA - float(d["objA"])) <= 1e-4 * (1 + abs(float(d["objA"])))
    names_a = [n for n, _ in actor.named_parameters()]
    names_c = [n for n, _ in critic.named_parameters()]
    for names, g, dg in ((names_a, ga, d["actor_grad_digest"]), (names_c, gc, d["critic_grad_digest"])):
        assert len(names) == len(dg)
        for k, ref in zip(names, dg):
            mine = digest(g[k])
            scale = max(1e-6, ref[2] / ref[0])  # mean |g|
            assert mine[0] == ref[0], k
            assert abs(mine[2] - ref[2]) <= 2e-3 * ref[2] + 1e-7, (k, mine[2], ref[2])
            assert np.max(np.abs(mine[4:] - ref[4:])) <= 2e-3 * max(scale, np.max(np.abs(ref[4:]))) + 1e-7, k
    for key in ("shared_net.MSG_layers.2.weight", "shared_net.MSG_layers.0.weight", "Mean.weight", "shared_net.MSG_layers.2.bias"):
        ref = d["agrad_" + key]
        assert np.max(np.abs(ga[key].numpy() - ref)) <= 1e-3 * np.max(np.abs(ref)) + 1e-7, key
    assert np.max(np.abs(gc["Mean.weight_orig"].numpy() - d["cgrad_Mean.weight_orig"])) <= 1e-3 * np.max(np.abs(d["cgrad_Mean.weight_orig"])) + 1e-7
    # G11: the Learner's optimiser step (runner.py:72-78): Adam(lr 5e-4 decayed, eps 1e-5) on ac_parameters
    shared = [sd_a[k] for k in names_a if k.startswith("shared_net.")]
    a_gru = [sd_a[k] for k in names_a if k.startswith("GRU.")]
    c_gru = [sd_c[k] for k in names_c if k.startswith("GRU.")]
    c_mean = [sd_c[k] for k in names_c if k.startswith("Mean.")]
    a_mean = [sd_a[k] for k in names_a if k.startswith("Mean.")]
    opt = torch.optim.Adam(shared + a_gru + c_gru + c_mean + a_mean, lr=cfg.algo.lr, eps=1e-5)
    total_steps = int(d["steps"])
    lr_now = cfg.algo.lr * (1 - total_steps / cfg.algo.max_train_steps)
    assert abs(lr_now - float(d["lr_after_train"])) < 1e-12
    for g in opt.param_groups:
        g["lr"] = lr_now
    opt.step()
    for sd, dg, keys in ((sd_a, d["actor_upd_digest"], d["actor_keys"]), (sd_c, d["critic_upd_digest"], d["critic_keys"])):
        for k, ref in zip(keys, dg):
            k = str(k)
            if k.endswith(("weight_u", "weight_v")):
                continue
            mine = digest(sd[k])
            assert abs(mine[1] - ref[1]) <= 1e-4 * ref[2] + 1e-6, (k, mine[1], ref[1])
            assert np.max(np.abs(mine[4:] - ref[4:])) <= 2e-4, k
