"""Golden-vector generator, model / algorithm part (G5-G11 of SURVEY 8c).

Runs the REAL reference MAPPO (DHGN encoder, GRU actor/critic, rollout, GAE, PPO loss, optimiser step, greedy
evaluate) in the build container and stores small fixtures under tests/golden/.  Weights are NOT stored: they are
reproduced from the torch seed by constructing the modules in the reference's order; per-tensor digests pin them.
Usage:  python tests/golden/gen/make_goldens_model.py
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
SHARPEN_OPS = [("Mean.weight", 20.0), ("shared_net.semantic_layer.weight", 4.0), ("GRU.weight_ih_l0", 3.0)]


def digest(t):
    a = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, np.float64).ravel()
    head = np.zeros(8); tail = np.zeros(8)
    head[:min(8, a.size)] = a[:8]
    tail[:min(8, a.size)] = a[-8:]
    return np.concatenate([[a.size, a.sum(), np.abs(a).sum(), (a * a).sum()], head, tail])


def sd_digest(sd):
    names = list(sd.keys())
    return names, np.stack([digest(sd[k].float()) for k in names])


def grads_digest(grads):
    return np.stack([digest(g) if g is not None else np.full(20, np.nan) for g in grads])


def capture(name, seed, P, W, H, blocks, variance, depth, T, n_epi, mb, E=128):
    from environment.pursuit_evasion_game.pursuit_env import Pursuit_Env
    from DHGN.mappo_parallel import MAPPO, AttributeDataset, EmbeddingDataset2
    from DHGN.replay_buffer import BigBuffer
    from torch.utils.data import DataLoader
    import evaluator as ref_eval

    cfg = refload.load_cfg(num_defender=P, map_size=(W, H), blocks=blocks, variance=variance, depth=depth, max_steps=T)
    cfg.algo.sample_epi_num = n_epi
    cfg.algo.max_train_steps = 100000
    cfg.algo.embedding_dim = E
    cfg.algo.rnn_hidden_dim = E
    refload.seed_all(seed)
    env = Pursuit_Env(cfg)
    agent = MAPPO(cfg, n_epi, mb, "Learner")
    out = {}
    names_a, dig_a = sd_digest(agent.actor.state_dict())
    names_c, dig_c = sd_digest(agent.critic.state_dict())
    out["actor_keys"] = np.asarray(names_a); out["critic_keys"] = np.asarray(names_c)
    out["actor_init_digest"] = dig_a; out["critic_init_digest"] = dig_c
    # the initial weights themselves: orthogonal_ goes through LAPACK, whose last bits differ between CPU models,
    # so re-creating them from the seed is only reproducible on the machine that made the fixture
    for k, v in agent.actor.state_dict().items():
        out["w_actor_" + k] = v.numpy().copy()
    for k, v in agent.critic.state_dict().items():
        if not k.startswith("shared_net."):
            out["w_critic_" + k] = v.numpy().copy()
    with torch.no_grad():  # make the policy input-sensitive so greedy actions are not degenerate at init
        for key, fac in SHARPEN_OPS:
            dict(agent.actor.named_parameters())[key].mul_(fac)

    # ---- instrument the env so the initial condition of every reset and every target draw is recorded
    inits, drawn, actions_log = [], [], []
    orig_reset, orig_init_target, orig_step = env.reset, env.init_target, env.step

    def logged_init_target(inflated_map):
        orig_init_target(inflated_map=inflated_map)
        drawn[-1].append(tuple(env.target[0]))

    def logged_reset():
        drawn.append([])
        orig_reset()
        inits.append(dict(grid=np.asarray(env.occupied_map.grid_map, np.uint8),
                          obs_xy=np.asarray(env.boundary_map.obstacles, np.int32).reshape(-1, 2),
                          target=np.asarray(env.target[0], np.int32),
                          defenders=np.asarray(env.get_state('defender'), np.float64),
                          evader=np.asarray(env.get_state('attacker')[0], np.float64)))

    def logged_step(a):
        actions_log.append(np.asarray(a, np.int64).copy())
        return orig_step(a)
    env.reset, env.init_target, env.step = logged_reset, logged_init_target, logged_step

    # ---- G6: rollout (stochastic actions from torch's CPU generator; shared-history quirk included)
    torch.set_grad_enabled(False)
    exp_reward, minibuffer, steps = agent.explore_env(env, n_epi)
    torch.set_grad_enabled(True)
    buf = minibuffer.buffer
    for k, v in buf.items():
        a = v.numpy()
        out["buf_" + k] = np.packbits(a.astype(np.uint8), axis=-1) if k in ("o_adj",) else a
    out["exp_reward"] = np.float64(exp_reward); out["steps"] = np.int64(steps)
    n_roll = len(inits)
    O = cfg.map.num_max_obstacle
    out["init_grid"] = np.stack([i["grid"] for i in inits])
    obs_pad = np.zeros((n_roll, O, 2), np.int32); n_obs = np.zeros(n_roll, np.int32)
    for n, i in enumerate(inits):
        n_obs[n] = len(i["obs_xy"]); obs_pad[n, :n_obs[n]] = i["obs_xy"]
    out["init_obs_xy"] = obs_pad; out["init_n_obs"] = n_obs
    out["init_target"] = np.stack([i["target"] for i in inits])
    out["init_defenders"] = np.stack([i["defenders"] for i in inits])
    out["init_evader"] = np.stack([i["evader"] for i in inits])
    tape = np.zeros((n_roll, 16, 2), np.int32); tape_n = np.zeros(n_roll, np.int32)
    for n, dr in enumerate(drawn[:n_roll]):
        extra = dr[1:]
        tape_n[n] = len(extra)
        for k, tgt in enumerate(extra[:16]):
            tape[n, k] = tgt
    out["init_tape"] = tape; out["init_tape_n"] = tape_n
    rn = agent.reward_norm.running_ms
    out["rn_n"] = np.int64(rn.n); out["rn_mean"] = np.asarray(rn.mean, np.float64); out["rn_S"] = np.asarray(rn.S, np.float64)

    # ---- G5 (mode 1): actor log-prob / entropy and critic values over the whole buffer, clean per-net histories
    big = BigBuffer(); big.concat_buffer(minibuffer)
    batch = big.get_training_data(torch.device("cpu"))
    idx = list(range(n_epi))
    with torch.no_grad():
        h0 = torch.zeros(cfg.algo.num_layers, n_epi * P, cfg.algo.rnn_hidden_dim)
        attrs = [batch['p_state'][idx], batch['e_state'][idx], batch['o_state'][idx]]
        adjs = [batch['p_adj'][idx], batch['e_adj'][idx], batch['o_adj'][idx]]
        a_dl = DataLoader(AttributeDataset(attribute=attrs, adjacent=adjs, is_critic=False), batch_size=1, shuffle=False)
        c_dl = DataLoader(AttributeDataset(attribute=attrs, adjacent=adjs, is_critic=True), batch_size=1, shuffle=False)
        ae_dl = DataLoader(EmbeddingDataset2(attribute=batch['actor_historical_embedding'][idx], adjacent=batch['p_adj'][idx], is_critic=False, depth=depth), batch_size=1, shuffle=False)
        ce_dl = DataLoader(EmbeddingDataset2(attribute=batch['critic_historical_embedding'][idx], adjacent=batch['p_adj'][idx], is_critic=True, depth=depth), batch_size=1, shuffle=False)
        logp, ent = agent.actor.get_logprob_and_entropy(a_dl, ae_dl, h0, batch['a_n'][idx])
        vals = agent.critic(c_dl, ce_dl, h0.clone(), mode=1).squeeze(-1)
    out["m1_logp"] = logp.numpy(); out["m1_entropy"] = ent.numpy(); out["m1_values"] = vals.numpy()

    # ---- G7 + G8: train() -> GAE/adv-norm (captured from its frame), losses, gradient lists
    cap = {}
    zg = agent.ac_optimizer.zero_grad

    def spy_zero_grad(*a, **k):
        f = sys._getframe(1)
        cap["adv"] = f.f_locals["adv"].clone(); cap["v_target"] = f.f_locals["v_target"].clone()
        return zg(*a, **k)
    agent.ac_optimizer.zero_grad = spy_zero_grad
    total_steps = int(steps)
    objC, objA, ag, cg = agent.train(big, total_steps)
    agent.ac_optimizer.zero_grad = zg
    out["gae_adv"] = cap["adv"].numpy(); out["gae_v_target"] = cap["v_target"].numpy()
    out["objC"] = np.float64(objC); out["objA"] = np.float64(objA)
    out["actor_grad_digest"] = grads_digest(ag); out["critic_grad_digest"] = grads_digest(cg)
    out["lr_after_train"] = np.float64(agent.ac_optimizer.param_groups[0]["lr"])
    # a few full gradient tensors (small ones) for element-wise comparison
    pa = dict(agent.actor.named_parameters())
    for key in ("shared_net.MSG_layers.2.weight", "shared_net.MSG_layers.0.weight", "Mean.weight", "shared_net.MSG_layers.2.bias"):
        out["agrad_" + key] = pa[key].grad.numpy().copy()
    pc = dict(agent.critic.named_parameters())
    out["cgrad_Mean.weight_orig"] = pc["Mean.weight_orig"].grad.numpy().copy()
    out["cgrad_GRU.bias_hh_l1"] = pc["GRU.bias_hh_l1"].grad.numpy().copy()

    # ---- G11: Learner.set_gradients_and_update (runner.py:72-78)
    agent.ac_optimizer.zero_grad()
    agent.actor.set_gradients(ag, torch.device("cpu"))
    agent.critic.set_gradients(cg, torch.device("cpu"))
    agent.ac_optimizer.step()
    agent.lr_decay(total_steps)
    _, out["actor_upd_digest"] = sd_digest(agent.actor.state_dict())
    _, out["critic_upd_digest"] = sd_digest(agent.critic.state_dict())

    # ---- G10: greedy evaluate (private history), updated weights
    torch.set_grad_enabled(False)
    del actions_log[:]
    n_before = len(inits)
    R, last = ref_eval.evaluate(env, agent.actor, cfg)
    torch.set_grad_enabled(True)
    out["eval_return"] = np.float64(R); out["eval_last_index"] = np.int64(last)
    out["eval_actions"] = np.stack(actions_log)
    ei = inits[n_before]
    k = len(ei["obs_xy"])
    eo = np.zeros((O, 2), np.int32); eo[:k] = ei["obs_xy"]
    out["eval_grid"] = ei["grid"]; out["eval_obs_xy"] = eo; out["eval_n_obs"] = np.int32(k)
    out["eval_target"] = ei["target"]; out["eval_defenders"] = ei["defenders"]; out["eval_evader"] = ei["evader"]
    et = np.zeros((16, 2), np.int32)
    for j, tgt in enumerate(drawn[n_before][1:][:16]):
        et[j] = tgt
    out["eval_tape"] = et
    out["meta"] = np.asarray([seed, P, W, H, blocks, variance, depth, T, n_epi, mb, E], np.int64)
    out["sharpen_keys"] = np.asarray([k for k, _ in SHARPEN_OPS]); out["sharpen_factors"] = np.asarray([f for _, f in SHARPEN_OPS])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "objC", objC, "objA", objA, "evalR", R, "logp range", float(logp.min()), float(logp.max()),
          "greedy action histogram", np.bincount(out["eval_actions"].ravel(), minlength=9), flush=True)


def main():
    capture("model_p4_20x20_d1", seed=11, P=4, W=20, H=20, blocks=2, variance=4, depth=1, T=16, n_epi=4, mb=2, E=64)
    capture("model_p8_40x40_d3", seed=13, P=8, W=40, H=40, blocks=5, variance=10, depth=3, T=12, n_epi=3, mb=2)


if __name__ == "__main__":
    main()
