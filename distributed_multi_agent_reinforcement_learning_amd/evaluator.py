"""Greedy evaluation, the evaluator process' bookkeeping and the learning curve -- the reference's evaluator.py API.

Mirrors (reference paths) evaluator.py:20-102 `EvaluatorProc`, :106-201 `evaluate`, :205-266 `draw_learning_curve`.
Ray is gone: evaluation episodes are the N environments of one batched `Pursuit_Env` on this rank's GPU instead of
`num_cpus_eval` remote tasks.
"""
import os
import time

import numpy as np
import torch

from . import ops
from .mappo import MAPPO
from .pursuit_env import Pursuit_Env


@torch.no_grad()
def evaluate(env, actor, cfg, init=None, return_actions=False):
    """evaluator.py:106-201: one greedy episode per environment with the actor's PRIVATE history (hop k reads the
    actor's own embedding of step t-1-k).  Returns [episode_reward (N,), last step index T-1] like the reference
    (which returns the last loop index, not a count)."""
    device = next(actor.parameters()).device
    env.reset(init)
    N, P, T = env.num_envs, env.num_defender, env.max_steps
    d, E, L, H = cfg.algo.depth, cfg.algo.embedding_dim, cfg.algo.num_layers, cfg.algo.rnn_hidden_dim
    hidden = torch.zeros(L, N * P, H, device=device)
    hist = [torch.zeros(N, P, E, device=device) for _ in range(d)]
    cur = torch.zeros(N, P, E, device=device)
    o_state = env.boundary_map.obstacle_agent
    episode_reward = torch.zeros(N, device=device)
    reward = torch.zeros(N, P, device=device)
    raw = torch.zeros(N, P, device=device)
    obs = env.sim.new_obs(packed=True)  # LiDAR rows bit-packed, as in the rollout
    actions = []
    env.observe(obs)
    env.attacker_step()
    step = 0
    for step in range(T):
        o = dict(obs)
        o["o_state"] = o_state
        if d:
            hist = (hist + [cur])[-d:]
        hops = [hist[d - 1 - k] for k in range(d)]
        prob, hidden, cur = actor(o, hops, hidden, 0)
        a_n, _ = ops.categorical_sample(prob, 0, 0, greedy=True)  # prob.argmax(-1) (DHGN/mappo_parallel.py:442-444)
        if return_actions:
            actions.append(a_n.clone())
        if step + 1 < T:
            env.tick(a_n, obs, reward, raw)
        else:
            env.sim.step(a_n, reward, raw)
            env.time_step += 1
        episode_reward += raw.sum(-1)
    out = [episode_reward, step]
    if return_actions:
        out.append(torch.stack(actions, 1))  # (N, T, P)
    return out


class EvaluatorProc:
    """evaluator.py:20-102 without Ray: keeps the recorder, prints the table, decides when to save."""

    def __init__(self, cfg, num_cpus_eval, rank=0):
        eval_cfg = cfg
        self.env = Pursuit_Env(eval_cfg, num_envs=num_cpus_eval, rank=10_000 + rank)
        self.agent = MAPPO(cfg, None, None, "Evaluator")
        self.total_step = 0
        self.start_time = time.time()
        self.break_step = cfg.algo.max_train_steps
        self.cfg = cfg
        self.num_cpus_eval = num_cpus_eval
        self.recorder = []
        self.max_r = -np.inf
        print(f"| Evaluator: {num_cpus_eval} greedy episodes per evaluation\n"
              f"{'Step':>8}{'Time':>8} |{'avgR':>8}{'stdR':>7}{'avgS':>7}{'stdS':>6} |{'expR':>8}{'objC':>7}{'objA':>7}")

    def run(self, actor_weights, critic_weights, total_step, exp_r, logging_tuple):
        self.agent.actor.set_weights(actor_weights)
        self.agent.critic.set_weights(critic_weights)
        ref_list = self.evaluate_and_save(total_step, exp_r, logging_tuple)
        if_train = self.total_step <= self.break_step
        return [if_train, ref_list]

    def evaluate_and_save(self, new_total_step, exp_r, logging_tuple):
        self.total_step = new_total_step
        rs = self.get_rewards_and_step()
        returns, steps = rs[:, 0], rs[:, 1]
        avg_r, std_r = returns.mean().item(), (returns.std().item() if len(returns) > 1 else 0.0)
        avg_s, std_s = steps.mean().item(), (steps.std().item() if len(steps) > 1 else 0.0)
        train_time = int(time.time() - self.start_time)
        self.recorder.append((self.total_step, avg_r, std_r, exp_r, *logging_tuple))
        prev_r = self.max_r
        self.max_r = max(self.max_r, avg_r)
        print(f"{self.total_step:8.2e}{train_time:8.0f} |{avg_r:8.2f}{std_r:7.1f}{avg_s:7.0f}{std_s:6.0f} |"
              f"{exp_r:8.2f}{''.join(f'{n:7.2f}' for n in logging_tuple)}")
        if avg_r >= prev_r:
            return [self.agent.actor, self.agent.critic, self.recorder]
        return []

    def get_recorder(self):
        return self.recorder

    def get_rewards_and_step(self):
        """(num_cpus_eval, 2) float32: episodic return and the reference's 'step' (= T-1) per evaluation episode."""
        R, last = evaluate(self.env, self.agent.actor, self.cfg)
        return torch.stack((R.float().cpu(), torch.full((len(R),), float(last))), 1)


def draw_learning_curve(recorder=None, fig_title="learning_curve", save_path="learning_curve.jpg", cwd=None):
    """evaluator.py:205-266: recorder rows (total_step, avg_r, std_r, exp_r, objC, objA) -> jpg."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    recorder = np.asarray(recorder, dtype=np.float64)
    if cwd is not None:
        save_path = os.path.join(cwd, os.path.basename(save_path))
    steps, r_avg, r_std, r_exp, obj_c, obj_a = (recorder[:, i] for i in range(6))
    fig, axs = plt.subplots(2, 1, figsize=(8, 8))
    ax = axs[0]
    ax.plot(steps, r_avg, color="tab:red", label="episode return (greedy)")
    ax.fill_between(steps, r_avg - r_std, r_avg + r_std, color="tab:red", alpha=0.25)
    ax.plot(steps, r_exp, color="tab:green", alpha=0.7, label="exploration return")
    ax.set_ylabel("return"); ax.legend(); ax.grid(True)
    ax = axs[1]
    ax.plot(steps, obj_a, color="tab:blue", label="objA")
    ax.set_ylabel("objA", color="tab:blue"); ax.set_xlabel("env-steps"); ax.grid(True)
    ax2 = ax.twinx()
    ax2.plot(steps, obj_c, color="tab:orange", label="objC")
    ax2.set_ylabel("objC", color="tab:orange")
    fig.suptitle(fig_title)
    plt.savefig(save_path)
    plt.close("all")
