"""`Learner` / `Worker`: the reference's actor classes (runner.py:14-97) without Ray.

The reference runs them as Ray actors in separate processes and moves weights, buffers and gradient lists through
the object store.  Here one process drives one GPU: a `Worker` owns a batched `Pursuit_Env` (its N environments play
the role of N reference workers), a `Learner` owns the update; method names, arguments and return values are the
reference's, so its main.py loop runs on them with `.remote()` calls replaced by direct calls.
The production loop (trainer.py) skips the weight / gradient copies these methods imply: it shares one agent
between rollout and update and all-reduces one flat gradient bucket over RCCL.
"""
import os

import torch

from .mappo import MAPPO, BigBuffer
from .pursuit_env import Pursuit_Env


class Learner:
    def __init__(self, cfg, batch_size, mini_batch_size, learner_id):
        self.learner_id = learner_id
        self.total_steps = 0
        self.agent = MAPPO(cfg, batch_size, mini_batch_size, "Learner")
        self.buffer = BigBuffer()
        self.cwd = cfg.algo.save_cwd
        os.makedirs(self.cwd, exist_ok=True)
        self.learner_device = torch.device(cfg.algo.learner_device)
        self.use_lr_decay = cfg.algo.use_lr_decay

    def collect_buffer(self, worker_run_ref):
        """runner.py:28-40: worker results (exp_reward, buffer, steps) -> one training batch."""
        self.buffer.reset()
        exp_r = 0.0
        exp_steps = 0
        for reward, buffer_items, steps in worker_run_ref:
            exp_r += reward
            exp_steps += steps
            self.buffer.concat_buffer(buffer_items)
        return exp_r / len(worker_run_ref), exp_steps

    def compute_and_get_gradients(self, total_steps):
        with torch.enable_grad():
            object_c, object_a, actor_grad, critic_grad = self.agent.train(self.buffer, total_steps)
        return (object_c, object_a), actor_grad, critic_grad

    def save(self):
        return [self.agent.actor, self.agent.critic]

    def get_actor(self):
        return self.agent.actor

    def get_weights(self):
        return self.agent.actor.get_weights(), self.agent.critic.get_weights()

    def set_weights(self, actor_weights, critic_weights):
        self.agent.actor.set_weights(actor_weights)
        self.agent.critic.set_weights(critic_weights)

    def set_gradients_and_update(self, actor_grad, critic_grad, total_steps):
        """runner.py:72-78: summed gradients in, identical Adam step on every learner, lr decay."""
        self.agent.ac_optimizer.zero_grad()
        self.agent.actor.set_gradients(actor_grad, self.learner_device)
        self.agent.critic.set_gradients(critic_grad, self.learner_device)
        self.agent.ac_optimizer.step()
        if self.use_lr_decay:
            self.agent.lr_decay(total_steps)


class Worker:
    def __init__(self, worker_id, cfg, num_envs=None):
        self.env = Pursuit_Env(cfg, num_envs=num_envs, rank=worker_id)
        self.agent = MAPPO(cfg, None, None, "Worker")
        self.agent.sample_rank = int(worker_id)  # every reference Worker has its own torch generator (runner.py:81-86)
        self.sample_epi_num = cfg.algo.sample_epi_num

    def run(self, actor_weights, critic_weights):
        with torch.no_grad():
            self.agent.actor.set_weights(actor_weights)
            self.agent.critic.set_weights(critic_weights)
            return self.agent.explore_env(self.env, self.sample_epi_num)
