"""Loads the reference (read-only, /root/reference) in THIS container to capture golden vectors.
Never imported by product code or by the GPU box (the reference does not travel)."""
import os
import sys
import random

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DMARL_REFERENCE", "/root/reference")


class NS(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def to_ns(d):
    return NS({k: to_ns(v) for k, v in d.items()}) if isinstance(d, dict) else d


def activate():
    sys.dont_write_bytecode = True
    for p in (REF, os.path.join(HERE, "stubs")):
        if p not in sys.path:
            sys.path.insert(0, p)


def load_cfg(num_defender=8, map_size=(40, 40), blocks=5, variance=10, depth=1, max_steps=150):
    raw = yaml.safe_load(open(os.path.join(REF, "config.yaml")))
    cfg = to_ns({k: raw[k] for k in ("env", "sensor", "map", "attacker", "defender", "algo")})
    cfg.env.num_defender = num_defender
    cfg.env.max_steps = max_steps
    cfg.map.map_size = list(map_size)
    cfg.map.center = [map_size[0] // 2, map_size[1] // 2]
    cfg.map.num_obstacle_block = blocks
    cfg.map.variance = variance
    cfg.algo.depth = depth
    cfg.algo.use_reward_norm = True
    cfg.algo.learner_device = "cpu"
    cfg.algo.worker_device = "cpu"
    cfg.algo.evaluator_device = "cpu"
    return cfg


def seed_all(s):
    import torch
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)
