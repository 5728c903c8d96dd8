"""ray_shim ("a thin .remote / ray.get shim keeps main.py's loop text runnable").

The first test executes the reference's OWN training-loop text -- the body of main.py:41-172 `train_agent_multiprocessing`, read
from /root/reference at run time (build container only; nothing of it is stored here) -- with `ray` bound to the shim and the
three actor classes bound to light stand-ins that implement the runner / evaluator protocol on numpy, and checks the protocol
the loop drove.  The GPU suite runs the same constructs on the real Learner / Worker / EvaluatorProc (tests/test_runner_gpu.py)."""
import os
import textwrap
import threading
import time

import numpy as np
import pytest

from distributed_multi_agent_reinforcement_learning_amd import ray_shim as ray
from distributed_multi_agent_reinforcement_learning_amd.config import load_config

REF_MAIN = "/root/reference/main.py"


class _Net:
    """stand-in for an actor / critic module: the three sub-objects main.py:146-156 saves separately"""

    def __init__(self, name):
        self.shared_net, self.GRU, self.Mean = name + "_gnn", name + "_gru", name + "_mean"


def test_refs_wait_and_background_actor():
    class Slow:
        def __init__(self, delay):
            self.delay = delay
            self.calls = []

        def work(self, x):
            time.sleep(self.delay)
            self.calls.append((x, threading.current_thread().name))
            return x * 2

        def boom(self):
            raise ValueError("inside the actor")

    A = ray.remote(Slow)
    inline = A.remote(0.0)
    r = inline.work.remote(ray.put(21))                      # ObjectRef arguments are resolved
    assert ray.get(r) == 42 and ray.get([r, r]) == [42, 42] and ray.get(7) == 7
    bad = inline.boom.remote()                               # the error surfaces at get()
    with pytest.raises(ValueError):
        ray.get(bad)
    bg = A.options(resources={"node_0": 0.001}, background=True).remote(0.15)
    t0 = time.monotonic()
    refs = [bg.work.remote(k) for k in range(3)]             # returns at once
    assert time.monotonic() - t0 < 0.1
    ready, rest = ray.wait(refs, num_returns=1, timeout=0.01)
    assert ready == [] and len(rest) == 3
    ready, rest = ray.wait(refs, num_returns=1, timeout=2.0)
    assert len(ready) == 1 and ready[0] is refs[0] and len(rest) == 2   # an actor runs its calls in submission order
    assert ray.get(refs) == [0, 2, 4]
    with pytest.raises(ValueError):
        ray.get(bg.boom.remote())
    assert ray.get(bg.work.remote(ray.put(5))) == 10
    with pytest.raises(TypeError):
        bg.work(1)
    bg._shutdown()

    @ray.remote(num_cpus=1)
    def f(a, b=1):
        return a + b
    assert ray.get(f.remote(1, b=ray.put(2))) == 3


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree exists in the build container only")
def test_reference_main_loop_text_runs_on_the_shim(tmp_path):
    src = open(REF_MAIN).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.startswith("def train_agent_multiprocessing("))
    body = []
    for l in src[start + 1:]:
        if l.strip() and not l.startswith((" ", "\t")):
            break
        body.append(l)
    code = "def _reference_loop(cfg):\n" + "\n".join(body) + "\n"
    log = {"runs": 0, "collect": 0, "grads": 0, "updates": [], "evals": [], "weights_set": 0, "draw": 0}
    n_params = 3

    class Learner:
        def __init__(self, cfg, batch_size, mini_batch_size, learner_id):
            assert mini_batch_size == round(batch_size / 10)                 # main.py:48
            self.id, self.w = learner_id, np.zeros(n_params)
            self.batch_size = batch_size

        def get_weights(self):
            return {"w": self.w.copy()}, {"w": self.w.copy() + 1}

        def set_weights(self, a, c):
            log["weights_set"] += 1
            self.w = a["w"].copy()

        def collect_buffer(self, worker_run_ref):
            assert len(worker_run_ref) == self.batch_size and all(isinstance(r, tuple) for r in worker_run_ref)   # refs arrive resolved
            log["collect"] += 1
            return float(np.mean([r[0] for r in worker_run_ref])), int(sum(r[2] for r in worker_run_ref))

        def compute_and_get_gradients(self, total_steps):
            log["grads"] += 1
            g = [np.full(2, 1.0 + self.id), None and 0 or np.full(3, 10.0 * (1 + self.id))]
            return (0.5 + self.id, -0.1), g, [np.ones(1) * (1 + self.id)]

        def set_gradients_and_update(self, a_grad, c_grad, total_steps):
            log["updates"].append((self.id, [x.copy() for x in a_grad], [x.copy() for x in c_grad], total_steps))
            self.w = self.w + 1

        def save(self):
            return [_Net("actor"), _Net("critic")]

    class Worker:
        def __init__(self, idx, cfg):
            self.idx = idx

        def run(self, actor_weights, critic_weights):
            log["runs"] += 1
            return (float(self.idx % 3), ("buffer", self.idx), 150)

    class EvaluatorProc:
        def __init__(self, cfg, num_cpus_eval):
            self.break_step, self.recorder = cfg.algo.max_train_steps, []

        def run(self, aw, cw, total_step, exp_r, logging_tuple):
            log["evals"].append((total_step, exp_r, tuple(logging_tuple)))
            self.recorder.append((total_step, 0.0, 0.0, exp_r, *logging_tuple))
            return [total_step <= self.break_step, [ray.put(_Net("actor")), ray.put(_Net("critic")), ray.put(list(self.recorder))]]

        def get_recorder(self):
            return self.recorder

    import torch

    def draw_learning_curve(recorder=None, cwd=None, **kw):
        log["draw"] += 1
    g = {"ray": ray, "np": np, "torch": torch, "time": time, "os": os, "draw_learning_curve": draw_learning_curve,
         "Learner": ray.remote(Learner), "Worker": ray.remote(Worker), "EvaluatorProc": ray.remote(EvaluatorProc), "print": lambda *a, **k: None}
    exec(compile(code, REF_MAIN, "exec"), g)
    cfg = load_config(REF_MAIN.replace("main.py", "config.yaml"), **{"algo.save_cwd": str(tmp_path), "algo.max_train_steps": 60000})
    workers_total = (128 - 8) + (80 - 8)                                   # main.py:43-45
    g["_reference_loop"](cfg)
    iters = len(log["updates"]) // 2
    assert iters >= 2 and log["runs"] == iters * workers_total and log["collect"] == 2 * iters and log["grads"] == 2 * iters
    assert log["weights_set"] == 2
    for k, (lid, a_grad, c_grad, total) in enumerate(log["updates"]):
        assert np.array_equal(a_grad[0], np.full(2, 3.0)) and np.array_equal(a_grad[1], np.full(3, 30.0)) and np.array_equal(c_grad[0], np.ones(1) * 3)   # SUM over the two learners (main.py:121-126)
        assert total == (k // 2 + 1) * workers_total * 150
    assert log["evals"][0][0] == workers_total * 150 and log["evals"][-1][0] > 60000      # the evaluator's verdict ended the loop
    for name in ("actor", "critic", "actor_gnn", "critic_gnn", "actor_gru", "critic_gru", "actor_mean", "critic_mean"):
        assert os.path.exists(tmp_path / f"{name}.pth") and os.path.exists(tmp_path / f"{name}_final.pth"), name
    assert os.path.exists(tmp_path / "recorder.npy") and log["draw"] >= 2
