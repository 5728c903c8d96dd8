"""GPU tests of the reference-API layer (Learner / Worker / EvaluatorProc / trainer) and of a 2-rank rehearsal."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import ROOT
from tests.helpers import product_cfg

pytestmark = pytest.mark.gpu


def small_cfg(tmp_path, **extra):
    return product_cfg(4, 20, 20, T=10, depth=1, blocks=2, variance=4,
                       **{"algo.save_cwd": str(tmp_path / "model"), "algo.max_train_steps": 2000, "runtime.num_envs": 8, **extra})


def test_reference_protocol_worker_learner(tmp_path):
    """main.py:79-132 with .remote() removed: weights down, buffers up, gradient lists summed over learners, same update."""
    from distributed_multi_agent_reinforcement_learning_amd.runner import Learner, Worker
    cfg = small_cfg(tmp_path)
    torch.manual_seed(0)
    learners = [Learner(cfg, 8, 4, i) for i in range(2)]
    workers = [Worker(i, cfg, num_envs=8) for i in range(2)]
    aw, cw = learners[0].get_weights()
    for lr in learners:
        lr.set_weights(aw, cw)
    total = 0
    for lr, wk in zip(learners, workers):
        exp_r, steps = lr.collect_buffer([wk.run(aw, cw)])
        assert steps == 8 * 10
        total += steps
    grads = [lr.compute_and_get_gradients(total) for lr in learners]
    a_sum = [np.stack(g).sum(0) for g in zip(*[g[1] for g in grads])]
    c_sum = [np.stack(g).sum(0) for g in zip(*[g[2] for g in grads])]
    for lr in learners:
        lr.set_gradients_and_update(a_sum, c_sum, total)
    w0, w1 = learners[0].get_weights(), learners[1].get_weights()
    for k in w0[0]:
        assert torch.equal(w0[0][k], w1[0][k]), k          # identical update on every learner
        if "weight" in k and "GRU" in k:
            assert not torch.equal(w0[0][k], aw[k])        # and the weights moved
    assert len(grads[0][1]) == len(list(learners[0].agent.actor.parameters()))
    # two workers with different ranks explore different maps
    assert not torch.equal(workers[0].env.sim.grid, workers[1].env.sim.grid)


def test_training_loop_checkpoints_and_recorder(tmp_path):
    from distributed_multi_agent_reinforcement_learning_amd import trainer
    from distributed_multi_agent_reinforcement_learning_amd.model import build_actor_critic
    cfg = small_cfg(tmp_path)
    tr = trainer.train_agent_multiprocessing(cfg, max_iterations=2, num_eval_envs=4)
    cwd = cfg.algo.save_cwd
    for name in ("actor", "critic", "actor_gnn", "critic_gnn", "actor_gru", "critic_gru", "actor_mean", "critic_mean"):
        assert os.path.exists(f"{cwd}/{name}_final.pth"), name
    rec = np.load(cwd + "/recorder.npy")
    assert rec.shape[1] == 6 and rec.shape[0] >= 1             # (total_step, avg_r, std_r, exp_r, objC, objA)
    assert os.path.exists(cwd + "/learning_curve.jpg")
    sd = torch.load(cwd + "/state_dicts_final.pt", weights_only=True)
    actor, critic = build_actor_critic(cfg, "cpu")
    actor.load_state_dict(sd["actor"]); critic.load_state_dict(sd["critic"])
    whole = torch.load(cwd + "/actor_final.pth", weights_only=False)
    assert list(whole.state_dict().keys()) == list(actor.state_dict().keys())
    assert tr.total_steps == 2 * 8 * 10


def test_two_rank_bench_rehearsal(tmp_path):
    """`python bench.py --gpus 2` -- the driver's own command form -- starts its two ranks itself (torch.distributed.run child)
    before touching the GPU; here both ranks share the one GPU, so gloo stands in for RCCL.  The N > 1 path -- shard seeds per
    rank, gradient all-reduce, max-over-ranks timing, one JSON line from rank 0 -- runs end to end."""
    env = dict(os.environ, DMARL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "cfg1",
           "--num-envs", "16", "--max-steps", "12", "--tick-samples", "12"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0 and "gloo" in j["config"]["collective"]
    assert abs(j["value"] - 2 * 16 * 12 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3
    # a launcher whose world size contradicts --gpus is an error, not silently ignored
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "disagrees" in (bad.stderr + bad.stdout)


def test_prefetched_reset_equals_inline_reset(tmp_path):
    """Overlapping the host reset of the next episode with the update does not change the episode sequence."""
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    cfg = small_cfg(tmp_path, **{"runtime.device_reset": False})
    acts = torch.randint(0, 9, (6, 8, 4), dtype=torch.int32, device="cuda")
    grids = []
    for prefetch in (False, True):
        env = Pursuit_Env(cfg, num_envs=8, seeds=list(range(40, 48)))
        seq = []
        for ep in range(3):
            env.reset()
            obs = env.observe(); env.attacker_step()
            for t in range(6):
                env.tick(acts[t], obs, torch.zeros(8, 4, device="cuda"))
            if prefetch:
                env.prefetch_reset()
            seq.append((env.sim.grid.clone(), env.sim.eva.clone(), env.sim.target.clone()))
        grids.append(seq)
    for a, b in zip(*grids):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_resume_bundle_continues_the_run(tmp_path):
    """weights + Adam moments + step counter + reward-normaliser + every RNG stream: a resumed run continues like the original."""
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
    cfg = small_cfg(tmp_path)
    a = Trainer(cfg)
    a.iterate(); a.iterate()
    a.save_resume(str(tmp_path / "resume.pt"))
    a.iterate()
    b = Trainer(cfg)
    b.load_resume(str(tmp_path / "resume.pt"))
    assert b.total_steps == 2 * 8 * 10 and b.iteration == 2
    b.iterate()
    assert b.total_steps == a.total_steps
    assert torch.equal(a.env.sim.grid, b.env.sim.grid) and torch.equal(a.env.sim.target, b.env.sim.target)   # same maps drawn
    assert torch.equal(a.env.sim.rn[:, 0], b.env.sim.rn[:, 0])
    for (k, p), (_, q) in zip(a.agent.actor.state_dict().items(), b.agent.actor.state_dict().items()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), k
    lr_a, lr_b = a.agent.ac_optimizer.param_groups[0]["lr"], b.agent.ac_optimizer.param_groups[0]["lr"]
    assert lr_a == lr_b


@pytest.mark.timeout(180)
@pytest.mark.parametrize("device_reset", [True, False])
def test_empty_map_trains(tmp_path, device_reset):
    """map.num_obstacle_block = 0: no boundary obstacles in any environment (n_obs == 0, all-zero o_adj, the critic's padded
    obstacle slots are all that relation 2 sees).  Rollout + update + evaluation stay finite and move the weights."""
    import math
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
    cfg = small_cfg(tmp_path, **{"map.num_obstacle_block": 0, "runtime.device_reset": device_reset})
    tr = Trainer(cfg)
    before = [p.detach().clone() for p in tr.agent.ac_parameters]
    for _ in range(2):
        steps, exp_r = tr.iterate()
    torch.cuda.synchronize()
    assert int(tr.env.n_obs.max().item()) == 0
    buf = tr.agent.minibuffer.buffer
    assert float(buf["o_adj"].abs().sum()) == 0.0
    assert all(math.isfinite(float(v)) for v in tr.last_log) and math.isfinite(float(exp_r))
    assert all(torch.isfinite(p).all() for p in tr.agent.ac_parameters)
    assert any(not torch.equal(p.detach(), b) for p, b in zip(tr.agent.ac_parameters, before))
    tr.env.check_status()


def test_default_loop_reports_kernel_status_bits(tmp_path):
    """with the default on-device reset nothing used to read the kernels' status bits.  A one-entry target tape is
    exhausted within an episode on a 20 x 20 map; the production loop (Trainer.iterate) must raise, not train on."""
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
    cfg = small_cfg(tmp_path, **{"runtime.tape_len": 1, "env.max_steps": 150, "runtime.num_envs": 64, "runtime.device_reset": True})
    tr = Trainer(cfg)
    with pytest.raises(RuntimeError, match="tape exhausted"):
        for _ in range(3):
            tr.iterate()


def test_status_bits_survive_the_device_reset(tmp_path):
    """the bits are sticky across pe_env_reset (k_reset carries them into the new meta record), so a caller that drives the
    environment by hand and only looks after the next reset still sees them"""
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    cfg = small_cfg(tmp_path, **{"runtime.tape_len": 1, "env.max_steps": 150, "runtime.device_reset": True})
    env = Pursuit_Env(cfg, num_envs=32)
    env.reset()
    obs = env.observe(); env.attacker_step()
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(149):
        env.tick(torch.randint(0, 9, (32, 4), generator=g, device="cuda", dtype=torch.int32), obs, torch.zeros(32, 4, device="cuda"))
    assert int(env.sim.status().max()) & 1
    with pytest.raises(RuntimeError, match="tape exhausted"):
        env.reset()
    assert int(env.sim.status().max()) & 1            # still there in the fresh episode's meta record


@pytest.mark.timeout(120)
def test_device_reset_reports_an_impossible_placement_instead_of_hanging(tmp_path):
    """8 defenders on 12 x 31 with 8 blocks: the reference's placement loop never ends for some seeds; k_reset's loops are bounded
    (PE_RESET_MAX_DRAWS, same bound as the host resetter and the oracle) and flag the environment."""
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    cfg = product_cfg(8, 12, 31, T=10, blocks=8, variance=3, **{"runtime.device_reset": True})
    env = Pursuit_Env(cfg, num_envs=8, seeds=list(range(1000, 1008)))
    with pytest.raises(RuntimeError, match="gave up"):
        env.reset()
    d = env.sim.defenders_aos()
    assert bool(((d[..., 0] >= 0) & (d[..., 0] <= 11) & (d[..., 1] >= 0) & (d[..., 1] <= 30)).all())   # valid fallback state


def test_ranks_draw_independent_exploration_noise(tmp_path):
    """every data-parallel rank sampled actions from the same Philox stream.  Rank r now starts its counter at
    r << 40: equal probabilities give different samples on different ranks, rank 0 keeps the stream of the goldens."""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    from distributed_multi_agent_reinforcement_learning_amd.mappo import MAPPO, _RolloutState
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    cfg = small_cfg(tmp_path)
    env = Pursuit_Env(cfg, num_envs=8)
    states = []
    for r in (0, 1):
        agent = MAPPO(cfg, 8, 4, "Learner")
        agent.sample_rank = r
        states.append(_RolloutState(agent, env))
    assert int(states[0].counter.item()) == 0 and int(states[1].counter.item()) == 1 << 40
    prob = torch.full((4096, 9), 1.0 / 9, device="cuda")
    a0, _ = ops.categorical_sample(prob, 7, 0, counter=states[0].counter.clone())
    a1, _ = ops.categorical_sample(prob, 7, 0, counter=states[1].counter.clone())
    legacy, _ = ops.categorical_sample(prob, 7, 0)
    assert torch.equal(a0, legacy) and (a0 != a1).float().mean() > 0.8


@pytest.mark.timeout(300)
def test_ray_style_driver_on_the_shim_with_a_background_evaluator(tmp_path):
    """The constructs of the reference's driver (main.py:55-158: `.options(...).remote(...)`, `method.remote(...)`, `ray.get`,
    `ray.wait(..., timeout)`, refs passed as arguments, a non-blocking evaluator) on the real Learner / Worker / EvaluatorProc
    through ray_shim; the evaluator is a background actor (own thread + stream) working on a snapshot while training goes on.
    (The reference's loop text itself runs on the shim in tests/test_ray_shim.py, where the reference tree exists.)"""
    import time
    import distributed_multi_agent_reinforcement_learning_amd.ray_shim as ray
    from distributed_multi_agent_reinforcement_learning_amd import evaluator as ev_mod, runner
    cfg = small_cfg(tmp_path)
    Learner, Worker, EvaluatorProc = ray.remote(runner.Learner), ray.remote(runner.Worker), ray.remote(ev_mod.EvaluatorProc)
    torch.manual_seed(0)
    learners = [Learner.options(resources={f"node_{n}": 0.001}).remote(cfg, 8, 4, n) for n in range(2)]
    workers = [[Worker.options(resources={f"node_{n}": 0.001}).remote(n, cfg, num_envs=8)] for n in range(2)]
    evaluator = EvaluatorProc.options(resources={"node_-1": 0.001}, background=True).remote(cfg, 4)
    actor_weights, critic_weights = ray.get(learners[0].get_weights.remote())
    for lr in learners:
        lr.set_weights.remote(actor_weights, critic_weights)
    total_steps, eval_run_ref, finished, if_train = 0, None, [], True
    for it in range(3):
        worker_run_ref = [[w.run.remote(actor_weights, critic_weights) for w in workers[n]] for n in range(2)]
        learner_run_ref = [learners[n].collect_buffer.remote(worker_run_ref[n]) for n in range(2)]   # a list of refs as an argument
        exp_r = 0.0
        while len(learner_run_ref) > 0:
            ret, learner_run_ref = ray.wait(learner_run_ref, num_returns=1, timeout=0.1)
            if len(ret) > 0:
                r, steps = ray.get(ret)[0]
                exp_r += r / 2
                total_steps += steps
        grads = ray.get([lr.compute_and_get_gradients.remote(total_steps) for lr in learners])
        a_sum = [np.stack(g).sum(axis=0) for g in zip(*[g[1] for g in grads])]
        c_sum = [np.stack(g).sum(axis=0) for g in zip(*[g[2] for g in grads])]
        for lr in learners:
            lr.set_gradients_and_update.remote(a_sum, c_sum, total_steps)
        actor_weights, critic_weights = ray.get(learners[0].get_weights.remote())
        if eval_run_ref is None:
            t0 = time.monotonic()
            eval_run_ref = [evaluator.run.remote(actor_weights, critic_weights, total_steps, exp_r, grads[0][0])]
            assert time.monotonic() - t0 < 0.5                       # returns at once: the evaluation runs beside the loop
        else:
            return_ref, eval_run_ref = ray.wait(object_refs=eval_run_ref, num_returns=1, timeout=0.1)
            if len(return_ref):
                obj = ray.get(return_ref)[0]
                if_train, ref_list = obj[0], obj[1]
                finished.append(obj)
                if len(ref_list) > 0:
                    actor = ray.get(ref_list[0])
                    assert hasattr(actor, "shared_net") and hasattr(actor, "GRU") and hasattr(actor, "Mean")
                eval_run_ref = [evaluator.run.remote(actor_weights, critic_weights, total_steps, exp_r, grads[0][0])]
    assert total_steps == 3 * 2 * 8 * 10
    last = ray.get(eval_run_ref)[0]
    recorder = ray.get(evaluator.get_recorder.remote())
    assert len(recorder) == len(finished) + 1 >= 1 and len(recorder[0]) == 6 and last[0] in (True, False)
    assert all(np.isfinite(row).all() for row in np.asarray(recorder, np.float64))
    w0, w1 = ray.get(learners[0].get_weights.remote()), ray.get(learners[1].get_weights.remote())
    for k in w0[0]:
        assert torch.equal(w0[0][k], w1[0][k]), k
    evaluator._shutdown()


def test_async_and_inline_evaluation_record_the_same_first_row(tmp_path):
    """trainer.train_agent_multiprocessing with the background evaluator (default) and with runtime.async_eval = false: same
    training (the evaluator only reads snapshots), the first recorder row -- evaluation of the weights after iteration 1 -- equal."""
    from distributed_multi_agent_reinforcement_learning_amd import trainer
    recs = []
    for mode in (True, False):
        cfg = small_cfg(tmp_path / ("a" if mode else "b"), **{"runtime.async_eval": mode})
        tr = trainer.train_agent_multiprocessing(cfg, max_iterations=3, num_eval_envs=4)
        recs.append((np.load(cfg.algo.save_cwd + "/recorder.npy"), [p.detach().clone() for p in tr.agent.ac_parameters]))
    assert len(recs[1][0]) == 3 and 1 <= len(recs[0][0]) <= 3
    assert np.allclose(recs[0][0][0], recs[1][0][0], rtol=1e-5, atol=1e-6), (recs[0][0][0], recs[1][0][0])
    for p, q in zip(recs[0][1], recs[1][1]):
        assert torch.equal(p, q)


RCCL_SMOKE = r"""
import json, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from distributed_multi_agent_reinforcement_learning_amd import trainer
from tests.helpers import product_cfg
cfg = product_cfg(4, 20, 20, T=10, depth=1, blocks=2, variance=4, **{"algo.max_train_steps": 2000, "runtime.num_envs": 8})
tr = trainer.Trainer(cfg)                      # init_process_group("nccl", world_size=1) + broadcast_weights_ over RCCL
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
flat = tr.bucket.flat
assert flat.is_cuda and all(p.grad.data_ptr() >= flat.data_ptr() for p in tr.agent.ac_parameters)
w0 = torch.cat([p.detach().reshape(-1) for p in tr.agent.ac_parameters]).clone()
calls = []
real = dist.all_reduce
def counted(t, *a, **k):
    calls.append((t.data_ptr(), t.is_cuda, t.numel()))
    return real(t, *a, **k)
dist.all_reduce = counted
steps, exp_r = tr.iterate()                    # rollout + update + in-place all-reduce of the device bucket + Adam
torch.cuda.synchronize()
g = flat.clone()
trainer.allreduce_sum_(flat)                   # SUM over one rank: the bucket is unchanged, bit for bit
torch.cuda.synchronize()
w1 = torch.cat([p.detach().reshape(-1) for p in tr.agent.ac_parameters])
print(json.dumps(dict(steps=steps, calls=len(calls), on_bucket=all(c == (flat.data_ptr(), True, flat.numel()) for c in calls),
                      identity=bool(torch.equal(g, flat)), grad_nonzero=float(g.abs().sum()), moved=float((w1 - w0).abs().max()),
                      finite=bool(torch.isfinite(w1).all()))))
dist.destroy_process_group()
"""


def test_one_rank_rccl_smoke():
    """The `nccl` (= RCCL) branch of the data-parallel path on hardware, with one rank: the communicator builds, the initial weight
    broadcast (main.py:73-75) and the per-epoch gradient all-reduce (main.py:121-129) run on the DEVICE bucket in place
    (trainer.py init_distributed / broadcast_weights_ / allreduce_sum_), one Trainer.iterate() completes, the group is destroyed."""
    env = dict(os.environ, DMARL_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    out = subprocess.run([sys.executable, "-c", RCCL_SMOKE, ROOT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["steps"] == 8 * 10 and j["calls"] == 2 and j["on_bucket"] and j["identity"], j     # one per epoch + the explicit call
    assert j["grad_nonzero"] > 0 and j["moved"] > 0 and j["finite"], j


def test_device_reset_issued_ahead_equals_inline_reset(tmp_path):
    """Pursuit_Env.prefetch_reset with the device resetter: the next episode's reset kernels run on the side stream (under the PPO update
    in the Trainer) and reset() only waits for them -- same maps, same initial conditions, same episode as the inline reset; an injected
    initial condition after the prefetch is refused; a resume bundle written while a reset is pending carries the state BEFORE it."""
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
    cfg = small_cfg(tmp_path, **{"runtime.device_reset": True})
    acts = torch.randint(0, 9, (6, 8, 4), dtype=torch.int32, device="cuda")
    seqs = []
    for prefetch in (False, True):
        env = Pursuit_Env(cfg, num_envs=8, seeds=list(range(40, 48)))
        seq = []
        for ep in range(3):
            env.reset()
            obs = env.observe(); env.attacker_step()
            for t in range(6):
                env.tick(acts[t], obs, torch.zeros(8, 4, device="cuda"))
            seq.append((env.sim.grid.clone(), env.sim.defs.clone(), env.sim.eva.clone(), env.sim.target.clone(), obs["p_state"].clone()))
            if prefetch:
                env.prefetch_reset()
                busy = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")   # work on the main stream beside it
                if ep == 1:
                    with pytest.raises(RuntimeError, match="already run"):
                        env.reset(init={"grid": None})
        seqs.append(seq)
    for a, b in zip(*seqs):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    # Trainer: a bundle saved after iterate() (reset pending) resumes into the same third iteration
    a = Trainer(cfg)
    a.iterate(); a.iterate()
    assert a.env._dev_prefetch is not None
    a.save_resume(str(tmp_path / "resume_dev.pt"))
    a.iterate()
    b = Trainer(cfg)
    b.iterate()                                   # b has its own pending reset when the bundle arrives: it is dropped
    b.load_resume(str(tmp_path / "resume_dev.pt"))
    b.iterate()
    assert b.total_steps == a.total_steps
    assert torch.equal(a.env.sim.grid, b.env.sim.grid) and torch.equal(a.env.sim.target, b.env.sim.target) and torch.equal(a.env.sim.rn[:, 0], b.env.sim.rn[:, 0])
    for (k, p), (_, q) in zip(a.agent.actor.state_dict().items(), b.agent.actor.state_dict().items()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), k
