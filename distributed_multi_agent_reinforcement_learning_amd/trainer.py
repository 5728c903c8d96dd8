"""Data-parallel MAPPO training loop: one process per GPU, environment shards per rank, one RCCL all-reduce per epoch.

Same protocol as the reference's driver (main.py:41-172): rollouts -> per-learner GAE with its own advantage
normalisation -> gradient SUM over learners -> the same Adam step everywhere -> evaluation / checkpoints.  What was a
parameter-server sum of numpy lists through the Ray object store (main.py:105-129) is one `all_reduce(SUM)` of a
single flat fp32 bucket (~0.6 M elements) over xGMI; weights are never re-broadcast because every rank applies the
identical update.  Rollout data never leaves the GPU that produced it.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .evaluator import EvaluatorProc, draw_learning_curve
from .mappo import MAPPO
from .pursuit_env import Pursuit_Env


TUNED_GEMM_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable_tuned_gemms(path=TUNED_GEMM_FILE):
    """Library GEMM solution selection (PyTorch TunableOp over rocBLAS / hipBLASLt) from the committed tuning file
    produced by tools/tune_gemms.py; nothing is tuned at run time.  Returns True when the file was loaded."""
    if os.environ.get("DMARL_TUNED_GEMMS", "1") == "0" or not os.path.exists(path) or not torch.cuda.is_available():
        return False
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        ok = bool(tunable.read_file(path))
        # TunableOp rewrites its results file at interpreter exit: point it away from the committed table (one per process)
        import tempfile
        tunable.set_filename(os.path.join(tempfile.gettempdir(), f"dmarl_tunableop_{os.getpid()}.csv"), insert_device_ordinal=False)
        return ok
    except Exception:
        return False


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend=None):
    """Joins the process group of a multi-rank job (one process per GPU; `nccl` = RCCL over xGMI on a GPU box, `gloo` on CPU).  A
    single-rank job needs no group; an explicitly requested backend (argument or DMARL_DIST_BACKEND) is honoured all the same, so
    the RCCL path -- communicator, broadcast, in-place all-reduce of the device bucket -- can be exercised on one GPU."""
    rank, local_rank, world = dist_env()
    forced = backend or os.environ.get("DMARL_DIST_BACKEND")
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("DMARL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def allreduce_sum_(flat):
    """gradient SUM over learners (main.py:121-126), in place; without a process group (single rank) a no-op"""
    if dist.is_initialized():
        if flat.is_cuda and dist.get_backend() == "gloo":  # rehearsal of the N > 1 path on one GPU: stage through the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


class GradBucket:
    """The gradients of `params` as views of ONE persistent flat fp32 tensor (ac_parameters order): autograd accumulates straight
    into the bucket, the collective reduces the bucket in place and the optimiser reads the same memory -- no torch.cat before the
    all-reduce and no per-parameter clone after it.  Parameters that receive no gradient (an unused GRU under algo.use_rnn: false)
    contribute zeros, which is what the SUM over learners of `None` entries amounts to (main.py:121-126)."""

    def __init__(self, params):
        self.params = list(params)
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        self.attach()

    def attach(self):
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            o += n

    def zero(self):
        if any(p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4
               for p in self.params):
            self.attach()   # someone replaced a gradient tensor (e.g. a zero_grad(set_to_none=True)): re-attach the views
        self.flat.zero_()


def broadcast_weights_(modules, src=0):
    """initial weight sync from learner 0 (main.py:73-75)"""
    if dist.is_initialized():
        seen = set()
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                if id(t) not in seen:
                    seen.add(id(t))
                    if t.is_cuda and dist.get_backend() == "gloo":
                        host = t.data.cpu()
                        dist.broadcast(host, src=src)
                        t.data.copy_(host)
                    else:
                        dist.broadcast(t.data, src=src)


def save_checkpoint(actor, critic, cwd, suffix=""):
    """main.py:146-172: eight whole-module files; plus state_dicts (loadable without this package)."""
    os.makedirs(cwd, exist_ok=True)
    torch.save(actor, f"{cwd}/actor{suffix}.pth")
    torch.save(critic, f"{cwd}/critic{suffix}.pth")
    torch.save(actor.shared_net, f"{cwd}/actor_gnn{suffix}.pth")
    torch.save(critic.shared_net, f"{cwd}/critic_gnn{suffix}.pth")
    torch.save(actor.GRU, f"{cwd}/actor_gru{suffix}.pth")
    torch.save(critic.GRU, f"{cwd}/critic_gru{suffix}.pth")
    torch.save(actor.Mean, f"{cwd}/actor_mean{suffix}.pth")
    torch.save(critic.Mean, f"{cwd}/critic_mean{suffix}.pth")
    torch.save({"actor": actor.state_dict(), "critic": critic.state_dict()}, f"{cwd}/state_dicts{suffix}.pt")


class _Done:
    """stand-in for an already finished prefetch thread"""

    def join(self):
        return None


class Trainer:
    """One rank of the data-parallel job."""

    def __init__(self, cfg, num_envs=None, mini_batch_size=None, tuned_gemms=True):
        self.rank, self.local_rank, self.world = init_distributed()
        self.tuned_gemms = enable_tuned_gemms() if tuned_gemms else False
        self.cfg = cfg
        self.device = torch.device("cuda", self.local_rank % max(1, torch.cuda.device_count()))
        torch.cuda.set_device(self.device)
        self.num_envs = int(num_envs if num_envs is not None else cfg.runtime.num_envs)
        epi = int(cfg.algo.sample_epi_num)
        batch = self.num_envs * epi
        # main.py:48: mini_batch_size = round(num_workers * sample_epi_num / 10)
        self.mini_batch_size = int(mini_batch_size if mini_batch_size is not None else max(1, round(batch / 10)))
        self.env = Pursuit_Env(cfg, num_envs=self.num_envs, rank=self.rank, device=self.device)
        torch.manual_seed(int(cfg.runtime.get("seed", 0)))
        self.agent = MAPPO(cfg, batch, self.mini_batch_size, "Learner")
        self.agent.sample_rank = self.rank  # disjoint action-sampling streams per rank (mappo.MAPPO.sample_rank)
        self.bucket = GradBucket(self.agent.ac_parameters)   # .grad of every parameter lives in one flat tensor
        self.agent.grad_bucket = self.bucket
        broadcast_weights_([self.agent.actor, self.agent.critic])
        self.total_steps = 0
        self.iteration = 0
        self.last_log = (0.0, 0.0)

    def iterate(self):
        """rollout + `epochs` updates; returns (env_steps_this_iteration_all_ranks, exp_reward)."""
        cfg, agent = self.cfg, self.agent
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        exp_r, buffer, steps = agent.explore_env(self.env, int(cfg.algo.sample_epi_num))
        ev[1].record()
        self.env.prefetch_reset()  # the next episode's reset (device: kernels on the side stream; host: a thread) runs while the GPU runs the update
        self.total_steps += steps * self.world
        for _ in range(int(cfg.algo.epochs)):
            with torch.enable_grad():
                obj_c, obj_a, _, _ = agent.train(buffer, self.total_steps, return_grads=False)
            allreduce_sum_(self.bucket.flat)      # gradient SUM over ranks, in place in the bucket the optimiser reads
            agent.ac_optimizer.step()
            if cfg.algo.use_lr_decay:
                agent.lr_decay(self.total_steps)
            self.last_log = (obj_c, obj_a)
        ev[2].record()
        self.iteration += 1
        self.last_events = ev  # (rollout incl. host reset, update) timings: read after a synchronize
        return steps * self.world, exp_r

    # ---- resume bundle (the reference only saves weights: no optimiser state, step counter or RNG state) ----
    def save_resume(self, path):
        env, agent = self.env, self.agent
        init = env._take_prefetched()  # the next episode may already be drawn: it belongs to the snapshot
        # ... or already reset on the device (prefetch_reset): the bundle then carries the generator streams and tape positions as they
        # stood BEFORE that reset, so the resumed run's first reset re-creates the same episode
        pre = getattr(env, "_pre_reset", None) if getattr(env, "_dev_prefetch", None) is not None else None
        st = getattr(agent, "_rstate", None)
        bundle = dict(actor=agent.actor.state_dict(), critic=agent.critic.state_dict(), optimizer=agent.ac_optimizer.state_dict(),
                      total_steps=self.total_steps, iteration=self.iteration, reward_norm=env.sim.rn.cpu(),
                      tape_pos=(env.sim.meta[:, 2] if pre is None else pre["tape_pos"]).cpu(), sample_counter=None if st is None else st.counter.cpu(),
                      resetter=env.resetter.get_state() if pre is None else env.resetter.get_state(pre["resetter"]), next_init=init,
                      num_envs=self.num_envs, world=self.world, rank=self.rank)
        torch.save(bundle, path)
        if init is not None:
            env._prefetch = (_Done(), {"init": init})

    def load_resume(self, path):
        b = torch.load(path, map_location=self.device, weights_only=False)  # our own file (numpy blobs inside)
        if b["num_envs"] != self.num_envs or b["world"] != self.world:
            raise ValueError("resume bundle was written for another num_envs / world size")
        env, agent = self.env, self.agent
        if getattr(env, "_dev_prefetch", None) is not None:    # a device reset issued ahead belongs to the run being replaced: let it finish, drop it
            torch.cuda.current_stream().wait_stream(env.sim._side)
            env._dev_prefetch = env._pre_reset = None
        agent.actor.load_state_dict(b["actor"]); agent.critic.load_state_dict(b["critic"])
        agent.ac_optimizer.load_state_dict(b["optimizer"])
        self.total_steps, self.iteration = b["total_steps"], b["iteration"]
        agent.lr_decay(self.total_steps)
        env.sim.rn.copy_(b["reward_norm"])
        env.sim.meta[:, 2] = b["tape_pos"].to(self.device)
        env.resetter.set_state(b["resetter"])
        if b["sample_counter"] is not None:
            agent._rollout_state(env).counter.copy_(b["sample_counter"])
        env._prefetch = (_Done(), {"init": b["next_init"]}) if b["next_init"] is not None else None

    def last_breakdown_ms(self):
        ev = self.last_events
        return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])


def _device_weights(module):
    """snapshot of a module's state_dict on its device (no host round trip); the evaluator loads it on its own stream"""
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def train_agent_multiprocessing(cfg, max_iterations=None, num_eval_envs=16, eval_every=1, async_eval=None):
    """main.py:41-172 on the Trainer: runs until the evaluator says stop (total_step > max_train_steps).

    Evaluation is asynchronous like the reference's (main.py:135-158: `evaluator.run.remote(...)`, polled with `ray.wait(...,
    timeout=0.1)` while training goes on): the evaluator is a background actor of ray_shim -- its own host thread and HIP stream on
    rank 0's GPU -- that works on a device-side snapshot of the weights; the training loop only collects a finished evaluation and
    starts the next one with the current weights, it never waits for one.  No rank sits in a collective for the length of an
    evaluation: the per-iteration flag broadcast carries whatever verdict rank 0 holds at that moment.
    `runtime.async_eval: false` (or async_eval=False) evaluates inline on rank 0 (deterministic recorder rows per iteration)."""
    from . import ray_shim
    tr = Trainer(cfg)
    if async_eval is None:
        async_eval = bool(cfg.runtime.get("async_eval", True))
    evaluator = None
    if tr.rank == 0:
        evaluator = ray_shim.remote(EvaluatorProc).options(background=bool(async_eval)).remote(cfg, num_eval_envs)
    cwd = cfg.algo.save_cwd
    if_train = True
    eval_run_ref = None

    def collect(obj):
        """main.py:139-156: the evaluator's verdict, and the checkpoint when the greedy return did not get worse"""
        ok, ref_list = obj[0], obj[1]
        if len(ref_list) > 0:
            actor, critic, recorder = (ray_shim.get(r) for r in ref_list)
            os.makedirs(cwd, exist_ok=True)
            np.save(cwd + "/recorder.npy", recorder)
            draw_learning_curve(recorder=np.array(recorder), cwd=cwd)
            save_checkpoint(actor, critic, cwd)
        return ok

    while if_train:
        t0 = time.time()
        steps, exp_r = tr.iterate()
        if tr.rank == 0:
            print(f"iteration {tr.iteration}: {steps} env-steps in {time.time() - t0:.2f}s")
            if tr.iteration % eval_every == 0:
                if eval_run_ref is not None:
                    done, _ = ray_shim.wait([eval_run_ref], num_returns=1, timeout=0 if async_eval else None)
                    if done:
                        if_train = collect(ray_shim.get(done[0]))
                        eval_run_ref = None
                if eval_run_ref is None and if_train:
                    eval_run_ref = evaluator.run.remote(_device_weights(tr.agent.actor), _device_weights(tr.agent.critic), tr.total_steps, exp_r,
                                                        tr.last_log)
                    if not async_eval:   # inline: this iteration's verdict and checkpoint, not the previous evaluation's
                        if_train = collect(ray_shim.get(eval_run_ref))
                        eval_run_ref = None
        if tr.world > 1:
            flag = torch.tensor([1 if if_train else 0], device=tr.device if dist.get_backend() != "gloo" else "cpu")
            dist.broadcast(flag, src=0)
            if_train = bool(flag.item())
        if max_iterations is not None and tr.iteration >= max_iterations:
            break
    if tr.rank == 0:
        if eval_run_ref is not None:   # the evaluation still in flight belongs to the run's record
            collect(ray_shim.get(eval_run_ref))
        save_checkpoint(tr.agent.actor, tr.agent.critic, cwd, "_final")
        np.save(cwd + "/recorder.npy", ray_shim.get(evaluator.get_recorder.remote()))
        evaluator._shutdown()
    return tr
