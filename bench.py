"""Headline benchmark: env-steps/s of the MAPPO hot path (batched rollout + PPO update) on N MI355X GPUs.

One "step" = one training iteration of every rank: reset N environments, roll them out for T ticks with the policy
in the loop (fused HIP env tick + batched actor/critic forward), GAE, PPO-clip/value-clip update over all N*T
samples, gradient all-reduce (RCCL) and the Adam step.  value = env-steps of all ranks / wall time.
Synthetic data: seeded random maps, random-init weights (no datasets or checkpoints exist for this workload).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3] [--num-envs E] [--no-cpu-baseline]
Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_env_step(P, W, H, O):
    """f64 state, fp32 observations, u8 grid, i16 boundary-index map"""
    reads = 32 * P + 40 + 8 + 16 + 4 * P + 3 * W * H
    writes = 32 * P + 40 + 16 * P + 16 + 4 * P * P + 4 * P + 4 * P * O + 4 * P + 1
    return reads + writes


def measure_env_tick(trainer, n_ticks):
    """Launch durations of the fused env tick (HIP events on the launch stream; random actions, policy excluded):
    the regular tick k_tick<step,observe,evader> and the replan variant that runs every `difficulty` ticks."""
    env = trainer.env
    N, P, D = env.num_envs, env.num_defender, env.pe_cfg.difficulty
    env.reset()
    overlap, env.sim.overlap_replan = env.sim.overlap_replan, False  # per-kernel durations: keep everything on one stream
    obs = env.sim.new_obs()
    reward = torch.zeros((N, P), dtype=torch.float32, device=trainer.device)
    acts = torch.randint(0, 9, (n_ticks, N, P), dtype=torch.int32, device=trainer.device)
    env.observe(obs)
    env.attacker_step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_ticks + 1)]
    ev[0].record()
    kinds = []
    for t in range(n_ticks):
        env.tick(acts[t], obs, reward)
        kinds.append(env.sim.t_host % D == 0)
        ev[t + 1].record()
    torch.cuda.synchronize()
    env.sim.overlap_replan = overlap
    dur = [ev[t].elapsed_time(ev[t + 1]) * 1e-3 for t in range(n_ticks)]
    reg = [d for d, k in zip(dur, kinds) if not k]
    rep = [d for d, k in zip(dur, kinds) if k]
    return sum(reg) / max(1, len(reg)), (sum(rep) / len(rep) if rep else float("nan")), sum(dur) / n_ticks


def measure_compute_kernels(trainer, cfg):
    """Isolated timings of the two compute-bound hand-written kernels at the update's mini-batch shape, priced against the
    fp32 MFMA / vector peak (157.3 TFLOP/s, MI355X_MICROARCH.md) with the ALGORITHMIC flops of SURVEY 8(d)."""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    dev = trainer.device
    T, P, O, E = cfg.env.max_steps, cfg.env.num_defender, cfg.map.num_max_obstacle, cfg.algo.embedding_dim
    mb = trainer.mini_batch_size

    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    out = {}
    if E == 128 and cfg.algo.rnn_hidden_dim == 128:
        B = mb * P
        h0 = torch.zeros(1, B, 128, device=dev)
        gi = torch.randn(T, B, 384, device=dev)
        o = torch.empty(T, B, 128, device=dev)
        L = ops.load_library()
        import ctypes as C
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        w, b = torch.randn(384, 128, device=dev) * 0.08, torch.zeros(384, device=dev)
        t = timeit(lambda: L.gru_seq_fwd(T, B, 128, ptr(gi), ptr(w), ptr(b), ptr(h0[0]), ptr(o), None, st))
        fl = 2.0 * T * B * 128 * 384
        out["gru_seq_fwd"] = {"bound": "mfma", "kernel": "k_gru_seq_fwd (recurrent GEMM h W_hh^T + gates, T steps in one launch)",
                              "achieved": round(fl / t / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(fl / t / 157.3e12, 4),
                              "us_per_launch": round(t * 1e6, 1), "rows": B, "steps": T}
        Kr = mb * T * P  # rows of one mini-batch: the weight gradient of a GRU projection reduces over all of them
        ga = torch.randn(Kr, 384, device=dev); xa = torch.randn(Kr, 128, device=dev)
        t = timeit(lambda: ops.wgrad(ga, xa))
        fl = 2.0 * Kr * 384 * 128
        out["wgrad"] = {"bound": "mfma", "kernel": "k_wgrad<3,1> + reduce (dW_ih [384][128] = dgi^T x over the mini-batch rows, split-K)",
                        "achieved": round(fl / t / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(fl / t / 157.3e12, 4),
                        "us_per_launch": round(t * 1e6, 1), "rows": Kr, "hbm_GBps": round((384 + 128) * 4.0 * Kr / t / 1e9, 1)}
        del ga, xa
        Br = trainer.num_envs * P
        from types import SimpleNamespace  # raw tensors: constructing torch.nn.GRU on the device initialises MIOpen
        gm = SimpleNamespace(num_layers=1, weight_ih_l0=torch.randn(384, 128, device=dev) * 0.08, weight_hh_l0=torch.randn(384, 128, device=dev) * 0.08,
                             bias_ih_l0=torch.zeros(384, device=dev), bias_hh_l0=torch.zeros(384, device=dev))
        xr = torch.randn(1, Br, 128, device=dev); hr = torch.randn(1, Br, 128, device=dev)
        with torch.no_grad():
            t = timeit(lambda: ops.gru(xr, hr, gm), n=20)
        fl = 2.0 * Br * 128 * 768
        out["gru_cell"] = {"bound": "mfma", "kernel": "k_gru_cell (one rollout GRU layer step: both projections + gates, one launch)",
                           "achieved": round(fl / t / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(fl / t / 157.3e12, 4),
                           "us_per_launch": round(t * 1e6, 1), "rows": Br}
    R = mb * T
    p = torch.rand(R, P, 4, device=dev) * 40; q = torch.rand(mb, O, 4, device=dev) * 40
    W = torch.randn(E, 4, device=dev) * 0.3; bb = torch.zeros(E, device=dev)
    with torch.no_grad():
        t = timeit(lambda: ops.msg_agg(p, q, None, None, W, bb, ops.ADJ_ONES, None, T))
    fl = R * P * O * (2 * 4 * E + 3 * E)
    out["msg_agg_fwd"] = {"bound": "mfma", "kernel": "k_msg_agg_fwd<8> (critic obstacle relation, dense)", "achieved": round(fl / t / 1e12, 2),
                          "peak": 157.3, "unit": "TFLOP/s", "frac": round(fl / t / 157.3e12, 4), "us_per_launch": round(t * 1e6, 1), "rows": R,
                          "note": "algorithmic flops of the reference formulation W(p_i - q_j); the kernel evaluates c_i - d_j (fewer)"}
    return out


def load_pmc_traffic(N):
    """HBM bytes per launch of the regular tick from the committed rocprofv3 PMC passes (profiles/r01_tick_pmc.json,
    produced with tools/profile_tick.py; FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as read)."""
    path = os.path.join(ROOT, "profiles", "r01_tick_pmc.json")
    try:
        j = json.load(open(path))
        if j.get("num_envs") == N:
            return j["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(cfg, n_envs, T, threads):
    """The same path on the host: CPU oracle env (C) + plain-torch oracle model, rollout + one update, bounded sample."""
    from oracle import model_oracle as mo
    from oracle import pe_oracle, reset_oracle
    import random
    torch.set_num_threads(threads)
    from distributed_multi_agent_reinforcement_learning_amd.model import build_actor_critic
    P, (W, H), O, d = cfg.env.num_defender, cfg.map.map_size, cfg.map.num_max_obstacle, cfg.algo.depth
    E, Hd, L = cfg.algo.embedding_dim, cfg.algo.rnn_hidden_dim, cfg.algo.num_layers
    torch.manual_seed(0)
    actor, critic = build_actor_critic(cfg, "cpu")
    sd_a = {k: v.detach().clone() for k, v in actor.state_dict().items()}
    sd_c = {k: v.detach().clone() for k, v in critic.state_dict().items()}
    ocfg = pe_oracle.make_config(W=W, H=H, P=P, O=O, max_steps=T, tape_len=16)
    buf = {k: torch.zeros(n_envs, T, *s) for k, s in (("p_state", (P, 4)), ("e_state", (1, 4)), ("o_state", (O, 4)), ("p_adj", (P, P)),
                                                      ("e_adj", (P, 1)), ("o_adj", (P, O)), ("a_n", (P,)), ("a_logprob_n", (P,)),
                                                      ("r", (P,)), ("active", (P,)))}
    buf["actor_historical_embedding"] = torch.zeros(n_envs, T + d, P, E)
    buf["critic_historical_embedding"] = torch.zeros(n_envs, T + d, P, E)
    buf["v_n"] = torch.zeros(n_envs, T + 1, P)
    t0 = time.time()
    with torch.no_grad():
        for n in range(n_envs):
            random.seed(n); np.random.seed(n)
            r0 = reset_oracle.reset_oracle(W, H, P, cfg.map.num_obstacle_block, list(cfg.map.center), cfg.map.variance, tape_len=16)
            oe = pe_oracle.OracleEnv(ocfg)
            oe.load(r0["grid"], r0["obs_xy"], r0["defenders"], r0["evader"], r0["target"], r0["tape"])
            k = len(r0["obs_xy"])
            o_t = torch.zeros(k, 4); o_t[:, :2] = torch.as_tensor(r0["obs_xy"], dtype=torch.float32)
            buf["o_state"][n, :, :k] = o_t
            ha = torch.zeros(L, P, Hd); hc = torch.zeros(L, P, Hd)
            shared = [torch.zeros(P, E) for _ in range(d)]
            a_cur = torch.zeros(P, E); c_cur = torch.zeros(P, E)
            for t in range(T):
                ps, es, pa, ea, oa = oe.observe()
                oe.evader_step()
                obs = dict(p=torch.as_tensor(ps), e=torch.as_tensor(es), o=o_t, p_adj=torch.as_tensor(pa), e_adj=torch.as_tensor(ea),
                           o_adj=torch.as_tensor(oa[:, :k]))
                if d:
                    shared = (shared + [a_cur, c_cur])[-d:]
                hops = [shared[d - 1 - j] for j in range(d)]
                prob, ha, a_cur = mo.actor_step(sd_a, obs, hops, ha, d)
                val, hc, c_cur = mo.critic_step(sd_c, obs, hops, hc, d)
                a = torch.multinomial(prob, 1).squeeze(-1)
                r, ok, done = oe.step(a.numpy().astype(np.int32))
                buf["p_state"][n, t] = obs["p"]; buf["e_state"][n, t] = obs["e"]; buf["p_adj"][n, t] = obs["p_adj"]
                buf["e_adj"][n, t] = obs["e_adj"]; buf["o_adj"][n, t] = torch.as_tensor(oa)
                buf["a_n"][n, t] = a.float(); buf["a_logprob_n"][n, t] = torch.log(prob.gather(-1, a[:, None]).squeeze(-1))
                buf["r"][n, t] = torch.as_tensor(oe.reward_norm(r), dtype=torch.float32); buf["active"][n, t] = 1.0
                buf["v_n"][n, t] = val.flatten()
                buf["actor_historical_embedding"][n, t + d] = a_cur; buf["critic_historical_embedding"][n, t + d] = c_cur
    t_roll = time.time() - t0
    t0 = time.time()
    mo.train(sd_a, sd_c, buf, d, max(1, n_envs // 2), cfg.algo.gamma, cfg.algo.lamda, cfg.algo.epsilon, cfg.algo.entropy_coef)
    t_train = time.time() - t0
    return n_envs * T / (t_roll + t_train), t_roll, t_train


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3"])
    ap.add_argument("--num-envs", type=int, default=None, help="environments per GPU (default: the config's)")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tick-samples", type=int, default=150)
    args = ap.parse_args()

    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer, dist_env

    rank, local_rank, world = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    ov = {}
    if args.num_envs:
        ov["runtime.num_envs"] = args.num_envs
    if args.max_steps:
        ov["env.max_steps"] = args.max_steps
    cfg = baseline_config(args.config, **ov)
    tr = Trainer(cfg)  # weak scaling: every rank owns runtime.num_envs environments
    N, T, P = tr.num_envs, cfg.env.max_steps, cfg.env.num_defender
    W, H = cfg.map.map_size
    O = cfg.map.num_max_obstacle

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not a benchmark step: the first iteration captures the rollout hipGraph and builds every autograd / optimiser
    # buffer, and the caching allocator still grows its segment pool (device mallocs, +8 GB at cfg3) during the iteration
    # after it; both happen once per process.  The W warm-up steps and the K timed steps below are then steady state.
    for _ in range(2):
        tr.iterate()
    for _ in range(args.warmup):
        tr.iterate()
    barrier()
    t0 = time.perf_counter()
    env_steps = 0
    for _ in range(args.steps):
        s, _ = tr.iterate()
        env_steps += s
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tr.device if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    rollout_ms, update_ms = tr.last_breakdown_ms()
    # roofline of the environment tick kernel: algorithmic bytes / measured launch duration
    t_tick, t_replan, t_avg = measure_env_tick(tr, args.tick_samples)
    bytes_per_step = algorithmic_bytes_per_env_step(P, W, H, O)
    achieved = N * bytes_per_step / t_tick / 1e9
    roofline = {"bound": "hbm", "kernel": "k_tick<step,observe,evader,no-replan> (csrc/pe_env.hip)", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": load_pmc_traffic(N) if args.config in ("cfg2", "cfg3") else None,
                "bytes_per_env_step": bytes_per_step, "us_per_launch": round(t_tick * 1e6, 2), "env_steps_per_launch": N,
                "replan_tick_us_per_launch": round(t_replan * 1e6, 2), "episode_avg_tick_us": round(t_avg * 1e6, 2)}
    extra = measure_compute_kernels(tr, cfg) if rank == 0 else {}

    out = None
    if rank == 0:
        out = {
            "metric": "env-steps/sec (whole node), pursuit-evasion 8-agent 4096-env", "value": round(env_steps / dt, 1),
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 environment / f32 policy", "data": "synthetic (seeded random maps, random-init weights)",
            "config": {"workload": f"{args.config}: pursuit_evasion_game {P} defenders, {W}x{H} map, {N} envs/GPU, T={T}, "
                                   f"DHGN depth {cfg.algo.depth} + 2-layer GRU actor/critic, rollout + PPO update",
                       "envs_per_gpu": N, "episode_steps": T, "mini_batch_size": tr.mini_batch_size, "parallelism": f"dp{world}"},
            "ppo_updates_per_s": round(args.steps * cfg.algo.epochs / dt, 4),
            "breakdown_ms": {"rollout_incl_reset": round(rollout_ms, 1), "gae_ppo_update_allreduce_adam": round(update_ms, 1)},
            "roofline": roofline,
            "roofline_compute_kernels": extra,
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            v, t_roll, t_train = cpu_baseline(cfg, 6, min(T, 150), threads)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "env-steps/s", "cores": threads, "kind": "port",
                                   "sample": f"6 envs x {min(T, 150)} steps: oracle C env + plain-torch model rollout ({t_roll:.1f}s) "
                                             f"+ one PPO update pass ({t_train:.1f}s) on the host"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
