"""Headline benchmark: env-steps/s of the MAPPO hot path (batched rollout + PPO update) on N MI355X GPUs.

One "step" = one training iteration of every rank: reset N environments, roll them out for T ticks with the policy
in the loop (fused HIP env tick + batched actor/critic forward), GAE, PPO-clip/value-clip update over all N*T
samples, gradient all-reduce (RCCL) and the Adam step.  value = env-steps of all ranks / wall time.
Synthetic data: seeded random maps, random-init weights (no datasets or checkpoints exist for this workload).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4] [--num-envs E] [--no-cpu-baseline] [--no-secondary]
                  [--scaling weak|strong [--total-envs E]]

`--gpus N` with N > 1 starts its own N rank processes (a `torch.distributed.run` child, one rank per GPU over RCCL) before
anything in this process touches the GPU; under an outer `torch.distributed.run` (WORLD_SIZE set) it is one of the ranks.
The headline line is BASELINE config 2 (depth 0, "GRU"); the same line carries `configs.cfg3` (DHGN depth 3, the
configuration with end-to-end reference parity) measured the same way right after it.
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TF = 157.3    # fp32 vector == fp32 MFMA peak
PMC_FILES = {"packed": [os.path.join(ROOT, "profiles", n) for n in ("r04_tick_pmc.json", "r03_tick_pmc.json", "r02_tick_pmc.json")],   # newest first
             "fp32": [os.path.join(ROOT, "profiles", n) for n in ("r04_tick_pmc_fp32.json", "r03_tick_pmc_fp32.json")]}


def algorithmic_bytes_per_env_step(P, W, H, O):
    """f64 state, fp32 observations, u8 grid, i16 boundary-index map"""
    reads = 32 * P + 40 + 8 + 16 + 4 * P + 3 * W * H
    writes = 32 * P + 40 + 16 * P + 16 + 4 * P * P + 4 * P + 4 * P * O + 4 * P + 1
    return reads + writes


def tick_kernel_hash():
    """identifies the environment kernel source a PMC capture belongs to"""
    h = hashlib.sha256()
    for rel in ("distributed_multi_agent_reinforcement_learning_amd/csrc/pe_env.hip", "include/pe_env.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def load_pmc_traffic(N, us_live, layout="packed"):
    """HBM bytes per launch of the regular tick from the committed rocprofv3 PMC passes (tools/pmc_summary.py; FETCH_SIZE
    doubled per MI355X_MICROARCH.md, WRITE_SIZE as read).  Counters cannot be collected inside this process, so the record
    is only used when it was captured on THIS kernel source (hash) at this size and its launch duration agrees with the
    live one; otherwise traffic is null and the reason is reported."""
    j, why = None, "no PMC record under profiles/"
    for path in PMC_FILES[layout]:
        try:
            cand = json.load(open(path))
        except Exception:
            continue
        if cand.get("num_envs") != N:
            why = f"PMC record is for {cand.get('num_envs')} environments"
        elif cand.get("kernel_hash") != tick_kernel_hash():
            why = "PMC record is stale: csrc/pe_env.hip / include/pe_env.h changed since it was captured"
        else:
            j = cand
            break
    if j is None:
        return None, why
    us_rec = j.get("us_per_launch")
    if us_rec and abs(us_rec - us_live) > 0.25 * us_live:
        return None, f"PMC record's launch duration ({us_rec:.1f} us) disagrees with the live run ({us_live:.1f} us)"
    return j["traffic_bytes_per_launch"], None


def measure_env_tick(trainer, n_ticks, packed=True):
    """Launch durations of the fused env tick, HIP events on the launch stream (random actions, policy excluded).
    Regular ticks: the D-1 launches between two replans are captured once as a hipGraph and replayed, so the events bracket
    back-to-back launches with no host time in between (the Python launch path costs more than the 11 us kernel); the figure
    is elapsed / launches, i.e. the kernel plus its ~1 us same-stream kernel boundary.  Replan ticks (every `difficulty`
    ticks, ~1 ms) are launched eagerly.  packed: LiDAR rows leave the kernel bit-packed (what the rollout consumes) or as
    the reference's fp32 (N, P, O) rows."""
    import torch
    env = trainer.env
    sim = env.sim
    N, P, D = env.num_envs, env.num_defender, env.pe_cfg.difficulty
    env.reset()
    overlap, sim.overlap_replan = sim.overlap_replan, False  # per-kernel durations: keep everything on one stream
    obs = sim.new_obs(packed=packed)
    reward = torch.zeros((N, P), dtype=torch.float32, device=trainer.device)
    acts = torch.randint(0, 9, (D, N, P), dtype=torch.int32, device=trainer.device)
    env.observe(obs)
    env.attacker_step()
    torch.cuda.synchronize()
    t_host0, time_step0 = sim.t_host, env.time_step
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):        # the D-1 regular ticks that follow a replan tick (t_host = 1 .. D-1)
        for k in range(D - 1):
            env.tick(acts[k], obs, reward)
    sim.t_host, env.time_step = t_host0, time_step0   # capture only recorded the launches
    blocks = max(1, n_ticks // D)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * blocks + 1)]
    ev[0].record()
    for b in range(blocks):
        graph.replay()
        sim.t_host += D - 1
        env.time_step += D - 1
        ev[2 * b + 1].record()
        env.tick(acts[D - 1], obs, reward)      # t_host becomes a multiple of D: the replan variant
        assert sim.t_host % D == 0
        ev[2 * b + 2].record()
    torch.cuda.synchronize()
    sim.overlap_replan = overlap
    reg = [ev[2 * b].elapsed_time(ev[2 * b + 1]) * 1e-3 / (D - 1) for b in range(blocks)]
    rep = [ev[2 * b + 1].elapsed_time(ev[2 * b + 2]) * 1e-3 for b in range(blocks)]
    exp = sim.meta[:, 5].float()
    if int(sim.status().max()):
        raise RuntimeError("environment kernel status bits set during the tick measurement")
    return dict(regular=sum(reg) / len(reg), replan=sum(rep) / len(rep), replan_max=max(rep),
                avg=(sum(reg) * (D - 1) + sum(rep)) / (blocks * D), astar_exp_mean=float(exp.mean()), astar_exp_max=float(exp.max()))


def measure_compute_kernels(trainer, cfg):
    """Isolated timings of the compute-bound hand-written kernels at the update's mini-batch shape against the 157.3 TFLOP/s
    fp32 MFMA / vector peak, ALGORITHMIC flops of SURVEY 8(d) (and the executed flops where the kernel evaluates fewer)."""
    import ctypes as C
    import torch
    from types import SimpleNamespace
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

    split = ops.MATMUL_MODE == "split_bf16"
    SPLIT_PEAK_TF = 2500.0 / 6.0   # fp32-equivalent rate of six bf16 MFMAs per product at the dense bf16 peak

    def entry(kernel, fl, t, bound="mfma", split_kernel=False, **kw):
        """algorithmic fp32 flops over time; `peak` is the pipe the kernel runs on: the fp32 MFMA peak, or for the split-bf16 kernels
        the bf16 peak / 6 (frac_of_fp32_mfma_peak keeps the figure the fp32 kernels are priced by: it may exceed 1)"""
        peak = SPLIT_PEAK_TF if split_kernel else FP32_PEAK_TF
        e = dict(bound=bound, kernel=kernel, achieved=round(fl / t / 1e12, 2), peak=round(peak, 1), unit="TFLOP/s",
                 frac=round(fl / t / (peak * 1e12), 4), us_per_launch=round(t * 1e6, 1), **kw)
        if split_kernel:
            e["frac_of_fp32_mfma_peak"] = round(fl / t / (FP32_PEAK_TF * 1e12), 4)
        return e
    out = {}
    if E == 128 and cfg.algo.rnn_hidden_dim == 128:
        B = mb * P
        h0 = torch.zeros(1, B, 128, device=dev)
        gi = torch.randn(T, B, 384, device=dev)
        o = torch.empty(T, B, 128, device=dev)
        L = ops.load_library()
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        w, b = torch.randn(384, 128, device=dev) * 0.08, torch.zeros(384, device=dev)
        seq_split = ops.SEQ_MODE == "split_bf16"
        k_fwd, k_bwd = ("k_gru_seq_fwd_sb", "k_gru_seq_bwd_sb") if seq_split else ("k_gru_seq_fwd2", "k_gru_seq_bwd2")
        f_fwd, f_bwd = (L.gru_seq_split_fwd_multi, L.gru_seq_split_bwd_multi) if seq_split else (L.gru_seq_fwd_multi, L.gru_seq_bwd_multi)
        FWD_B, BWD_B = 4096.0, 5120.0     # HBM bytes per sequence row and step: gi + out + four saved gate planes | saved gates + h_prev + dout + dgi + dnr

        def seq_records(Bs):
            """forward / backward records of len(Bs) independent layers (one weight set: timing only)"""
            keep, fa, ba = [], (ops.GruSeqNet * len(Bs))(), (ops.GruSeqBwdNet * len(Bs))()
            for k_, b_ in enumerate(Bs):
                ts = dict(gi=torch.randn(T, b_, 384, device=dev), h0=torch.zeros(b_, 128, device=dev), out=torch.empty(T, b_, 128, device=dev),
                          save=torch.empty(L.gru_seq_save_elems(T, b_), device=dev), dout=torch.randn(T, b_, 128, device=dev),
                          dgi=torch.empty(T, b_, 384, device=dev), dnr=torch.empty(T, b_, 128, device=dev), dh0=torch.empty(b_, 128, device=dev),
                          dbi=torch.empty(384, device=dev), dbh=torch.empty(384, device=dev),
                          ws=torch.empty(L.gru_seq_bwd_workspace(b_), dtype=torch.uint8, device=dev))
                keep.append(ts)
                a_ = fa[k_]
                a_.gi, a_.w_hh, a_.b_hh, a_.h0, a_.out, a_.save, a_.B = (ts["gi"].data_ptr(), w.data_ptr(), b.data_ptr(), ts["h0"].data_ptr(), ts["out"].data_ptr(),
                                                                          ts["save"].data_ptr(), b_)
                a_ = ba[k_]
                a_.dout, a_.save, a_.out, a_.h0, a_.w_hh, a_.dgi, a_.dgh, a_.dnr = (ts["dout"].data_ptr(), ts["save"].data_ptr(), ts["out"].data_ptr(), ts["h0"].data_ptr(),
                                                                                    w.data_ptr(), ts["dgi"].data_ptr(), None, ts["dnr"].data_ptr())
                a_.dh0, a_.db_ih, a_.db_hh, a_.workspace, a_.B = ts["dh0"].data_ptr(), ts["dbi"].data_ptr(), ts["dbh"].data_ptr(), ts["ws"].data_ptr(), b_
            return keep, fa, ba

        def seq_entry(kernel, t, rows, nbytes, **kw):
            e = entry(kernel, 2.0 * T * rows * 128 * 384, t, split_kernel=seq_split, rows=rows, steps=T, **kw)
            e.update(hbm_bytes_per_row_step=nbytes, hbm_GBps=round(nbytes * rows * T / t / 1e9, 1), frac_of_hbm=round(nbytes * rows * T / t / 1e9 / HBM_PEAK_GBS, 4))
            return e
        keep, fa, ba = seq_records([B])
        t = timeit(lambda: f_fwd(1, C.cast(fa, C.c_void_p), T, B, 128, 0, st))
        out["gru_seq_fwd"] = seq_entry(f"{k_fwd} (recurrent product h W_hh^T + gates of one layer of one mini-batch, T steps in one launch)", t, B, FWD_B)
        del keep
        # the form the update launches (MAPPO._train_grouped): this layer of every mini-batch of the epoch and both networks at once
        N_envs = trainer.num_envs
        Bs = [(min(n0 + mb, N_envs) - n0) * P for n0 in range(0, N_envs, mb) for _ in range(2)]
        if 2 <= len(Bs) <= ops.GRU_MULTI_MAX_NETS:
            keep, fa, ba = seq_records(Bs)
            wgs = sum((b_ + 15) // 16 for b_ in Bs)
            t = timeit(lambda: f_fwd(len(Bs), C.cast(fa, C.c_void_p), T, max(Bs), 128, 0, st))
            out["gru_seq_fwd_grouped"] = seq_entry(f"{k_fwd}, {len(Bs)} layers (every mini-batch of the epoch x actor, critic) in one launch: {wgs} workgroups", t, sum(Bs), FWD_B)
            t = timeit(lambda: f_bwd(len(Bs), C.cast(ba, C.c_void_p), T, max(Bs), 128, 0, st))
            out["gru_seq_bwd_grouped"] = seq_entry(f"{k_bwd}, the same {len(Bs)} layers backward in one launch (the largest single launch of an iteration)", t, sum(Bs), BWD_B)
            del keep
        Kr = mb * T * P  # rows of one mini-batch: the weight gradient of a GRU projection reduces over all of them
        ga = torch.randn(Kr, 384, device=dev); xa = torch.randn(Kr, 128, device=dev)
        t = timeit(lambda: ops.wgrad(ga, xa))
        out["wgrad"] = entry(("k_sb_wgrad<3,1> (exact bf16 operand splits)" if ops.WGRAD_MODE == "split_bf16" else "k_wgrad<3,1>")
                             + " + reduce (dW_ih [384][128] = dgi^T x over the mini-batch rows, split-K)", 2.0 * Kr * 384 * 128, t,
                             split_kernel=ops.WGRAD_MODE == "split_bf16", rows=Kr, hbm_GBps=round((384 + 128) * 4.0 * Kr / t / 1e9, 1))
        del ga, xa
        Br = trainer.num_envs * P
        xr, hr, ho = torch.randn(Br, 128, device=dev), torch.randn(Br, 128, device=dev), torch.empty(Br, 128, device=dev)
        wi, wh, bi, bh = w, torch.randn(384, 128, device=dev) * 0.08, b, torch.zeros(384, device=dev)
        if ops.CELL_MODE == "split_bf16":   # what the tick launches: actor's and critic's cell of one layer, split-bf16 kernel, C ABI directly
            arr = (ops.GruCellNet * 2)()
            for a_ in arr:
                a_.x, a_.h_prev, a_.h_out, a_.w_ih, a_.w_hh, a_.b_ih, a_.b_hh = (t_.data_ptr() for t_ in (xr, hr, ho, wi, wh, bi, bh))
            t = timeit(lambda: L.gru_cell_split_fwd_multi(2, C.cast(arr, C.c_void_p), Br, 128, st), n=20)
            out["gru_cell"] = entry("k_gru_cell_sb (one rollout GRU layer step of actor and critic: both projections + gates, one launch, exact bf16 "
                                    "operand splits)", 2 * 2.0 * Br * 128 * 768, t, split_kernel=True, rows=2 * Br)
        else:
            t = timeit(lambda: L.gru_cell_fwd(Br, 128, ptr(xr), ptr(hr), ptr(wi), ptr(wh), ptr(bi), ptr(bh), ptr(ho), st), n=20)  # C ABI directly
            out["gru_cell"] = entry("k_gru_cell (one rollout GRU layer step: both projections + gates, one launch)", 2.0 * Br * 128 * 768, t, rows=Br)
    # DHGN message + mean aggregation: the kernels never form the (rows, P, K, E) message, so flops of the reference formulation are
    # not a hardware figure for them; they are priced by the bytes they must move (inputs + the (rows, P, relations, E) output)
    R = mb * T
    p = torch.rand(R, P, 4, device=dev) * 40; e = torch.rand(R, 1, 4, device=dev) * 40
    q = torch.zeros(mb, O, 4, device=dev); q[:, :, :2] = torch.randint(0, 40, (mb, O, 2), device=dev).float()
    adj_p = (torch.rand(R, P, P, device=dev) < 0.6).float(); adj_e = (torch.rand(R, P, 1, device=dev) < 0.3).float()
    adj_o = ops.pack_adj_bits((torch.rand(R, P, O, device=dev) < 0.1).float())
    Ws = [torch.randn(E, 8, device=dev) * 0.3, torch.randn(E, 4, device=dev) * 0.3, torch.randn(E, 4, device=dev) * 0.3]
    bs = [torch.zeros(E, device=dev) for _ in range(3)]

    def hbm_entry(kernel, nbytes, t, **kw):
        return dict(bound="hbm", kernel=kernel, achieved=round(nbytes / t / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), us_per_launch=round(t * 1e6, 1), bytes_per_launch=int(nbytes), **kw)
    in_bytes = 4 * (p.numel() + e.numel() + q.numel() + adj_p.numel() + adj_e.numel() + adj_o.numel())
    with torch.no_grad():
        t = timeit(lambda: ops.msg_agg3(p, e, q, adj_p, adj_e, adj_o, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2], False, None, T))
    out["msg_agg3_fwd_actor"] = hbm_entry("k_msgw3_fwd: the actor's three relations of one mini-batch in one launch (packed LiDAR rows walked edge by edge)",
                                          in_bytes + 4.0 * R * P * 3 * E, t, rows=R)
    L = ops.load_library()
    ptr = lambda t_: C.c_void_p(t_.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    oo = torch.empty(R, P, E, device=dev)
    t = timeit(lambda: L.dhgn_msg_agg_ones_sorted_fwd(R, P, O, E, ptr(p), p.stride(0), ptr(q), q.stride(0), T, ptr(Ws[2]), ptr(bs[2]), ptr(oo), E, None, None, st))
    out["msg_agg_ones_sorted_fwd_critic"] = hbm_entry("k_msg_ones_sorted_fwd: the update's critic obstacle relation (all-ones adjacency): per-episode sort + prefix "
                                                      "sums + a two-level rank search (16 register splitters, one 16-key bucket), O(log K) per pair", 4.0 * (p.numel() + q.numel()) + 4.0 * R * P * E, t, rows=R)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the same path on the host, one process per core, bounded sample
def cpu_worker(config, n_envs, T, seed0):
    """One host process, one thread: CPU oracle env (C) + plain-torch oracle model; rollout of n_envs episodes + one PPO
    update pass over them.  Prints one JSON line.  Never touches the GPU."""
    import random
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from oracle import model_oracle as mo
    from oracle import pe_oracle, reset_oracle
    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.model import build_actor_critic
    cfg = baseline_config(config)
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
            random.seed(seed0 + n); np.random.seed(seed0 + n)
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
    mo.train(sd_a, sd_c, buf, d, 1, cfg.algo.gamma, cfg.algo.lamda, cfg.algo.epsilon, cfg.algo.entropy_coef)
    t_train = time.time() - t0
    print(json.dumps({"cpu_worker": True, "steps": n_envs * T, "t_roll": t_roll, "t_train": t_train}), flush=True)


def cpu_baseline(config, T, envs_per_proc=4):
    """One single-threaded process per host core, all at once (the reference runs one Ray worker per core, main.py:42-62)."""
    cores = max(1, min(64, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # torch's autograd engine opens the GPU on the first backward() of ANY process of a ROCm build (device-count query); the
    # workers are CPU processes and the box admits few processes with the device open: oracle/nogpu_shim.c answers that query
    shim = os.path.join(ROOT, "oracle", "libnogpu_shim.so")
    if not os.path.exists(shim):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libnogpu_shim.so"], stdout=subprocess.DEVNULL)
    env["LD_PRELOAD"] = shim + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", config, str(envs_per_proc), str(T), str(1000 * (i + 1))],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT) for i in range(cores)]
    res = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        lines = [l for l in so.splitlines() if l.startswith('{"cpu_worker"')]
        if p.returncode != 0 or not lines:
            raise RuntimeError("cpu baseline worker failed: " + se[-1500:])
        res.append(json.loads(lines[-1]))
    wall = max(r["t_roll"] + r["t_train"] for r in res)
    steps = sum(r["steps"] for r in res)
    return dict(value=round(steps / wall, 2), unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} single-threaded processes at once (one per host core), each {envs_per_proc} envs x {T} steps: oracle C env + "
                       f"plain-torch model rollout (max {max(r['t_roll'] for r in res):.1f}s) + one PPO update pass over its episodes "
                       f"(max {max(r['t_train'] for r in res):.1f}s); value = all steps / slowest process")


# ---------------------------------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """--gpus N > 1 outside a launcher: start the N ranks as a child job (nothing here has touched the GPU) and relay it."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def _matmul_mode():
    from distributed_multi_agent_reinforcement_learning_amd import ops
    return ops.MATMUL_MODE


def run_config(name, args, with_roofline):
    """2 set-up iterations (graph capture, allocator growth: once per process) + W warm-up + exactly K timed iterations,
    bracketed by barrier + synchronize, MAX over ranks.  Returns the result dict (rank 0 fills the JSON from it)."""
    import torch
    import torch.distributed as dist
    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer, dist_env
    rank, local_rank, world = dist_env()
    ov = {}
    if args.num_envs:
        ov["runtime.num_envs"] = args.num_envs
    if args.max_steps:
        ov["env.max_steps"] = args.max_steps
    if args.update_group is not None:
        ov["runtime.update_group"] = args.update_group
    cfg = baseline_config(name, **ov)
    if args.scaling == "strong":   # fixed total: the job's environments are divided over the ranks (BASELINE configs 4 / 5 state totals)
        total = int(args.total_envs or cfg.runtime.num_envs)
        if total % world:
            raise SystemExit(f"--scaling strong: {total} environments do not divide over {world} ranks")
        cfg.runtime.num_envs = total // world
    if torch.cuda.is_available():
        import gc
        torch.cuda.set_device(local_rank % torch.cuda.device_count())   # this rank's GPU (what Trainer selects), not device 0
        gc.collect()
        torch.cuda.empty_cache()   # a previous configuration's cached blocks (the grouped epoch holds 125-171 GB) go back to the driver
    tr = Trainer(cfg)  # weak scaling: every rank owns runtime.num_envs environments
    N, T, P = tr.num_envs, cfg.env.max_steps, cfg.env.num_defender
    W, H = cfg.map.map_size
    O = cfg.map.num_max_obstacle

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(2 + args.warmup):
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
    res = dict(value=round(env_steps / dt, 1), ms_per_step=round(dt / args.steps * 1e3, 2),
               ppo_updates_per_s=round(args.steps * cfg.algo.epochs / dt, 4),
               breakdown_ms={"rollout_incl_reset": round(rollout_ms, 1), "gae_ppo_update_allreduce_adam": round(update_ms, 1)},
               workload=f"{name}: pursuit_evasion_game {P} defenders, {W}x{H} map, {N} envs/GPU, T={T}, DHGN depth {cfg.algo.depth} + "
                        f"2-layer GRU actor/critic, rollout + PPO update",
               envs_per_gpu=N, episode_steps=T, mini_batch_size=tr.mini_batch_size, backend=(dist.get_backend() if world > 1 else None),
               update_group=tr.agent.update_group, hbm_peak_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1), matmul=_matmul_mode(),
               epochs=int(cfg.algo.epochs))
    # GEMM-shaped algorithmic work of one iteration (SURVEY 8d F_net without the message terms, which the kernels do not execute as
    # flops): per network and env-step; the rollout runs each network forward once, every epoch of the update forward + backward (3x)
    E_, H_, A_, d_ = cfg.algo.embedding_dim, cfg.algo.rnn_hidden_dim, cfg.env.action_dim, cfg.algo.depth
    f_net = (3 * 2 * P * E_ * E_ + 2 * P * (3 * E_ + 4) * E_ + d_ * (2 * P * P * E_ + 2 * P * E_ * E_ + 2 * P * 2 * E_ * E_)
             + (cfg.algo.num_layers * 2 * (2 * P * E_ * 3 * H_) if cfg.algo.get("use_rnn", True) else 0) + 2 * P * H_ * A_)
    flops_iter = 2 * f_net * (1 + 3 * int(cfg.algo.epochs)) * N * T
    res["roofline_iteration"] = {"bound": "mfma", "what": "GEMM-shaped algorithmic flops of one iteration (both networks: rollout forward + update forward and "
                                 "backward) over the measured iteration time, against the fp32 MFMA peak (the price of the fp32 formulation; with runtime.matmul: split_bf16 part of "
                                 "the products runs on the bf16 pipe at up to 2.67 x that rate)", "flops_per_net_env_step": f_net, "flops_per_iteration": flops_iter,
                                 "achieved": round(flops_iter / (dt / args.steps) / 1e12, 2), "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                                 "frac": round(flops_iter / (dt / args.steps) / 1e12 / FP32_PEAK_TF, 4),
                                 # the same flops against the pipe the default mode runs them on: six bf16 MFMAs per fp32 product, 2 500 / 6 TFLOP/s
                                 "split_pipe_peak": round(2500.0 / 6.0, 1),
                                 "frac_of_split_pipe": round(flops_iter / (dt / args.steps) / 1e12 / (2500.0 / 6.0), 4)}
    if with_roofline:
        tk = measure_env_tick(tr, args.tick_samples)
        tk_f32 = measure_env_tick(tr, args.tick_samples, packed=False)
        bytes_f32 = algorithmic_bytes_per_env_step(P, W, H, O)                       # reference fp32 observation layout
        bytes_packed = bytes_f32 - 4 * P * O + 4 * P * (((O + 31) // 32 + 3) // 4 * 4)   # LiDAR rows as PE_RASER_ROW_WORDS(O) words
        traffic, why = load_pmc_traffic(N, tk["regular"] * 1e6)
        traffic_f32, why_f32 = load_pmc_traffic(N, tk_f32["regular"] * 1e6, "fp32")
        # Every figure divides the bytes of ONE layout by the launch duration of the kernel variant that writes THAT layout.
        # `roofline` is the variant SURVEY 8(d) grades (fp32 rows, the C ABI's pe_obs_out.o_adj); the timed training loop issues the
        # packed variant, reported next to it with its own bytes and time and with the PMC-measured HBM rate.
        ach_f32 = N * bytes_f32 / tk_f32["regular"] / 1e9
        ach_packed = N * bytes_packed / tk["regular"] / 1e9
        res["roofline"] = {"bound": "hbm", "kernel": "k_tick<step,observe,evader,no-replan> writing the reference's fp32 observation layout (csrc/pe_env.hip)",
                           "achieved": round(ach_f32, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_f32 / HBM_PEAK_GBS, 5),
                           "traffic": traffic_f32,
                           "hbm_actual_GBps": None if traffic_f32 is None else round(traffic_f32 / tk_f32["regular"] / 1e9, 1),
                           "bytes_per_env_step": bytes_f32, "us_per_launch": round(tk_f32["regular"] * 1e6, 2), "env_steps_per_launch": N,
                           "episode_avg_tick_us": round(tk_f32["avg"] * 1e6, 2)}
        res["roofline_packed_rows"] = {"bound": "hbm", "kernel": "the same kernel with LiDAR rows bit-packed (32 B per defender instead of 704 B): the launch "
                                       "the rollout issues", "achieved": round(ach_packed, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(ach_packed / HBM_PEAK_GBS, 5), "traffic": traffic, "bytes_per_env_step": bytes_packed,
                                       "us_per_launch": round(tk["regular"] * 1e6, 2), "env_steps_per_launch": N,
                                       "episode_avg_tick_us": round(tk["avg"] * 1e6, 2),
                                       "hbm_actual_GBps": None if traffic is None else round(traffic / tk["regular"] / 1e9, 1),
                                       "hbm_actual_frac": None if traffic is None else round(traffic / tk["regular"] / 1e9 / HBM_PEAK_GBS, 5),
                                       "fp32_equivalent_GBps": round(N * bytes_f32 / tk["regular"] / 1e9, 1),
                                       "fp32_equivalent_note": "reference-layout bytes over the packed launch's time: a cross-layout throughput equivalent, "
                                                               "not a bandwidth (rounds 1-2 reported this as roofline.achieved)"}
        if why:
            res["roofline_packed_rows"]["traffic_note"] = why
        if why_f32:
            res["roofline"]["traffic_note"] = why_f32
        res["roofline_replan_tick"] = {"kernel": "k_tick<..., replan> (rescan + weighted A* of every evader, every `difficulty` ticks)",
                                       "bound": "instruction issue of the slowest wave (stragglers), not bytes",
                                       "us_per_launch": round(tk["replan"] * 1e6, 1), "us_max": round(tk["replan_max"] * 1e6, 1),
                                       "astar_expansions_mean": round(tk["astar_exp_mean"], 1), "astar_expansions_max": tk["astar_exp_max"]}
        if rank == 0 and not args.no_kernel_probes:
            res["roofline_compute_kernels"] = measure_compute_kernels(tr, cfg)
    del tr
    gc.collect()
    torch.cuda.empty_cache()
    return res


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4"])
    ap.add_argument("--num-envs", type=int, default=None, help="environments per GPU (default: the config's)")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--update-group", default=None, help="runtime.update_group: mini-batches per autograd graph of the update (default: auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-probes", action="store_true", help="skip the isolated kernel timings (roofline_compute_kernels): profiling runs")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg3 measurement that rides along with cfg2")
    ap.add_argument("--tick-samples", type=int, default=150)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank owns the config's environments; strong: --total-envs (default: the config's count) divided over the ranks")
    ap.add_argument("--total-envs", type=int, default=None, help="--scaling strong: environments of the whole job")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    world_env = int(os.environ.get("WORLD_SIZE", 1))
    if launched and world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} disagrees with WORLD_SIZE={world_env} of the launcher")
    if not launched and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    # The CPU baseline runs FIRST, while this process has not touched the GPU: its one-process-per-core workers are started from a
    # clean parent (the GPU box allows only a few processes with the device open; a worker never opens it -- tools/probe_gpu_open.py).
    cpu = None
    if not args.no_cpu_baseline and world_env == 1:
        T_cpu = min(args.max_steps or 150, 150)
        cpu = cpu_baseline(args.config, T_cpu)

    import torch.distributed as dist
    from distributed_multi_agent_reinforcement_learning_amd.trainer import dist_env
    rank, local_rank, world = dist_env()
    main_res = run_config(args.config, args, with_roofline=True)
    second = fp32_run = None
    if args.config == "cfg2" and not args.no_secondary:
        second = run_config("cfg3", args, with_roofline=False)
        # the same headline configuration with every matrix product on the fp32 MFMA kernels / the BLAS library (runtime.matmul: fp32),
        # for comparison: the default evaluates fp32 products from exact three-way bf16 operand splits
        from distributed_multi_agent_reinforcement_learning_amd import ops as _ops
        mode = _ops.MATMUL_MODE
        if mode != "fp32":
            _ops.set_matmul_mode("fp32")
            try:
                fp32_run = run_config("cfg2", args, with_roofline=False)
            finally:
                _ops.set_matmul_mode(mode)
    if rank == 0:
        out = {
            "metric": "env-steps/sec (whole node), pursuit-evasion 8-agent 4096-env", "value": main_res["value"],
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("f64 environment / f32 policy (f32 matrix products evaluated from exact 3-way bf16 operand splits, f32 accumulation and storage)"
                      if main_res["matmul"] == "split_bf16" else "f64 environment / f32 policy"),
            "data": "synthetic (seeded random maps, random-init weights)",
            "matmul": ("fp32 products from exact three-way bf16 splits of the fp32 operands (6 bf16 MFMAs per product, fp32 accumulation, fp32 storage; error "
                       "against f64 = an fp32 GEMM's, tests/test_ops_gpu.py); configs.cfg2_fp32_mfma is the same run on fp32 MFMA / BLAS"
                       if main_res["matmul"] == "split_bf16" else "fp32 MFMA kernels / BLAS library fp32 GEMMs"),
            "config": {"workload": main_res["workload"], "envs_per_gpu": main_res["envs_per_gpu"], "episode_steps": main_res["episode_steps"],
                       "mini_batch_size": main_res["mini_batch_size"], "update_group": main_res["update_group"],
                       "parallelism": f"dp{world} ({args.scaling} scaling: " + (f"{main_res['envs_per_gpu'] * world} environments in total, {main_res['envs_per_gpu']} per rank)"
                                                                                 if args.scaling == "strong" else f"{main_res['envs_per_gpu']} environments on every rank)"),
                       "collective": (f"torch.distributed {main_res['backend']} all_reduce(SUM) of one flat fp32 gradient bucket per epoch"
                                      if world > 1 else None)},
            "ppo_updates_per_s": main_res["ppo_updates_per_s"], "breakdown_ms": main_res["breakdown_ms"], "hbm_peak_GB": main_res["hbm_peak_GB"],
            "roofline": main_res["roofline"], "roofline_packed_rows": main_res["roofline_packed_rows"],
            "roofline_iteration": main_res["roofline_iteration"], "roofline_replan_tick": main_res["roofline_replan_tick"],
            "roofline_compute_kernels": main_res.get("roofline_compute_kernels", {}),
        }
        dom = main_res.get("roofline_compute_kernels", {}).get("gru_seq_bwd_grouped")
        if dom is not None:   # where the iteration's largest launch stands, without opening profiles/
            out["dominant_kernel"] = {"kernel": dom["kernel"], "us_per_launch": dom["us_per_launch"], "launches_per_iteration": int(2 * main_res.get("epochs", 1)),
                                      "bound": "hbm (5 120 B per sequence row and step) beside the matrix pipe", "frac_of_hbm": dom["frac_of_hbm"],
                                      "frac_of_matrix_pipe": dom["frac"], "matrix_pipe_peak_TFLOPs": dom["peak"]}
            try:              # HBM bytes per launch from the committed PMC record of this kernel (tools/pmc_gru_traffic.py), if its shape is the probe's
                rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_gru_traffic.json")))
                k = [r for r in rec["kernels"] if r["kernel"] == "k_gru_seq_bwd_sb"]
                if k and abs(k[0]["algorithmic_bytes"] - 5120.0 * dom.get("rows", 0) * dom.get("steps", 0)) < 1:
                    out["dominant_kernel"]["traffic"] = k[0]["fetch_bytes_x2"] + k[0]["write_bytes"]
                    out["dominant_kernel"]["algorithmic_bytes"] = k[0]["algorithmic_bytes"]
            except (OSError, ValueError, KeyError):
                pass
        if second is not None:
            out["configs"] = {"cfg3": {k: second[k] for k in ("value", "ms_per_step", "ppo_updates_per_s", "breakdown_ms", "hbm_peak_GB", "workload", "roofline_iteration")}}
        if fp32_run is not None:
            out.setdefault("configs", {})["cfg2_fp32_mfma"] = {k: fp32_run[k] for k in ("value", "ms_per_step", "ppo_updates_per_s", "breakdown_ms", "matmul")}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
