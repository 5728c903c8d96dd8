"""MAPPO agent: batched on-device rollout, GAE, PPO-clip / value-clip update -- the reference's `MAPPO` API.

Mirrors (reference paths) DHGN/mappo_parallel.py:548-831 `MAPPO` and DHGN/replay_buffer.py:5-99.  Differences that
are design, not behaviour: rollouts of N environments advance in lockstep on the GPU (env tick = one fused HIP
launch, policy = batched torch + fused HIP ops), transitions are written straight into device-resident
(N, T, ...) buffers, the static obstacle list is stored once per episode instead of T times.  The reference's
quirks that change numbers are reproduced when `runtime.reference_quirks` is true (default): the history list
shared by actor and critic during rollouts and the critic's all-ones adjacency over padded obstacle
slots in training only (Q5).
"""
import contextlib
import gc
import logging
import threading
import types

import numpy as np
import torch

from . import ops
from .model import build_actor_critic, pair_embeddings, pair_heads, sequence_forward_pair
from .pe_env import status_or, status_text

_log = logging.getLogger(__name__)

BUFFER_KEYS = ("p_state", "e_state", "o_state", "p_adj", "e_adj", "o_adj", "actor_historical_embedding",
               "critic_historical_embedding", "v_n", "a_n", "a_logprob_n", "r", "active")


class _Buffer(dict):
    """The reference's buffer dict.  `o_adj` (N, T, P, O) fp32 -- 3.5 GB at 4096 environments, 1/29 of it information -- is
    STORED bit-packed as `o_adj_bits` (N, T, P, RW) int32, the form the env kernel emits and the msg-agg kernels read; the
    reference-layout key is materialised on first access (tests, tools, a user's own code), never on the hot path."""

    def __missing__(self, key):
        if key == "o_adj" and "o_adj_bits" in self:
            v = ops.unpack_adj_bits(self["o_adj_bits"], self.num_obstacle_slots)
            self[key] = v
            return v
        raise KeyError(key)

    def stored_items(self):
        return [(k, v) for k, v in self.items() if k not in ("o_adj", "o_state")]


class ReplayBuffer:
    """Zero-padded episode buffer with the reference's keys and (N, T, ...) layouts (DHGN/replay_buffer.py:24-40),
    resident in HBM.  `o_state` is stored once per episode as (N, O, 4) (`o_static`) and exposed in the reference's
    (N, T, O, 4) shape as a broadcast view.
    Deviation at `algo.depth: 0` only: `actor_historical_embedding` / `critic_historical_embedding` are allocated with the
    reference's shape but left ZERO by the rollout unless `runtime.record_unused_embeddings` is true -- the update reads them as
    FCRA history only, which does not exist at depth 0 (the reference records them regardless and never reads them there)."""

    def __init__(self, cfg, num_rows=None, device=None):
        self.episode_limit = cfg.env.max_steps
        self.batch_size = int(num_rows if num_rows is not None else cfg.algo.sample_epi_num)
        self.device = torch.device(device if device is not None else cfg.algo.worker_device)
        self.max_p_num, self.max_e_num, self.max_o_num = cfg.env.num_defender, cfg.env.num_attacker, cfg.map.num_max_obstacle
        self.p_dim = self.e_dim = self.o_dim = cfg.env.state_dim
        self.embedding_size = cfg.algo.embedding_dim
        self.depth = cfg.algo.depth
        self.buffer = None

    def reset_buffer(self):
        N, T, P, O, E, d = self.batch_size, self.episode_limit, self.max_p_num, self.max_o_num, self.embedding_size, self.depth
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.device)
        b = _Buffer()
        b.num_obstacle_slots = O
        b["p_state"] = z(N, T, P, self.p_dim)
        b["e_state"] = z(N, T, self.max_e_num, self.e_dim)
        b["p_adj"] = z(N, T, P, P)
        b["e_adj"] = z(N, T, P, self.max_e_num)
        b["o_adj_bits"] = torch.zeros((N, T, P, ops.adj_row_words(O)), dtype=torch.int32, device=self.device)
        b["actor_historical_embedding"] = z(N, T + d, P, E)
        b["critic_historical_embedding"] = z(N, T + d, P, E)
        b["v_n"] = z(N, T + 1, P)
        b["a_n"] = z(N, T, P)
        b["a_logprob_n"] = z(N, T, P)
        b["r"] = z(N, T, P)
        b["active"] = z(N, T, P)
        self.o_static = z(N, O, self.o_dim)
        self.o_kvalid = torch.zeros(N, dtype=torch.int32, device=self.device)
        b["o_state"] = self.o_static[:, None].expand(N, T, O, self.o_dim)
        self.buffer = b
        return self

    @classmethod
    def from_tensors(cls, cfg, tensors, o_kvalid=None, device=None):
        """Builds a buffer from reference-layout tensors (e.g. golden fixtures); o_state (N,T,O,4) is static over T."""
        N = tensors["r"].shape[0]
        rb = cls(cfg, N, device).reset_buffer()
        for k in BUFFER_KEYS:
            if k == "o_state":
                rb.o_static.copy_(tensors[k][:, 0])
            elif k == "o_adj":
                rb.buffer["o_adj_bits"].copy_(ops.pack_adj_bits(tensors[k].to(rb.device)))
            else:
                rb.buffer[k].copy_(tensors[k])
        if o_kvalid is not None:
            rb.o_kvalid.copy_(torch.as_tensor(o_kvalid, dtype=torch.int32))
        return rb


class BigBuffer:
    """Learner-side concatenation of worker buffers (DHGN/replay_buffer.py:68-99)."""

    def __init__(self):
        self.buffer = None
        self.o_static = None

    def reset(self):
        self.buffer = None
        self.o_static = None

    def concat_buffer(self, mini_buffer):
        if self.buffer is None:
            self.buffer = _Buffer(mini_buffer.buffer.stored_items())
            self.buffer.num_obstacle_slots = mini_buffer.buffer.num_obstacle_slots
            self.buffer["o_state"] = mini_buffer.buffer["o_state"]
            self.o_static = mini_buffer.o_static
        else:
            self.o_static = torch.cat([self.o_static, mini_buffer.o_static], dim=0)
            self.buffer.pop("o_adj", None)  # a materialised view of the old rows
            for key, _ in self.buffer.stored_items():
                self.buffer[key] = torch.cat([self.buffer[key], mini_buffer.buffer[key]], dim=0)
            N, T = self.buffer["r"].shape[:2]
            self.buffer["o_state"] = self.o_static[:, None].expand(N, T, *self.o_static.shape[1:])

    def get_training_data(self, device):
        device = torch.device(device)
        if self.o_static.device != device:
            self.o_static = self.o_static.to(device)
            self.buffer.pop("o_adj", None)
            for key, _ in self.buffer.stored_items():
                self.buffer[key] = self.buffer[key].to(device)
            N, T = self.buffer["r"].shape[:2]
            self.buffer["o_state"] = self.o_static[:, None].expand(N, T, *self.o_static.shape[1:])
        return self.buffer


class _SegmentParams:
    """G sets of detached aliases of the trained parameters -- same storage, they ARE the weights -- each set with its own
    gradient storage.  The mini-batches of a group run forward and backward as ONE autograd graph (their GRU recurrences share
    launches, MAPPO._train_grouped) and still deliver their gradients separately, which the clip of the running sum after every
    mini-batch needs.  `use(k)` makes set k the modules' parameters for the duration of a forward."""

    def __init__(self, modules, params, G):
        self.params = list(params)
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(G, total, dtype=torch.float32, device=self.params[0].device)
        self.sets = [[p.detach().requires_grad_() for p in self.params] for _ in range(G)]
        self.attach()
        index = {id(p): i for i, p in enumerate(self.params)}
        self.owners, seen = [], set()
        for root in modules:
            for m in root.modules():
                for name, p in m._parameters.items():
                    if p is not None and id(p) in index and (id(m), name) not in seen:
                        seen.add((id(m), name))
                        self.owners.append((m, name, index[id(p)]))

    def attach(self):
        for k, aliases in enumerate(self.sets):
            o = 0
            for a in aliases:
                a.grad = self.flat[k, o:o + a.numel()].view_as(a)
                o += a.numel()

    def stale(self):
        """a parameter's storage was replaced since the aliases were made (module.to(), load of a whole module)"""
        return any(a.data_ptr() != p.data_ptr() for a, p in zip(self.sets[0], self.params))

    def zero(self):
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * 4
        if any(a.grad is None or not lo <= a.grad.data_ptr() < hi for aliases in self.sets for a in aliases):
            self.attach()
        self.flat.zero_()

    def grads(self, k):
        return [a.grad for a in self.sets[k]]

    @contextlib.contextmanager
    def use(self, k):
        aliases = self.sets[k]
        for m, name, i in self.owners:
            m._parameters[name] = aliases[i]
        try:
            yield
        finally:
            for m, name, i in self.owners:
                m._parameters[name] = self.params[i]


def _gru_weights(gru):
    """the torch.nn.GRU attributes ops.gru_multi reads, as they are bound right now (inside _SegmentParams.use: one set's aliases)"""
    names = [f"{kind}_l{layer}" for layer in range(gru.num_layers) for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    return types.SimpleNamespace(num_layers=gru.num_layers, **{n: getattr(gru, n) for n in names})


class MAPPO:
    def __init__(self, cfg, batch_size, mini_batch_size, agent_type):
        self.batch_size = batch_size
        self.mini_batch_size = mini_batch_size
        a = cfg.algo
        self.max_train_steps, self.lr, self.gamma, self.lamda = a.max_train_steps, a.lr, a.gamma, a.lamda
        self.epsilon, self.K_epochs, self.entropy_coef = a.epsilon, a.epochs, a.entropy_coef
        self.use_grad_clip, self.use_lr_decay = a.use_grad_clip, a.use_lr_decay
        self.use_adv_norm, self.use_value_clip = a.use_adv_norm, a.use_value_clip
        self.action_dim, self.input_dim = cfg.env.action_dim, cfg.env.state_dim
        self.num_layers, self.embedding_dim = a.num_layers, a.embedding_dim
        self.rnn_input_dim, self.rnn_hidden_dim = a.embedding_dim, a.rnn_hidden_dim
        self.sn = a.use_spectral_norm
        key = "learner_device" if "Learner" in agent_type else ("worker_device" if "Worker" in agent_type else "evaluator_device")
        self.device = torch.device(getattr(a, key))
        if self.device.type != "cuda":
            raise RuntimeError(f"algo.{key}={getattr(a, key)!r}: this MAPPO runs on the GPU only (HIP kernels, no CPU fallback)")
        self.depth = a.depth
        self.actor, self.critic = build_actor_critic(cfg, self.device)
        enc = self.actor.shared_net
        # same parameter order as the reference's ac_parameters (:631): Adam state / clip_grad_norm_ follow it
        # (algo.encoder=gnn_extractor: one encoder per network, the critic's follows the actor's)
        enc_params = list(enc.parameters()) + ([] if self.critic.shared_net is enc else list(self.critic.shared_net.parameters()))
        self.ac_parameters = (enc_params + list(self.actor.GRU.parameters()) + list(self.critic.GRU.parameters())
                              + list(self.critic.Mean.parameters()) + list(self.actor.Mean.parameters()))
        self.ac_optimizer = torch.optim.Adam(self.ac_parameters, lr=self.lr, eps=1e-5)
        self.minibuffer = None
        self.total_step = 0
        self.cfg = cfg
        rt = cfg.get("runtime", {})
        self.reference_quirks = bool(rt.get("reference_quirks", True))
        self.sample_seed = int(rt.get("seed", 0))
        # position of this agent's action-sampling stream: rank r starts its Philox counter at r << 40, so data-parallel
        # ranks (the reference's Workers are separate processes with independent torch generators, runner.py:81-86) draw
        # independent exploration noise; rank 0 keeps the stream the single-process goldens were made with.  Set by the
        # Trainer / Worker before the first rollout; the counter itself is part of the resume bundle.
        self.sample_rank = int(rt.get("sample_rank", 0))
        self.use_graphs = bool(rt.get("use_graphs", True))
        ug = rt.get("update_group", "auto")    # mini-batches per autograd graph in train(): "auto" | int (see _update_group)
        self.update_group = "auto" if str(ug) == "auto" else int(ug)
        if rt.get("matmul") is not None:   # "split_bf16" (default) | "fp32": how the fp32 matrix products are evaluated (ops.set_matmul_mode;
            ops.set_matmul_mode(str(rt.get("matmul")))   # process-wide, like torch's float32 matmul precision switch)
        self.update_group_max_GB = float(rt.get("update_group_max_GB", 200.0))
        # depth 0: nothing reads buffer["{actor,critic}_historical_embedding"], so the rollout does not fill them (they stay zero;
        # the reference stores every tick's embedding regardless, DHGN/mappo_parallel.py:795-798).  true = record them anyway.
        self.record_unused_embeddings = bool(rt.get("record_unused_embeddings", False))
        self.last_adv = self.last_v_target = None

    # ---- update (:638-723) ------------------------------------------------------------------------------------
    def train(self, replay_buffer, total_steps, return_grads=True):
        batch = replay_buffer.get_training_data(self.device) if hasattr(replay_buffer, "get_training_data") else replay_buffer.buffer
        o_static = replay_buffer.o_static
        N, T, P = batch["r"].shape
        if N != self.batch_size:
            raise ValueError(f"buffer holds {N} episodes, MAPPO was built for batch_size={self.batch_size}")
        with torch.no_grad():
            adv, v_target = ops.gae_advnorm(batch["r"], batch["v_n"], batch["active"], self.gamma, self.lamda, self.use_adv_norm)
        self.last_adv, self.last_v_target = adv, v_target
        object_critics = object_actors = 0.0
        update_time = 0
        if getattr(self, "grad_bucket", None) is not None:
            self.grad_bucket.zero()          # persistent flat gradient storage (trainer.GradBucket): zeroed, not dropped
        else:
            self.ac_optimizer.zero_grad()
        starts = list(range(0, N, self.mini_batch_size))  # BatchSampler(SequentialSampler, mini_batch_size, drop_last=False)
        G = self._update_group(len(starts), min(self.mini_batch_size, N) * P, T)
        while G > 1:
            try:
                object_critics, object_actors = self._train_grouped(batch, o_static, adv, v_target, starts, G)
                update_time = len(starts)
                starts = []
                break
            except torch.cuda.OutOfMemoryError:
                # the estimate was too low for this device's free memory: drop the half-built graph, start the epoch over with
                # half the group (the gradients accumulated so far are discarded; the mini-batch loop below is the G = 1 case)
                G = self._group_limit = G // 2
                _log.warning("update: out of device memory, retrying the epoch with %d mini-batches per autograd graph", G)
                self.last_update_group = G
                if getattr(self, "grad_bucket", None) is not None:
                    self.grad_bucket.zero()
                else:
                    self.ac_optimizer.zero_grad()
                torch.cuda.empty_cache()
        for n0 in starts:
            n1 = min(n0 + self.mini_batch_size, N)
            mb = n1 - n0
            obs, hist_a, hist_c = self._minibatch_inputs(batch, o_static, n0, n1)
            # One stream: running the critic branch beside the actor's on a second stream was tried in round 1 and removed --
            # it put two library GEMMs in flight at once, and with DHGN depth > 0 at 4096 environments the update stopped making
            # progress ("two-stream update").  A whole-device library GEMM is not a kernel to co-schedule.
            # actor and critic together: their GRU recurrences share one launch per layer and direction (model.sequence_forward_pair)
            prob, values_now = sequence_forward_pair(self.actor, self.critic, obs, hist_a, hist_c, mb, T)
            actor_loss, critic_loss = self._minibatch_losses(batch, adv, v_target, n0, n1, prob, values_now)
            (actor_loss + critic_loss).backward()
            if self.use_grad_clip:  # on the gradients accumulated so far, after every mini-batch
                torch.nn.utils.clip_grad_norm_(self.ac_parameters, 5.0)
            object_critics = object_critics + critic_loss.detach().double()   # f64 sum on the device: no host sync per mini-batch
            object_actors = object_actors + actor_loss.detach().double()
            update_time += 1
        object_critics, object_actors = float(object_critics), float(object_actors)
        if self.use_lr_decay:
            self.lr_decay(total_steps)
        if not return_grads:  # device-side path: gradients stay in .grad for the flat all-reduce
            return object_critics / update_time, object_actors / update_time, None, None
        return object_critics / update_time, object_actors / update_time, self.actor.get_gradients(), self.critic.get_gradients()

    def _minibatch_inputs(self, batch, o_static, n0, n1):
        """rows [n0, n1) of the buffer as one sequence batch: (obs dict, actor history slices, critic history slices)"""
        T, P, d = batch["r"].shape[1], batch["r"].shape[2], self.depth
        R = (n1 - n0) * T
        obs = dict(p_state=batch["p_state"][n0:n1].reshape(R, P, -1), e_state=batch["e_state"][n0:n1].reshape(R, 1, -1),
                   o_state=o_static[n0:n1], q_div=T, p_adj=batch["p_adj"][n0:n1].reshape(R, P, P),
                   e_adj=batch["e_adj"][n0:n1].reshape(R, P, 1), o_adj_bits=batch["o_adj_bits"][n0:n1].reshape(R, P, -1))
        # EmbeddingDataset2 (:95-113): hop k reads the stored embeddings of step t-1-k (clean per-net history)
        # (the (mb, T, P, E) slices are read in place by the neighbour-mean kernel: no gathered copies)
        hist_a = [batch["actor_historical_embedding"][n0:n1, d - 1 - k: d - 1 - k + T] for k in range(d)]
        hist_c = [batch["critic_historical_embedding"][n0:n1, d - 1 - k: d - 1 - k + T] for k in range(d)]
        return obs, hist_a, hist_c

    def _minibatch_losses(self, batch, adv, v_target, n0, n1, prob, values_now):
        v_old = batch["v_n"][n0:n1, :-1] if self.use_value_clip else None
        if ops.ppo_loss_prob_ok(prob, values_now.squeeze(-1)):
            # Categorical(prob).log_prob / .entropy() (get_logprob_and_entropy, :451-456) evaluated inside the loss launch
            return ops.ppo_loss_prob(prob, batch["a_n"][n0:n1], values_now.squeeze(-1), batch["a_logprob_n"][n0:n1], adv[n0:n1], batch["active"][n0:n1],
                                     v_old, v_target[n0:n1], self.epsilon, self.entropy_coef, self.use_value_clip)
        dist = torch.distributions.Categorical(prob)      # get_logprob_and_entropy (:451-456)
        a_logprob_n_now, dist_entropy = dist.log_prob(batch["a_n"][n0:n1]), dist.entropy()
        return ops.ppo_loss(a_logprob_n_now, dist_entropy, values_now.squeeze(-1), batch["a_logprob_n"][n0:n1], adv[n0:n1],
                            batch["active"][n0:n1], batch["v_n"][n0:n1, :-1] if self.use_value_clip else None,
                            v_target[n0:n1], self.epsilon, self.entropy_coef, self.use_value_clip)

    def _update_group(self, n_minibatches, rows, T):
        """how many consecutive mini-batches run as one autograd graph (`runtime.update_group`: an int, or "auto" = all of them:
        at a data-parallel rank's share of the batch ten launches of 26 workgroups each take as long as ten launches of 205 do,
        and at the full batch 10 x 2 x 205 workgroups in one launch fill the last round of the 256 CUs that 20 launches leave
        a fifth empty).  The activations of a group are alive together -- measured 25 KB per GRU row and step at depth 0, 35 KB
        at depth 3 (125 / 171 GB for the 4096-environment benchmark batch against 17.5 / 25 GB for the loop) --;
        `runtime.update_group_max_GB` (default 200 of the 288 GB) bounds the estimate."""
        if n_minibatches < 2 or not (self.actor.use_rnn and self.critic.use_rnn) or T < ops.PERSISTENT_GRU_MIN_T or self.device.type != "cuda":
            return 1
        per_minibatch = rows * T * (26e3 + 3.2e3 * self.depth)
        # the budget: the configured bound, and what the device can actually give right now -- free memory plus what the caching
        # allocator holds without using (another tenant, the background evaluator or a smaller part than the 288 GB one all show up here)
        free, _total = torch.cuda.mem_get_info(self.device)
        idle = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        budget = min(self.update_group_max_GB * 1e9, 0.85 * (free + idle))
        cap = min(n_minibatches, ops.GRU_MULTI_MAX_NETS // 2, max(1, int(budget / per_minibatch)))
        if getattr(self, "_group_limit", None):      # an earlier out-of-memory retry settled on a smaller group
            cap = min(cap, self._group_limit)
        G = cap if self.update_group == "auto" else max(1, min(int(self.update_group), cap))
        if G != getattr(self, "last_update_group", None):
            _log.info("update: %d mini-batches per autograd graph (%d mini-batches, estimate %.1f GB each, budget %.1f GB)", G, n_minibatches,
                      per_minibatch / 1e9, budget / 1e9)
            self.last_update_group = G
        return G

    def _train_grouped(self, batch, o_static, adv, v_target, starts, G):
        """The update's mini-batches in groups of G: the weights are the same for all of them (one optimiser step per update,
        after the last mini-batch), so a group's forward passes are independent and run as one autograd graph in which the GRU
        recurrences of ALL its mini-batches (both networks) share one persistent launch per layer and direction.  Each mini-batch
        differentiates its own aliases of the parameters (_SegmentParams), so its gradient arrives separately and the
        accumulate-then-clip sequence over mini-batches (Q9) is replayed afterwards in order: same numbers as the loop in
        train(), bit for bit."""
        N, T, P = batch["r"].shape
        segs = getattr(self, "_segs", None)
        if segs is None or len(segs.sets) < G or segs.stale():
            segs = self._segs = _SegmentParams((self.actor, self.critic), self.ac_parameters, G)
        main_grads = [p.grad for p in self.ac_parameters]
        if any(g is None for g in main_grads):
            for p in self.ac_parameters:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            main_grads = [p.grad for p in self.ac_parameters]
        object_critics = object_actors = 0.0
        H = self.actor.rnn_hidden_dim
        for g0 in range(0, len(starts), G):
            group = [(n0, min(n0 + self.mini_batch_size, N)) for n0 in starts[g0:g0 + G]]
            segs.zero()
            xs, mods, links = [], [], []
            for k, (n0, n1) in enumerate(group):
                obs, hist_a, hist_c = self._minibatch_inputs(batch, o_static, n0, n1)
                with segs.use(k):
                    emb_a, emb_c = pair_embeddings(self.actor, self.critic, obs, hist_a, hist_c)
                    mods += [_gru_weights(self.actor.GRU), _gru_weights(self.critic.GRU)]
                rows = (n1 - n0) * T * P
                xs += [emb_a.reshape(rows, self.actor.rnn_input_dim), emb_c.reshape(rows, self.critic.rnn_input_dim)]
                links += [getattr(emb_a, "relu_link", None), getattr(emb_c, "relu_link", None)]   # depth > 0: the last hop's ReLU
            h0s = [torch.zeros(m.num_layers, x.shape[0] // T, H, dtype=x.dtype, device=x.device) for x, m in zip(xs, mods)]
            feats = ops.gru_multi(xs, h0s, mods, agents=P, steps=T, grouped=True, x_links=links, zero_state=True)
            losses = []
            for k, (n0, n1) in enumerate(group):
                mb = n1 - n0
                with segs.use(k):
                    prob, values_now = pair_heads(self.actor, self.critic, feats[2 * k].reshape(T, mb, P, H), feats[2 * k + 1].reshape(T, mb, P, H))
                losses.append(self._minibatch_losses(batch, adv, v_target, n0, n1, prob, values_now))
            torch.autograd.backward([l for pair in losses for l in pair])
            for k, (actor_loss, critic_loss) in enumerate(losses):
                torch._foreach_add_(main_grads, segs.grads(k))
                if self.use_grad_clip:  # on the gradients accumulated so far, after every mini-batch
                    torch.nn.utils.clip_grad_norm_(self.ac_parameters, 5.0)
                object_critics = object_critics + critic_loss.detach().double()
                object_actors = object_actors + actor_loss.detach().double()
        return object_critics, object_actors

    def lr_decay(self, total_steps):
        lr_now = self.lr * (1 - total_steps / self.max_train_steps)
        for p in self.ac_optimizer.param_groups:
            p["lr"] = lr_now
        self.total_step = total_steps

    # ---- rollout (:731-827) -----------------------------------------------------------------------------------
    def explore_env(self, env, num_episode, actions_override=None, init=None):
        N = env.num_envs
        self.minibuffer = ReplayBuffer(self.cfg, N * num_episode, self.device).reset_buffer()
        exp_reward = 0.0
        sample_steps = 0
        for k in range(num_episode):
            ep_reward, ep_steps = self.run_episode(env, num_episode=k, actions_override=actions_override, init=init)
            # one read-back per episode: the mean return and the kernels' status bits (tape exhausted / A* cap / path
            # underflow silently break parity with the reference, so they are an error, not a statistic)
            mean_r, bits = torch.stack((ep_reward.mean().double(), status_or(env.sim.meta).double())).tolist()
            if bits:
                raise RuntimeError("environment kernel status: " + status_text(int(bits)))
            exp_reward += float(mean_r)
            sample_steps += ep_steps * N
        return exp_reward / num_episode, self.minibuffer, sample_steps

    @torch.no_grad()
    def run_episode(self, env, num_episode=0, actions_override=None, init=None):
        """N episodes in lockstep; rows [num_episode*N, (num_episode+1)*N) of the buffer.  Returns the per-environment
        episode reward (N,) and the episode length.  The per-tick policy program (actor + critic forward, history
        bookkeeping, sampling) runs on static device storage and is replayed as one captured hipGraph per tick
        (`runtime.use_graphs`), which removes ~70 launches of host overhead per tick; the eager path executes the very same
        program, so both produce identical buffers."""
        N, P, T, d = env.num_envs, env.num_defender, env.max_steps, self.depth
        buf = self.minibuffer.buffer
        rows = slice(num_episode * N, (num_episode + 1) * N)
        env.reset(init)
        st = self._rollout_state(env)
        st.reset(env)
        self.minibuffer.o_static[rows].copy_(st.o_state)
        self.minibuffer.o_kvalid[rows].copy_(st.o_kvalid)
        episode_reward = torch.zeros(N, device=self.device)
        raw = st.raw
        obs_keys = ("p_state", "e_state", "p_adj", "e_adj", "o_adj_bits")
        env.observe(st.obs)
        env.attacker_step()
        use_graph = self.use_graphs and actions_override is None
        for t in range(T):
            items = [(st.obs[k], buf[k][rows, t]) for k in obs_keys]   # the policy program only reads st.obs
            if actions_override is not None:
                st.step(actions_override[rows, t].to(self.device).long())
            elif use_graph:
                st.replay_policy_step()
            else:
                st.step()
            items += [(st.v, buf["v_n"][rows, t]), (st.a_n, buf["a_n"][rows, t]),      # int32 -> float32 like the reference buffer
                      (st.logp, buf["a_logprob_n"][rows, t])]
            if d or self.record_unused_embeddings:  # the update reads the stored embeddings as FCRA history only (EmbeddingDataset2, :95-113)
                items += [(st.a_cur, buf["actor_historical_embedding"][rows, t + d]), (st.c_cur, buf["critic_historical_embedding"][rows, t + d])]
            # one launch records the tick (and adds the previous tick's raw reward to the episode return)
            ops.rollout_record(items, raw if t > 0 else None, episode_reward)
            if t + 1 < T:
                env.tick(st.a_n, st.obs, buf["r"][rows, t], raw)
            else:
                env.sim.step(st.a_n, buf["r"][rows, t], raw)
                env.time_step += 1
        episode_reward += raw.sum(-1)
        buf["active"][rows].fill_(1.0)
        # bootstrap value of the state after the last step (:807-825): only the critic's embedding enters the history
        env.observe(st.obs)
        st.value_step()
        buf["v_n"][rows, T].copy_(st.v)
        return episode_reward, T

    def _rollout_state(self, env):
        st = getattr(self, "_rstate", None)
        if st is None or st.N != env.num_envs:
            st = _RolloutState(self, env)
            self._rstate = st
        return st

    def save_model(self, cwd):
        torch.save(self.actor.state_dict(), cwd + "actor.pth")
        torch.save(self.critic.state_dict(), cwd + "critic.pth")


_gc_lock, _gc_depth, _gc_was_on = threading.Lock(), 0, False


@contextlib.contextmanager
def _collector_paused():
    """The cyclic collector must not run while a stream captures: what it frees may be another agent's graphs or their private pool, and
    releasing those inside a capture aborts the process (torch.cuda.graph collects once on entry; an allocation-count threshold can still
    trip mid-capture).  Nested / concurrent (the trainer's and the background evaluator's thread): the collector comes back when the last
    capture has ended."""
    global _gc_depth, _gc_was_on
    with _gc_lock:
        if _gc_depth == 0:
            _gc_was_on = gc.isenabled()
            gc.disable()
        _gc_depth += 1
    try:
        yield
    finally:
        with _gc_lock:
            _gc_depth -= 1
            if _gc_depth == 0 and _gc_was_on:
                gc.enable()


class _RolloutState:
    """Static device storage of one lockstep rollout and the per-tick policy program on it.

    History (FCRA hops): the embeddings of the last ticks live in a RING of pair slots `ring[M][2]` (slot t % M holds tick t's actor
    [0] and critic [1] embedding, written there directly by the forward pass), and a hop is a reference to a slot -- nothing is
    shifted or copied per tick (rounds 1-2 copied three (N, P, E) tensors per tick).  The addresses a tick reads and writes depend
    on t % M only, so the captured per-tick program is one hipGraph per phase (M = 3 at depth 3 with the shared-history quirk).
    GRU state: two buffers per network, tick t reads [t % 2] and writes [(t + 1) % 2] -- the split-bf16 cell (ops.CELL_MODE) shares a
    row tile between two workgroups and cannot update the state in place; `ha` / `hc` are the current state.  The tick program
    therefore repeats with period lcm(M, 2)."""

    def __init__(self, agent, env):
        self.agent = agent
        N, P = env.num_envs, env.num_defender
        O = env.pe_cfg.O
        d, L, H, E = agent.depth, agent.num_layers, agent.rnn_hidden_dim, agent.embedding_dim
        dev = agent.device
        self.N, self.P, self.d = N, P, d
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.obs = dict(p_state=z(N, P, 4), e_state=z(N, 1, 4), p_adj=z(N, P, P), e_adj=z(N, P, 1),
                        o_adj_bits=torch.zeros((N, P, ops.adj_row_words(O)), dtype=torch.int32, device=dev))  # LiDAR rows, bit-packed
        self.o_state = z(N, O, 4)
        self.o_kvalid = torch.zeros(N, dtype=torch.int32, device=dev)
        self.hbuf_a, self.hbuf_c = z(2, L, N * P, H), z(2, L, N * P, H)
        # the shared history list is a quirk of the DHGN agent's rollout; separate encoders keep their own histories
        self.quirk = bool(agent.reference_quirks and agent.actor.shared_net is agent.critic.shared_net)
        # quirk: the list is (.., a_{t-1}, c_{t-1}) for BOTH networks -- two entries per tick, so d hops reach ceil(d / 2) ticks back;
        # clean: every network reads its own last d embeddings.  One more slot for the tick being written.
        self.M = ((d + 1) // 2 if self.quirk else d) + 1
        self.ring = z(self.M, 2, N, P, E)
        self.t = 0                                            # policy steps taken in this episode (host counter: selects the phase)
        self.a_n = torch.zeros((N, P), dtype=torch.int32, device=dev)
        self.logp, self.v, self.raw = z(N, P), z(N, P), z(N, P)
        self.counter = torch.full((1,), int(agent.sample_rank) << 40, dtype=torch.int64, device=dev)  # position in the sampling stream (persists)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)   # scratch of ops.head_sample (left zero by every launch)
        self.period = self.M if self.M % 2 == 0 else 2 * self.M   # of the tick program's addresses: ring slot and state parity
        self.graphs = {}                                      # phase (t % period) -> captured tick program
        # one encoder pass per tick for both networks (DHGN.forward_pair): they hold the same DHGN instance (:582-616)
        self.pair_forward = agent.actor.shared_net is agent.critic.shared_net
        ops.gemm_workspace(dev)                               # this thread's hipBLASLt scratch exists before any tick program is captured

    @property
    def ha(self):   # the actor's current GRU state (what tick self.t reads)
        return self.hbuf_a[self.t & 1]

    @property
    def hc(self):
        return self.hbuf_c[self.t & 1]

    # the embeddings of the tick most recently computed (what the rollout records into the buffer)
    @property
    def a_cur(self):
        return self.ring[(self.t - 1) % self.M][0]

    @property
    def c_cur(self):
        return self.ring[(self.t - 1) % self.M][1]

    @property
    def hist_c(self):   # kept for callers that ask whether the networks keep separate histories
        return None if self.quirk else self.ring[:, 1]

    def reset(self, env):
        for t in (self.hbuf_a, self.hbuf_c, self.ring):
            t.zero_()
        self.t = 0
        self.o_state.copy_(env.boundary_map.obstacle_agent)
        self.o_kvalid.copy_(env.n_obs)

    def _obs(self):
        o = dict(self.obs)
        o["o_state"], o["o_kvalid"] = self.o_state, self.o_kvalid
        return o

    def _hops(self, t):
        """(hops_a, hops_c) for the forward pass of tick t: hop k = the k-th most recent entry of the history list.  Slots of ticks
        before the episode's first are still zero (the reference starts from zero embeddings, :749-753)."""
        d, M, ring = self.d, self.M, self.ring
        if self.quirk:   # (.., a_{t-1}, c_{t-1}): hop 0 = c_{t-1}, hop 1 = a_{t-1}, hop 2 = c_{t-2}, ...
            hops = [ring[(t - 1 - j // 2) % M][1 - j % 2] for j in range(d)]
            return hops, hops
        return [ring[(t - 1 - k) % M][0] for k in range(d)], [ring[(t - 1 - k) % M][1] for k in range(d)]

    def policy_step(self, forced_actions=None):
        """the tick program of tick self.t (the caller advances self.t afterwards: step() / replay_policy_step())"""
        ag = self.agent
        hops_a, hops_c = self._hops(self.t)
        slot = self.ring[self.t % self.M]                    # this tick's embeddings land here
        a_cur, c_cur = slot[0], slot[1]
        o = self._obs()
        ha_cur, hc_cur = self.ha, self.hc
        ha_new, hc_new = self.hbuf_a[(self.t + 1) & 1], self.hbuf_c[(self.t + 1) & 1]
        # every result lands directly in the static rollout storage (no copies behind the model)
        if self.pair_forward and not torch.is_grad_enabled():
            # one encoder pass for both networks (they hold the same DHGN instance), then the two GRU trunks and heads
            emb = ag.actor.shared_net.forward_pair(o["p_state"], o["e_state"], o["o_state"], o["p_adj"], o["e_adj"], o["o_adj_bits"],
                                                   hops_a, hops_c, o["o_kvalid"], 1, slot)
            a_emb, c_emb = emb[0], emb[1]
            if ag.actor.use_rnn and ag.critic.use_rnn:
                # the two GRU trunks layer by layer, actor's and critic's cell in one launch; the new states land in the other buffers
                E = a_emb.shape[-1]
                fa, fc = ops.gru_step_multi([a_emb.reshape(-1, E), c_emb.reshape(-1, E)], [ha_cur, hc_cur], [ag.actor.GRU, ag.critic.GRU],
                                            hiddens_out=[ha_new, hc_new])
                feat_a, feat_c = fa.reshape(self.N, self.P, -1), fc.reshape(self.N, self.P, -1)
                ha, hc = ha_new, hc_new
            else:
                feat_a, ha = ag.actor._rollout_features(a_emb, ha_cur, True)
                feat_c, hc = ag.critic._rollout_features(c_emb, hc_cur, True)
            v = ag.critic.head(feat_c, out=self.v)            # the value lands in the static storage
            w_a = ag.actor.head_weight() if forced_actions is None else None
            feat_a = feat_a.contiguous()
            if w_a is not None and ops._head_ok(feat_a, w_a, ag.actor.Mean.bias):
                # action head, softmax, sample, log-probability and the stream counter in one launch
                ops.head_sample(feat_a, w_a, ag.actor.Mean.bias, ag.sample_seed, self.counter, self.ticket, (self.a_n, self.logp))
                prob = None
            else:
                prob = torch.softmax(ag.actor.head(feat_a), dim=-1)
        else:
            prob, ha, a_emb = ag.actor(o, hops_a, ha_cur, 0, inplace_hidden=True, emb_out=a_cur)
            v, hc, c_emb = ag.critic(o, hops_c, hc_cur, 0, rollout=True, inplace_hidden=True, emb_out=c_cur)
        if forced_actions is not None:
            self.a_n.copy_(forced_actions.to(torch.int32))
            self.logp.copy_(torch.distributions.Categorical(probs=prob).log_prob(forced_actions))
        elif prob is not None:
            ops.categorical_sample(prob, ag.sample_seed, 0, counter=self.counter, out=(self.a_n, self.logp))
        if ha is not ha_new:       # paths that return a fresh state or updated the current one in place
            ha_new.copy_(ha)
        if hc is not hc_new:
            hc_new.copy_(hc)
        if a_emb.data_ptr() != a_cur.data_ptr():
            a_cur.copy_(a_emb)
        if c_emb.data_ptr() != c_cur.data_ptr():
            c_cur.copy_(c_emb)
        if v.data_ptr() != self.v.data_ptr():
            self.v.copy_(v.reshape(self.N, self.P))

    def step(self, forced_actions=None):
        """eager tick"""
        self.policy_step(forced_actions)
        self.t += 1

    def value_step(self):
        """bootstrap value after the last step (:807-825): only the critic's embedding of the last tick enters the history"""
        ag, d, t, M = self.agent, self.d, self.t, self.M
        hops = []
        if d:
            if self.quirk:   # [c_{t-1}] + the list as it stood at the last tick, (c_{t-2}, a_{t-2}, c_{t-3}, ..)
                hops = ([self.ring[(t - 1) % M][1]] + self._hops(t - 1)[1])[:d]
            else:
                hops = self._hops(t)[1]
        v, hc, c_emb = ag.critic(self._obs(), hops, self.hc, 0, rollout=True)
        self.v.copy_(v.reshape(self.N, self.P))

    def replay_policy_step(self):
        phase = self.t % self.period
        g = self.graphs.get(phase)
        if g is None:
            g = self.graphs[phase] = self._capture()
        g.replay()
        self.t += 1

    def _capture(self):
        """Records the tick program of the current phase once.  The warm-up runs mutate the rollout state, so it is saved and
        restored."""
        live = (self.hbuf_a, self.hbuf_c, self.ring, self.counter, self.a_n, self.logp, self.v)
        keep = [t.clone() for t in live]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.policy_step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with _collector_paused():
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # a background evaluator may launch on its own stream meanwhile
                self.policy_step()
        for t, k in zip(live, keep):
            t.copy_(k)
        return g
