"""DHGN encoder + GRU actor / critic with the reference's parameter names and initialisation stream.

Mirrors (reference paths) DHGN/mappo_parallel.py:116-387 `DHGN`, :390-474 `SharedActor`, :477-545 `SharedCritic`.
* state_dict keys, shapes and `.parameters()` order equal the reference's (checkpoint / gradient-list contract,
  SURVEY 8a M1-M5); modules are constructed in the reference's order with the same torch initialisers, so the same
  `torch.manual_seed` yields bit-identical initial weights.
* forward is written on whole tensors (rows = environments or environment-steps) instead of the reference's
  DataLoader(batch_size=1) plumbing, and the relation message + mean aggregation runs in the fused HIP op
  `ops.msg_agg` (csrc/mappo_ops.hip).  Only the working aggregator ('mean') exists.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm
from torch.nn.utils.spectral_norm import SpectralNorm

from . import ops


class HeadLinear(nn.Linear):
    """nn.Linear (same parameters, initialisation and state_dict) for the action / value heads: a handful of outputs, so under
    autograd the weight and bias gradients take the streaming kernel (ops.linear_skinny) instead of 128 x {1, 9} x rows GEMMs."""

    def forward(self, x):
        return ops.linear_skinny(x, self.weight, self.bias)


def _ortho_linear(n_in, n_out, cls=nn.Linear):
    """reference preproc_layer without spectral norm (DHGN/mappo_parallel.py:19-31): default Linear init is drawn
    first (it consumes the generator), then weights are re-drawn orthogonal (gain 1) and the bias zeroed."""
    layer = cls(n_in, n_out)
    for name, param in layer.named_parameters():
        if "bias" in name:
            nn.init.constant_(param, 0)
        elif "weight" in name:
            nn.init.orthogonal_(param, gain=1.0)
    return layer


def _make_linear(n_in, n_out, is_sn, cls=nn.Linear):
    # `preproc_layer(a, b) if is_sn else nn.Linear(a, b)` -- note is_sn is NOT forwarded
    return _ortho_linear(n_in, n_out, cls) if is_sn else cls(n_in, n_out)


class DHGN(nn.Module):
    def __init__(self, input_dim, embedding_dim, is_sn, algo_config, device=None):
        super().__init__()
        for key in ("vertex_level_aggregator", "fcra_aggregator"):
            if getattr(algo_config, key) != "mean":
                raise NotImplementedError(f"{key}={getattr(algo_config, key)!r}: only 'mean' works in the reference too")
        self.ReLU = nn.ReLU()
        self.MSG_layers = nn.ModuleList()
        self.AGG_layers = nn.ModuleDict()
        self.FCRA_layers = nn.ModuleList()
        self.alpha = nn.ModuleDict()
        self.depth = int(algo_config.depth)
        self.num_relation = int(algo_config.num_relation)
        if self.num_relation != 3:
            raise NotImplementedError("num_relation must be 3 (defender, evader, obstacle)")
        self.input_dim, self.embedding_dim = input_dim, embedding_dim
        E = embedding_dim
        self.semantic_layer = _make_linear(3 * E + input_dim, E, is_sn)
        for r in range(self.num_relation):
            self.MSG_layers.append(_make_linear(2 * input_dim if r == 0 else input_dim, E, is_sn))
        for _ in range(self.depth):
            self.FCRA_layers.append(_make_linear(2 * E, E, is_sn))
        self.AGG_layers["AGG_vertex_0"] = _make_linear(E, E, is_sn)  # shared by the three relations (:151-152, :278)
        for k in range(self.depth):
            self.AGG_layers[f"AGG_fcra_{k}"] = _make_linear(E, E, is_sn)

    # -- encoder (:241-304) ----------------------------------------------------------------------------
    def encoder(self, p, e, o, adj_p, adj_e, adj_o, is_critic, o_kvalid=None, q_div=1, out=None):
        """p (R,P,4), e (R,1,4), o (R/q_div,O,4), adj_* (R,P,{P,1,O}) (adj_o may be bit-packed int32 rows) -> h0 (R,P,E).
        is_critic: adjacency := ones (AttributeDataset, :64-65); in a batched rollout the obstacle relation uses ones
        over the first o_kvalid[row] (real) obstacles, in training over all padded slots."""
        M = self.MSG_layers
        m3 = ops.msg_agg3(p, e, o, adj_p, adj_e, adj_o, M[0].weight, M[0].bias, M[1].weight, M[1].bias, M[2].weight, M[2].bias,
                          is_critic, o_kvalid, q_div)                                  # (R, P, 3, E)
        return self._after_messages(p, m3, out)

    def _after_messages(self, p, m3, out=None):
        """AGG_vertex_0 on the three relation aggregates, then the semantic layer (:278-303)"""
        R, P = p.shape[0], p.shape[1]
        E, ind = self.embedding_dim, self.input_dim
        agg0 = self.AGG_layers["AGG_vertex_0"]
        link = ops.ReluLink()    # emb feeds the semantic layer only: its relu' and bias gradient ride in that layer's input gradient
        emb = ops.linear(m3, agg0.weight, agg0.bias, relu=True, y_link=link)          # one GEMM (+relu epilogue) for the three relations
        # semantic_layer([p, emb0, emb1, emb2]) without materialising the concatenation (:284-303)
        Ws = self.semantic_layer.weight
        # the position part (K = 4) first, the embedding part accumulates INTO it (beta = 1, no copy of the addend)
        p2, e2 = p.reshape(R * P, ind), emb.reshape(R * P, 3 * E)
        if out is not None:  # rollout: both GEMMs write the static storage
            o2 = out.view(R * P, E)
            torch.addmm(self.semantic_layer.bias, p2, Ws[:, :ind].t(), out=o2)
            h0 = o2.addmm_(e2, Ws[:, ind:].t())
        else:
            h0 = ops.linear(e2, Ws[:, ind:], ops.linear_skinny(p2, Ws[:, :ind], self.semantic_layer.bias), consume_addend=True, x_link=link)
        return h0.reshape(R, P, E)

    def encoder_pair_train(self, p, e, o, adj_p, adj_e, adj_o, q_div=1):
        """the update's encoder of actor and critic together -> (h0_actor, h0_critic): one message pass for both networks
        (ops.msg_agg3_pair_train), then the layers per network; same numbers as encoder(.., False) and encoder(.., True)."""
        M = self.MSG_layers
        m3a, m3c = ops.msg_agg3_pair_train(p, e, o, adj_p, adj_e, adj_o, M[0].weight, M[0].bias, M[1].weight, M[1].bias, M[2].weight, M[2].bias,
                                           q_div)
        return self._after_messages(p, m3a), self._after_messages(p, m3c)

    # -- fixed-depth recursive aggregation over neighbours' historical embeddings (:204-233) --------------
    def fcra(self, h0, hist, adj_p, is_critic, out=None):
        """hist: sequence of `depth` tensors (R,P,E), hop k = hist[k] (k = 0 is the most recent)."""
        if self.depth == 0:
            return h0
        h, carry = h0, None
        for k in range(self.depth):
            aggk, fk = self.AGG_layers[f"AGG_fcra_{k}"], self.FCRA_layers[k]
            # matmul(normalize(adj or ones, p=1), hist[k]): one pass, the history slice read in place (hist is stored data)
            nb = ops.fcra_mean(z_critic=hist[k]) if is_critic else ops.fcra_mean(z_actor=hist[k], adj=adj_p)
            last = k == self.depth - 1
            # FCRA_k([relu(AGG_k(nb)) | h]) as one K = 2E GEMM whose operand halves are written in place by their producers
            # (ops.fcra_hop); the last hop of a rollout lands in the static storage
            # (each hop's output feeds the next hop only: `carry` hands its operand buffer and its relu' to that hop)
            h, carry = ops.fcra_hop(nb, h, carry, aggk.weight, aggk.bias, fk.weight, fk.bias, last, out if last else None)
        h.relu_link = carry[1]   # for a caller that is the ONLY consumer of h (the update hands it to the first GRU layer)
        return h

    def forward(self, p, e, o, adj_p, adj_e, adj_o, hist, is_critic, o_kvalid=None, q_div=1, out=None):
        """out (R,P,E), rollout only: the embedding is written into it (static storage of the captured tick program)."""
        h0 = self.encoder(p, e, o, adj_p, adj_e, adj_o, is_critic, o_kvalid, q_div, out if self.depth == 0 else None)
        return self.fcra(h0, hist, adj_p, is_critic, out)

    # -- actor and critic of one rollout tick together ---------------------------------------------------------------------
    # The critic is built on the actor's DHGN instance (DHGN/mappo_parallel.py:582-616), so in a rollout the two embeddings
    # are the same layers applied to the same observation with two adjacencies (the observed one / ones).  forward_pair runs
    # them as one batch of 2R rows: one message pass, one GEMM per layer instead of two, the position part of the semantic
    # layer once.  Slot 0 = actor, slot 1 = critic; the numbers are those of forward(.., False) and forward(.., True).
    def forward_pair(self, p, e, o, adj_p, adj_e, adj_o, hist_a, hist_c, o_kvalid=None, q_div=1, out=None):
        """-> (2, R, P, E); out: static storage of that shape (rollout).  No autograd."""
        R, P = p.shape[0], p.shape[1]
        E, ind = self.embedding_dim, self.input_dim
        M = self.MSG_layers
        Ws = self.semantic_layer.weight
        d = self.depth
        # [agg | h] operands of the FCRA layers, one per hop: each layer is ONE K = 2E GEMM (bias + ReLU in its epilogue) whose
        # halves are written in place by their producers -- the neighbour-mean kernel (left) and the previous layer / the
        # semantic layer (right) -- so neither a concatenation nor a separate ReLU pass exists
        cats = [torch.empty((2, R, P, 2 * E), dtype=p.dtype, device=p.device) for _ in range(d)]
        if d:
            h0 = cats[0][..., E:]
        else:
            h0 = out if out is not None else torch.empty((2, R, P, E), dtype=p.dtype, device=p.device)
        h0_2d = ops.block2d(h0)
        fused_pos = ind == 4
        # the message launch also leaves the semantic layer's position part (the same for both networks) in h0
        m3 = ops.msg_agg3_pair(p, e, o, adj_p, adj_e, adj_o, M[0].weight, M[0].bias, M[1].weight, M[1].bias, M[2].weight, M[2].bias,
                               o_kvalid, q_div, pos=(Ws[:, :ind], self.semantic_layer.bias, h0) if fused_pos else None)   # (2, R, P, 3, E)
        agg0 = self.AGG_layers["AGG_vertex_0"]
        m3_2d = m3.view(-1, E)
        if ops.split_linear_ok(m3_2d, agg0.weight):     # the rollout's Linear layers on the split-bf16 kernel (ops.CELL_MODE)
            emb = ops.split_linear(m3_2d, agg0.weight, agg0.bias, relu=True).view(m3.shape)
        else:
            emb = ops.linear(m3, agg0.weight, agg0.bias, relu=True)
        if not fused_pos:
            torch.addmm(self.semantic_layer.bias, p.reshape(R * P, ind), Ws[:, :ind].t(), out=h0_2d[:R * P])
            h0_2d[R * P:].copy_(h0_2d[:R * P])
        emb_2d, Wse = emb.view(2 * R * P, 3 * E), Ws[:, ind:]
        if ops.split_linear_ok(emb_2d, Wse, h0_2d, h0_2d):
            ops.split_linear(emb_2d, Wse, None, False, out=h0_2d, addend=h0_2d)
        elif d:   # h0 is the right half of the first hop's operand: accumulate into it in place (beta = 1, strided output)
            ops.gemm_nt(emb_2d, Wse, None, False, out=h0_2d, addend=h0_2d)
        else:
            h0_2d.addmm_(emb_2d, Wse.t())
        if d == 0:
            return h0
        h = h0
        # relu((abar @ hist) W^T + b) evaluated as relu(abar @ (hist W^T) + b): in the reference's rollout both networks read the same
        # history list, so the GEMM runs once for the two of them.  Every hop's aggregate depends on stored history only,
        # so the d neighbour means (both networks each, bias and ReLU included) are ONE launch that writes the left halves of all
        # hops' operands before the chain of FCRA layers runs
        zas, zcs = [], []
        for k in range(d):
            aggk = self.AGG_layers[f"AGG_fcra_{k}"]

            def lin(z, W=aggk.weight):
                if z.is_contiguous() and ops.split_linear_ok(z.view(-1, E), W):
                    return ops.split_linear(z.view(-1, E), W).view(z.shape)
                return ops.linear(z, W)
            za = lin(hist_a[k])
            zcs.append(za if hist_c[k] is hist_a[k] or hist_c[k].data_ptr() == hist_a[k].data_ptr() else lin(hist_c[k]))
            zas.append(za)
        if d <= 4 and all(z.is_contiguous() for z in zas + zcs) and adj_p.dtype == torch.float32:
            ops.fcra_mean_pair_multi(zas, zcs, adj_p, [self.AGG_layers[f"AGG_fcra_{k}"].bias for k in range(d)], [cats[k][..., :E] for k in range(d)])
        else:
            for k in range(d):
                ops.fcra_mean(z_actor=zas[k], z_critic=zcs[k], adj=adj_p, bias=self.AGG_layers[f"AGG_fcra_{k}"].bias, relu=True, out=cats[k][..., :E])
        for k in range(d):
            fk = self.FCRA_layers[k]
            last = k == d - 1
            if not last:
                h = cats[k + 1][..., E:]
            else:
                h = out if out is not None else torch.empty((2, R, P, E), dtype=p.dtype, device=p.device)
            cat_2d, h_2d = cats[k].view(2 * R * P, 2 * E), ops.block2d(h)
            if ops.split_linear_ok(cat_2d, fk.weight, h_2d):
                ops.split_linear(cat_2d, fk.weight, fk.bias, True, out=h_2d)
            else:
                ops.gemm_nt(cat_2d, fk.weight, fk.bias, True, out=h_2d)
        return h


def _o_adj(obs):
    """the obstacle adjacency of an observation dict: bit-packed rows (`o_adj_bits`, int32; what the env kernel emits and the
    replay buffer stores) when present, else the reference-layout float rows `o_adj`"""
    return obs["o_adj_bits"] if "o_adj_bits" in obs else obs["o_adj"]


class _Trunk(nn.Module):
    """shared_net -> GRU -> Mean, common to actor and critic.  use_rnn = False is the "MLP" ablation BASELINE config 1 names
    (the reference parses --use_rnn, MAPPO_parallel_main.py:261, and never reads it): encoder -> head, the GRU (whose
    parameters stay in the module, so checkpoints keep their keys) is bypassed and the hidden state passes through untouched."""
    use_rnn = True

    def _rollout_features(self, embedding, hidden_state, inplace_hidden=False):
        R, P, E = embedding.shape
        if not self.use_rnn:
            return embedding, hidden_state
        feat, hidden_state = ops.gru(embedding.reshape(1, R * P, E), hidden_state, self.GRU, inplace_hidden)
        return feat.reshape(R, P, self.rnn_hidden_dim), hidden_state

    def _sequence_features(self, embedding, batch, steps):
        """embedding: (batch steps, P, E) rows in (episode, step) order -> GRU features TIME-MAJOR (steps, batch, P, H).  The
        reference permutes the embedding to (steps, batch P, E) and the features back (:426-437); here the first GRU layer
        reads the rows where they lie and the (small) head outputs are permuted instead of the features."""
        P = embedding.shape[1]
        if not self.use_rnn:
            return embedding.reshape(batch, steps, P, self.rnn_input_dim).permute(1, 0, 2, 3)
        h0 = torch.zeros(self.num_layers, batch * P, self.rnn_hidden_dim, dtype=embedding.dtype, device=embedding.device)
        feat, _ = ops.gru(embedding.reshape(batch * steps * P, self.rnn_input_dim), h0, self.GRU, agents=P, steps=steps)
        return feat.reshape(steps, batch, P, self.rnn_hidden_dim)

    def head_weight(self):
        """the output layer's effective weight outside autograd, or None when the module call is needed.  A spectrally normalised
        head (the critic's, :485) takes its pre-forward hook -- power iteration on u, v in place, weight / sigma -- as one launch
        (ops.spectral_norm_weight) instead of the hook's ~14; under autograd the module call (the hook differentiates through
        weight / sigma)."""
        m = self.Mean
        if torch.is_grad_enabled() or not m.bias.is_cuda:
            return None
        hooks = list(m._forward_pre_hooks.values())
        if not hooks:
            return m.weight
        if (len(hooks) == 1 and isinstance(hooks[0], SpectralNorm) and hooks[0].name == "weight" and hooks[0].dim == 0 and m.weight_orig.dim() == 2
                and m.weight_orig.shape[0] <= 16 and m.weight_orig.shape[1] <= 1024):
            return ops.spectral_norm_weight(m.weight_orig, m.weight_u, m.weight_v, hooks[0].eps,
                                            hooks[0].n_power_iterations if m.training else 0)
        return None

    def head(self, feat, out=None):
        """the output layer on GRU features; out: dense storage written in place (rollout)."""
        w = self.head_weight()
        if w is None:
            y = self.Mean(feat)
            return y if out is None else out.copy_(y.reshape(out.shape))
        return ops.head_linear(feat, w, self.Mean.bias, out=out)

    def get_weights(self):
        return {k: v.cpu() for k, v in self.state_dict().items()}

    def set_weights(self, weights):
        self.load_state_dict(weights)

    def get_gradients(self):
        """list in .parameters() order, None allowed (reference :464-469)"""
        return [None if p.grad is None else p.grad.data.cpu().numpy() for p in self.parameters()]

    def set_gradients(self, gradients, device):
        for g, p in zip(gradients, self.parameters()):
            if g is not None:
                p.grad = torch.as_tensor(g).to(device)


class SharedActor(_Trunk):
    def __init__(self, shared_net, rnn_input_dim, action_dim, num_layers, rnn_hidden_dim, is_sn=False):
        super().__init__()
        self.shared_net = shared_net
        self.num_layers = num_layers
        self.rnn_input_dim = rnn_input_dim
        self.rnn_hidden_dim = rnn_hidden_dim
        self.GRU = nn.GRU(rnn_input_dim, rnn_hidden_dim, num_layers)
        self.Mean = _make_linear(rnn_hidden_dim, action_dim, is_sn, HeadLinear)

    def forward(self, obs, hist, hidden_state=None, mode=0, batch=None, steps=None, inplace_hidden=False, emb_out=None):
        """mode 0 (one step for R environments): returns prob (R,P,A), hidden, embedding (R,P,E)   (:422-425)
        mode 1 (sequences, rows ordered (episode, step)): returns prob (batch,steps,P,A), None, embedding  (:426-437)"""
        emb = self.shared_net(obs["p_state"], obs["e_state"], obs["o_state"], obs["p_adj"], obs["e_adj"], _o_adj(obs), hist,
                              False, None, obs.get("q_div", 1), emb_out)
        if mode == 0:
            feat, hidden_state = self._rollout_features(emb, hidden_state, inplace_hidden)
            return torch.softmax(self.Mean(feat), dim=-1), hidden_state, emb
        feat = self._sequence_features(emb, batch, steps)                       # (steps, batch, P, H)
        prob = torch.softmax(self.Mean(feat), dim=-1).permute(1, 0, 2, 3)       # the (.., A) result is what gets permuted
        return prob, None, emb

    def get_logprob_and_entropy(self, obs, hist, action, batch, steps):
        """Categorical(prob).log_prob / entropy (:451-456)"""
        prob, _, _ = self.forward(obs, hist, None, 1, batch, steps)
        dist = torch.distributions.Categorical(prob)
        return dist.log_prob(action), dist.entropy()


class SharedCritic(_Trunk):
    def __init__(self, shared_net, rnn_input_dim, value_dim, num_layers, rnn_hidden_dim, is_sn=False):
        super().__init__()
        self.shared_net = shared_net
        self.num_layers = num_layers
        self.rnn_input_dim = rnn_input_dim
        self.rnn_hidden_dim = rnn_hidden_dim
        self.GRU = nn.GRU(rnn_input_dim, rnn_hidden_dim, num_layers)
        head = _ortho_linear(rnn_hidden_dim, value_dim, HeadLinear)
        self.Mean = spectral_norm(head) if is_sn else head  # the only spectrally normalised layer (:485)

    def forward(self, obs, hist, hidden_state=None, mode=0, batch=None, steps=None, rollout=False, inplace_hidden=False,
                emb_out=None):
        kvalid = obs.get("o_kvalid") if rollout else None
        emb = self.shared_net(obs["p_state"], obs["e_state"], obs["o_state"], obs["p_adj"], obs["e_adj"], _o_adj(obs), hist,
                              True, kvalid, obs.get("q_div", 1), emb_out)
        if mode == 0:
            feat, hidden_state = self._rollout_features(emb, hidden_state, inplace_hidden)
            return self.head(feat), hidden_state, emb
        feat = self._sequence_features(emb, batch, steps)                       # (steps, batch, P, H)
        return self.Mean(feat).permute(1, 0, 2, 3)


def pair_embeddings(actor, critic, obs, hist_a, hist_c):
    """the encoder half of sequence_forward_pair: (emb_a, emb_c), rows in (episode, step, agent) order"""
    enc, q_div = actor.shared_net, obs.get("q_div", 1)
    if (enc is critic.shared_net and isinstance(enc, DHGN) and obs["p_adj"].dtype == torch.float32
            and ops.msg_agg3_pair_train_ok(obs["p_state"], obs["o_state"], enc.MSG_layers[2].weight, q_div)):
        # shared DHGN: one message pass for both networks, forward and backward
        h0a, h0c = enc.encoder_pair_train(obs["p_state"], obs["e_state"], obs["o_state"], obs["p_adj"], obs["e_adj"], _o_adj(obs), q_div)
        return enc.fcra(h0a, hist_a, obs["p_adj"], False), enc.fcra(h0c, hist_c, obs["p_adj"], True)
    emb_a = actor.shared_net(obs["p_state"], obs["e_state"], obs["o_state"], obs["p_adj"], obs["e_adj"], _o_adj(obs), hist_a, False, None, q_div, None)
    emb_c = critic.shared_net(obs["p_state"], obs["e_state"], obs["o_state"], obs["p_adj"], obs["e_adj"], _o_adj(obs), hist_c, True, None, q_div, None)
    return emb_a, emb_c


def pair_heads(actor, critic, feat_a, feat_c):
    """the output layers on time-major GRU features (steps, batch, P, H) -> prob (batch, steps, P, A), values (batch, steps, P, 1)"""
    return torch.softmax(actor.Mean(feat_a), dim=-1).permute(1, 0, 2, 3), critic.Mean(feat_c).permute(1, 0, 2, 3)


def sequence_forward_pair(actor, critic, obs, hist_a, hist_c, batch, steps):
    """SharedActor.forward(mode 1) and SharedCritic.forward(mode 1) of one mini-batch together (DHGN/mappo_parallel.py:426-437,
    :503-520): the two encoders as before, then the two GRUs layer by layer with actor and critic in ONE persistent launch each way
    (ops.gru_multi) -- same numbers as the two module calls.  -> (prob (batch, steps, P, A), values (batch, steps, P, 1))."""
    emb_a, emb_c = pair_embeddings(actor, critic, obs, hist_a, hist_c)
    P = emb_a.shape[1]
    if not (actor.use_rnn and critic.use_rnn):
        feat_a, feat_c = actor._sequence_features(emb_a, batch, steps), critic._sequence_features(emb_c, batch, steps)
    else:
        h0 = [torch.zeros(m.num_layers, batch * P, m.rnn_hidden_dim, dtype=emb_a.dtype, device=emb_a.device) for m in (actor, critic)]
        fa, fc = ops.gru_multi([emb_a.reshape(batch * steps * P, actor.rnn_input_dim), emb_c.reshape(batch * steps * P, critic.rnn_input_dim)],
                               h0, [actor.GRU, critic.GRU], agents=P, steps=steps,
                               x_links=[getattr(emb_a, "relu_link", None), getattr(emb_c, "relu_link", None)], zero_state=True)
        feat_a, feat_c = fa.reshape(steps, batch, P, actor.rnn_hidden_dim), fc.reshape(steps, batch, P, critic.rnn_hidden_dim)
    return pair_heads(actor, critic, feat_a, feat_c)


def build_actor_critic(cfg, device):
    """encoder -> actor -> critic, the construction order of MAPPO.__init__ (:582-616); encoder is shared.
    `algo.encoder: gnn_extractor` (ours; default `dhgn`) selects the older one-hop encoder instead (GnnEncoder below; reference
    obstacle_differ_3hop/mappo_parallel.py:277-302: one instance per network, constructed actor first)."""
    sn = cfg.algo.use_spectral_norm
    kind = str(cfg.algo.get("encoder", "dhgn")).lower()
    if kind in ("gnn_extractor", "gnnextractor", "gnn"):
        if int(cfg.algo.depth) != GnnEncoder.HISTORY:
            raise ValueError(f"algo.encoder=gnn_extractor reads each network's own last {GnnEncoder.HISTORY} embeddings: set algo.depth={GnnEncoder.HISTORY}")
        enc = GnnEncoder(cfg.env.state_dim, cfg.algo.embedding_dim, sn, cfg.algo)
        enc_c = GnnEncoder(cfg.env.state_dim, cfg.algo.embedding_dim, sn, cfg.algo)
    elif kind == "dhgn":
        enc = enc_c = DHGN(cfg.env.state_dim, cfg.algo.embedding_dim, sn, cfg.algo, device)
    else:
        raise ValueError(f"algo.encoder={kind!r}: 'dhgn' or 'gnn_extractor'")
    actor = SharedActor(enc, cfg.algo.embedding_dim, cfg.env.action_dim, cfg.algo.num_layers, cfg.algo.rnn_hidden_dim, is_sn=sn)
    critic = SharedCritic(enc_c, cfg.algo.embedding_dim, 1, cfg.algo.num_layers, cfg.algo.rnn_hidden_dim, is_sn=sn)
    if not bool(cfg.algo.get("use_rnn", True)):
        if cfg.algo.embedding_dim != cfg.algo.rnn_hidden_dim:
            raise ValueError("algo.use_rnn=false feeds the embedding to the heads: embedding_dim must equal rnn_hidden_dim")
        actor.use_rnn = critic.use_rnn = False
    return actor.to(device), critic.to(device)


class GnnExtractor(nn.Module):
    """The older one-hop encoder variant (reference obstacle_differ_3hop/mappo_parallel.py:34-70): per-pair
    features through a two-layer MLP, mean over the L1-normalised adjacency, concatenated with the aggregated previous
    communication embeddings and squeezed by a bottleneck layer.  Same parameter names (`one_hop.{0,2}`, `bottleneck.0`) and
    initialisation stream as the reference, so its checkpoints load.  Its environment is not in the reference: block-level
    parity only (tests/test_gnn_extractor.py against tests/golden/gnn_extractor.npz).
      obs (*, A, K, F), last_comm_embedding (*, A, 2 O), adj (*, A, K) -> (*, A, O)"""

    def __init__(self, input_size, middle_size, output_size, n_hops: int = 1, is_sn: bool = False):
        super().__init__()
        self.n_hop = n_hops
        lin = (lambda a, b: _ortho_linear(a, b)) if is_sn else (lambda a, b: nn.Linear(a, b))
        self.one_hop = nn.Sequential(lin(input_size, middle_size), nn.ReLU(), lin(middle_size, output_size), nn.ReLU())
        self.bottleneck = nn.Sequential(lin(output_size + 2 * output_size, output_size), nn.ReLU())

    def forward(self, obs, last_comm_embedding=None, adj=None):
        A = adj.shape[-2]
        abar = F.normalize(adj, p=1, dim=-1)
        h0 = self.one_hop(obs)                                               # (*, A, K, O)
        h0_agg = torch.matmul(abar.unsqueeze(-2), h0).squeeze(-2)            # (*, A, O)
        comm_agg = torch.matmul(abar[..., :A], last_comm_embedding)          # the first A neighbours are the agents (:62-65)
        return self.bottleneck(torch.cat([h0_agg, comm_agg], dim=-1))


class GnnEncoder(GnnExtractor):
    """`algo.encoder: gnn_extractor` -- GnnExtractor as the `shared_net` of SharedActor / SharedCritic on the pursuit-evasion game
    ("the obstacle_differ_3hop.GnnExtractor variant as an alternative encoder").  Same parameters and forward as the
    block above (`one_hop.{0,2}`, `bottleneck.0`: the reference's checkpoints of that encoder load); what is OURS is the observation
    contract, because the environment that fed the reference's 3-hop agent its `state (*, A, K, F)` / `adj (*, A, K)` tensors is not
    in the reference tree (parity beyond the block is unpinned, tests/test_gnn_extractor.py):
      entities of a row: the P defenders, the evader, the O padded boundary obstacles (K = P + 1 + O);
      obs[i][j] = [state_j - state_i (4), one-hot(type of j) (3)];  adj = [p_adj | e_adj | o_adj] (the actor; the critic: ones,
      obstacle_differ_3hop/mappo_parallel.py:193);  last_comm_embedding = this network's own last two embeddings [t-2 | t-1]
      (:459-461, :526-527) -- MAPPO's history plumbing with algo.depth = 2 and a clean per-network history.
    The per-pair two-layer MLP (P K (2 F E + 2 E^2) flops per row: 51 MFLOP at P = 8, K = 185 -- seven times the whole DHGN network)
    runs in plain torch GEMMs, row-chunked; it is an alternative for small batches, not a benchmark path."""
    HISTORY = 2
    PAIR_FEATURES = 7
    CHUNK_PAIRS = 1 << 21

    def __init__(self, input_dim, embedding_dim, is_sn, algo_config):
        super().__init__(self.PAIR_FEATURES, int(algo_config.get("gnn_middle_dim", embedding_dim)), embedding_dim, is_sn=bool(is_sn))
        self.depth = self.HISTORY
        self.input_dim, self.embedding_dim = input_dim, embedding_dim

    def observation(self, p, e, o, adj_p, adj_e, adj_o, is_critic, q_div=1):
        """-> obs (R, P, K, 7), adj (R, P, K)"""
        R, P = p.shape[0], p.shape[1]
        O = o.shape[1]
        if adj_o.dtype == torch.int32:
            adj_o = ops.unpack_adj_bits(adj_o, O)
        oq = o if q_div == 1 else o.repeat_interleave(q_div, dim=0)
        q = torch.cat([p, e.reshape(R, 1, -1), oq], dim=1)                                  # (R, K, 4)
        idx = torch.arange(P + 1 + O, device=p.device)       # device-side construction: this runs inside captured tick programs
        kind = torch.stack((idx < P, idx == P, idx > P), dim=-1).to(p.dtype)
        rel = q[:, None, :, :] - p[:, :, None, :]                                           # (R, P, K, 4)
        obs = torch.cat([rel, kind.expand(R, P, P + 1 + O, 3)], dim=-1)
        adj = torch.cat([adj_p, adj_e, adj_o], dim=-1)
        return obs, (torch.ones_like(adj) if is_critic else adj)

    def forward(self, p, e, o, adj_p, adj_e, adj_o, hist, is_critic, o_kvalid=None, q_div=1, out=None):
        R, P = p.shape[0], p.shape[1]
        K = P + 1 + o.shape[1]
        last = torch.cat([hist[1], hist[0]], dim=-1).reshape(R, P, 2 * self.embedding_dim)  # hop k = the embedding of step t-1-k
        step = max(q_div, (self.CHUNK_PAIRS // (P * K)) // q_div * q_div) if q_div > 1 else max(1, self.CHUNK_PAIRS // (P * K))
        parts = []
        for r0 in range(0, R, step):
            r1 = min(R, r0 + step)
            oc = o[r0 // q_div: (r1 + q_div - 1) // q_div] if q_div > 1 else o[r0:r1]
            obs, adj = self.observation(p[r0:r1], e[r0:r1], oc, adj_p[r0:r1], adj_e[r0:r1], adj_o[r0:r1], is_critic, q_div)
            parts.append(GnnExtractor.forward(self, obs, last[r0:r1], adj))
        y = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        if out is not None:
            out.copy_(y)
            return out
        return y
