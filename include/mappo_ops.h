/*
 * mappo_ops.h -- C ABI of the MI355X (gfx950) fused ops of the MAPPO update/rollout path.
 *
 * These replace torch op sequences of the reference (DHGN/mappo_parallel.py); there is no native boundary in the
 * reference, the seam is the Python methods cited per entry point.  All pointers are DEVICE pointers owned by the
 * caller, kernels are enqueued on the caller's hipStream_t (passed as void*), return value 0 == success
 * (otherwise a hipError_t / MO_ERR_* code).  fp32 throughout (the 1e-4 parity tolerance rules out bf16).
 */
#ifndef MAPPO_OPS_H
#define MAPPO_OPS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MO_ERR_BAD_ARG 20001

/* adjacency source of dhgn_msg_agg_* */
enum { MO_ADJ_TENSOR = 0,   /* actor: the observed adjacency                                            */
       MO_ADJ_ONES = 1,     /* critic: torch.ones_like(adj) over all K neighbours (AttributeDataset, :64-65) */
       MO_ADJ_VALID = 2,    /* critic in a batched rollout: ones over the first kvalid[row] real neighbours  */
       MO_ADJ_BITS = 3 };   /* actor, 0/1 adjacency bit-packed: adj is uint32 [R][P][MO_ADJ_ROW_WORDS(K)], bit j of row i =
                               adj[i][j] (the env's o_adj_bits, include/pe_env.h); identical results to MO_ADJ_TENSOR  */
#define MO_ADJ_ROW_WORDS(K) (((((K) + 31) >> 5) + 3) & ~3)

/*
 * Vertex-level message + mean aggregation of one DHGN relation, never materialising the (R,P,K,E) message:
 *   out[r,i,:] = sum_j abar[r,i,j] * ReLU( W[:, :4] (p[r,i] - q[r',j]) + W[:, 4:8] (p[r,i] - e[r]) + b )
 *   abar = adj / max(sum_j |adj|, 1e-12)            (F.normalize(adj, p=1, dim=-1))
 * i.e. DHGN.coordinate + DHGN.message + the matmul of DHGN.mean_operator (DHGN/mappo_parallel.py:235-239,
 * 323-334, 346-347) for relation 0 (q = p, din = 8, e given), 1 (q = e, K = 1) and 2 (q = obstacles).
 *   R rows, P agents, K neighbours, E features (multiple of 64, <= 256), din in {4, 8};
 *   p [R][P][4] (rows p_row_stride elements apart, likewise e and adj: rows may be slices buffer[:, t] of (N,T,..)
 *   replay-buffer tensors); q [R/q_div][K][4] (q_div p-rows share one q-row: obstacles are static over an episode);
 *   e [R][4] or NULL (din == 4); adj [R][P][K] float (MO_ADJ_TENSOR) or packed words (MO_ADJ_BITS); kvalid [R/q_div] (MO_ADJ_VALID);
 *   W [E][din]; b [E]; out [R][P][E] with out_stride elements between consecutive [E] vectors (out_stride = 3E
 *   writes relation r of an [R][P][3][E] tensor when out points at slot r).
 */
int dhgn_msg_agg_fwd(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, const float *p, int64_t p_row_stride,
                     const float *q, int64_t q_row_stride, int32_t q_div, const float *e, int64_t e_row_stride, const void *adj,
                     int64_t adj_row_stride, int32_t adj_mode, const int32_t *kvalid, const float *W, const float *b,
                     float *out, int64_t out_stride /* elements between consecutive (row, agent) vectors, >= E */, void *stream);

/*
 * The three relations of DHGN.encoder for the same R rows in ONE launch (DHGN/mappo_parallel.py:256-281): relation r writes
 * slot r of out [R][P][3][E] (out_stride = elements between consecutive (row, agent) vectors, >= 3 E).  Same arithmetic as
 * three dhgn_msg_agg_fwd calls (bit-identical); exists because the rollout's per-tick calls are launch-latency bound.
 */
typedef struct mo_msg_rel {
    int32_t K, din, q_div, adj_mode;
    const float *q; int64_t q_rs;       /* [R/q_div][K][4] */
    const float *e; int64_t e_rs;       /* [R][4] or NULL (din == 4) */
    const void *adj; int64_t adj_rs;    /* MO_ADJ_TENSOR: float [R][P][K]; MO_ADJ_BITS: packed words; else NULL */
    const int32_t *kvalid;              /* MO_ADJ_VALID: [R/q_div] */
    const float *W, *b;                 /* [E][din], [E] */
} mo_msg_rel;
int dhgn_msg_agg3_fwd(const mo_msg_rel *rel /* [3] */, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_row_stride, float *out,
                      int64_t out_stride, void *stream);
/*
 * Actor AND critic of one rollout tick in one launch.  Both networks hold the same DHGN instance (mappo_parallel.py:582-616,
 * critic built on the actor's shared_net), so the messages ReLU(W [p_i - q_j, ..] + b) are the same numbers; only the mean's
 * weights differ: rel[r] carries the ACTOR's adjacency (MO_ADJ_TENSOR or MO_ADJ_BITS; rel[r].kvalid unused), the critic's is
 * all ones (AttributeDataset :64-65) -- for the obstacle relation over the first o_kvalid[row / q_div] neighbours when
 * o_kvalid != NULL (batched rollout), else over all K.  out_actor / out_critic [R][P][3][E], bit-identical to
 * dhgn_msg_agg3_fwd called once per network.
 */
int dhgn_msg_agg3_pair_fwd(const mo_msg_rel *rel /* [3] */, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_row_stride,
                           const int32_t *o_kvalid, float *out_actor, float *out_critic, int64_t out_stride, void *stream);

/* The update's form of the paired launch: the actor's three relations and the critic's relations 0 and 1 (ones adjacency); slot 2
 * of out_critic is NOT written -- in training the critic's obstacle relation averages over all K padded slots and is
 * left to dhgn_msg_agg_ones_sorted_fwd. */
int dhgn_msg_agg3_pair01_fwd(const mo_msg_rel *rel /* [3] */, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_row_stride,
                             float *out_actor, float *out_critic, int64_t out_stride, void *stream);
/* Backward of ONE relation for actor and critic together (shared weights W, b: DHGN/mappo_parallel.py:582-616): the actor's
 * gradient under the float adjacency adj [R][P][K] and the critic's under ones over all K neighbours, summed into dW / db in one
 * pass over the messages (g_ij = [z_ij > 0] (abar_ij gout_actor_i + gout_critic_i / K)).  Arguments as dhgn_msg_agg_bwd. */
int dhgn_msg_agg_bwd_pair(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, const float *p, int64_t p_row_stride, const float *q,
                          int64_t q_row_stride, int32_t q_div, const float *e, int64_t e_row_stride, const float *adj, int64_t adj_row_stride,
                          const float *W, const float *b, const float *gout_actor, const float *gout_critic, int64_t gout_stride, float *dW,
                          float *db, void *workspace, void *stream);
/* The same launch, which also writes the position part of DHGN's semantic layer for these rows, pos[r][i][:] = bp + Wp p[r][i]
 * (Wp [E][4]: the first four input columns of semantic_layer.weight, wp_row_stride elements between its rows; mappo_parallel.py:
 * 284-303) -- the addend the embedding part of that layer accumulates into; the same numbers for both networks, written to
 * pos_actor and (if not NULL) pos_critic, [R][P] vectors of E floats, pos_stride (>= E) elements apart (E: dense; 2 E: the right
 * half of a [R P][2 E] operand that a later layer reads as one concatenated input). */
int dhgn_msg_agg3_pair_pos_fwd(const mo_msg_rel *rel /* [3] */, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_row_stride,
                               const int32_t *o_kvalid, float *out_actor, float *out_critic, int64_t out_stride, const float *Wp,
                               int64_t wp_row_stride, const float *bp, float *pos_actor, float *pos_critic, int64_t pos_stride, void *stream);

/*
 * Backward of the above w.r.t. W and b (the inputs are data, they carry no gradient): recomputes the
 * pre-activation, masks with ReLU', reduces over all (r,i,j) without atomics (per-workgroup partials in
 * `workspace`, then one deterministic second pass).  gout [R][P][E]; dW [E][din]; db [E] are OVERWRITTEN.
 * workspace: at least dhgn_msg_agg_bwd_workspace(E, din) bytes.
 */
int64_t dhgn_msg_agg_bwd_workspace(int32_t E, int32_t din);
int dhgn_msg_agg_bwd(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, const float *p, int64_t p_row_stride,
                     const float *q, int64_t q_row_stride, int32_t q_div, const float *e, int64_t e_row_stride, const void *adj,
                     int64_t adj_row_stride, int32_t adj_mode, const int32_t *kvalid, const float *W, const float *b,
                     const float *gout, int64_t gout_stride, float *dW, float *db, void *workspace, void *stream);

/*
 * The same aggregate for adjacency == ones over a neighbour set shared by q_div consecutive rows (the critic's obstacle relation
 * in training: torch.ones_like(adj) over the padded obstacle slots, AttributeDataset :64-65, q = the episode's obstacles) in
 * O(log K) per (row, agent, feature) instead of O(K): per (q-row, feature) the K pre-activation offsets d_j = W q_j are sorted
 * once, sum_j relu(c - d_j) = m c - prefix_d[m] with m = #{d_j < c} found by a two-level rank search (csrc/mappo_ops.hip
 * k_msg_ones_sorted_*).  din == 4, K <= 255, E a multiple of 16, R a multiple of q_div (dhgn_msg_agg_ones_sorted_ok).
 * fwd: save_m (R P E bytes) and qtab ((R / q_div) E 4 (K + 1) floats) -- both may be NULL without a backward pass -- are what bwd
 * consumes; their layouts are private to the kernel pair ([q-row][block of 16 features][pair][16] and [q-row][block][4][K+1][16]);
 * bwd: dW [E][4], db [E] OVERWRITTEN; partials: scratch of the size dhgn_msg_agg_ones_sorted_workspace reports.
 * Sums are reassociated (prefix sums): results agree with dhgn_msg_agg_fwd/bwd(MO_ADJ_ONES) to fp32 rounding, not bit for bit.
 */
int dhgn_msg_agg_ones_sorted_ok(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, int32_t q_div);
int64_t dhgn_msg_agg_ones_sorted_workspace(int32_t R, int32_t P, int32_t K, int32_t E, int32_t q_div, int64_t *m_bytes, int64_t *qtab_bytes,
                                           int64_t *partial_bytes);
int dhgn_msg_agg_ones_sorted_fwd(int32_t R, int32_t P, int32_t K, int32_t E, const float *p, int64_t p_row_stride, const float *q,
                                 int64_t q_row_stride, int32_t q_div, const float *W, const float *b, float *out, int64_t out_stride,
                                 uint8_t *save_m, float *qtab, void *stream);
int dhgn_msg_agg_ones_sorted_bwd(int32_t R, int32_t P, int32_t K, int32_t E, const float *p, int64_t p_row_stride, int32_t q_div,
                                 const float *gout, int64_t gout_stride, const uint8_t *save_m, const float *qtab, float *dW, float *db,
                                 void *partials, void *stream);

/*
 * GAE reverse scan + value target + advantage normalisation (DHGN/mappo_parallel.py:643-658):
 *   delta = (r + gamma v[:,1:] - v[:,:-1]) * active ; gae_t = delta_t + gamma lamda gae_{t+1}
 *   v_target = adv + v[:,:-1] ; if use_adv_norm: adv = (adv - mean) / (std_unbiased + 1e-5) * active
 * r, active, adv, v_target [N][T][P]; v [N][T+1][P].  Statistics over all N*T*P elements, accumulated in f64 through per-workgroup
 * partials added in index order (no atomics: the same input gives the same bits every run).
 * stats (device, gae_advnorm_workspace() bytes; the first 4 doubles: sum, sum of squared deviations, mean, std) is scratch + output.
 */
int64_t gae_advnorm_workspace(void);
int gae_advnorm(int32_t N, int32_t T, int32_t P, const float *r, const float *v, const float *active, float gamma,
                float lamda, int32_t use_adv_norm, float *adv, float *v_target, double *stats, void *stream);

/*
 * Categorical(probs).sample() + log_prob for a batch of rows (DHGN/mappo_parallel.py:446-448) with a counter-based
 * generator (Philox4x32-10; stream = (seed, offset + row)): probs [R][A] -> action [R] int32, logp [R].
 * greedy != 0 gives probs.argmax(-1) instead (choose_action(deterministic=True), :442-444; first maximum wins).
 */
int categorical_sample(int32_t R, int32_t A, const float *probs, uint64_t seed, uint64_t offset, int32_t greedy,
                       int32_t *action, float *logp, void *stream);
/* Same with the stream offset kept in device memory (*counter, advanced by R after sampling), so the call can be
 * captured in a hipGraph and replayed with fresh random numbers. */
int categorical_sample_counter(int32_t R, int32_t A, const float *probs, uint64_t seed, uint64_t *counter, int32_t greedy,
                               int32_t *action, float *logp, void *stream);

/*
 * The pre-forward hook of torch.nn.utils.spectral_norm on a small head (the reference's value head, DHGN/mappo_parallel.py:485)
 * as one launch: n_power_iterations times  v = normalize(W^T u), u = normalize(W v)  IN PLACE (as the hook does, also under
 * no_grad), then sigma = u . (W v) and w_eff = W / sigma.  W, w_eff [A][H], u [A], v [H]; A <= 16, H <= 1024;
 * normalize(x) = x / max(||x||_2, eps).  n_power_iterations = 0 is the hook's eval-mode form (u, v untouched).
 */
int spectral_norm_weight(int32_t A, int32_t H, const float *W, float *u, float *v, float eps, int32_t n_power_iterations, float *w_eff,
                         void *stream);

/*
 * Output heads of one rollout tick (choose_action / get_value, DHGN/mappo_parallel.py:422-448), H = 128 features, A <= 16 outputs:
 *   head_linear: y [R][A] = feat [R][H] W^T + b  (the value head: A = 1, y = the rollout's value storage).
 *   head_sample: softmax(feat W^T + b) -> Categorical sample (Philox stream (seed, *counter + row), as categorical_sample_counter)
 *                and its log-probability, or the argmax when greedy != 0: action [R] int32, logp [R]; *counter advances by R.
 *                ticket: one zero-initialised uint32 in device memory owned by the caller (the kernel leaves it zero).
 * One launch each instead of GEMM + bias (+ softmax + sample + counter update); no autograd (rollout only).
 */
int head_linear(int32_t R, int32_t A, int32_t H, const float *feat, const float *W, const float *b, float *y, void *stream);
int head_sample(int32_t R, int32_t A, int32_t H, const float *feat, const float *W, const float *b, uint64_t seed, uint64_t *counter,
                uint32_t *ticket, int32_t greedy, int32_t *action, float *logp, void *stream);

/*
 * torch.nn.GRU cell between the two MFMA GEMMs (reference DHGN/mappo_parallel.py:397,424,434; gate order r, z, n):
 *   r = s(gi_r + gh_r + b_hr) ; z = s(gi_z + gh_z + b_hz) ; hn = gh_n + b_hn ; n = tanh(gi_n + r hn) ; h' = (1-z) n + z h
 * gi = x W_ih^T + b_ih [B][3H], gh = h W_hh^T [B][3H] (bias b_hh [3H] added here), h_prev / h_out [B][H];
 * save (training) [4][B][H] receives r, z, n, hn.  The backward kernel turns dL/dh' (dout + dcarry, either may be
 * NULL... dcarry may be NULL) into dgi, dgh [B][3H] and the direct path dh_direct = dh' * z [B][H].
 */
int gru_gates_fwd(int32_t B, int32_t H, const float *gi, const float *gh, const float *b_hh, const float *h_prev, float *h_out,
                  float *save, void *stream);
int gru_gates_bwd(int32_t B, int32_t H, const float *dout, const float *dcarry, const float *save, const float *h_prev, float *dgi,
                  float *dgh, float *dh_direct, void *stream);

/*
 * Whole-sequence GRU layer in one persistent launch (H = 128 only): gi [T B][3H] is the input projection
 * x W_ih^T + b_ih (one MFMA GEMM), the recurrence over T runs inside the kernel with W_hh held in registers as
 * v_mfma_f32_16x16x4_f32 operands and the h tile (16 batch rows per workgroup) in LDS.  out [T][B][H] time-major.
 * save (or NULL: no backward): gru_seq_save_elems(T, B) floats, the gates (r, z, n, hn) of every step in the kernel pair's own
 * lane order -- opaque, only gru_seq_bwd reads it.  The backward kernel consumes dout [T][B][H] (dL/d out) and produces
 * dgi [T B][3H] = (dr, dz, dn), dh0 [B][H] and, for the recurrent weight gradient dW_hh = [dr dz dnr]^T h_prev, EITHER
 * dgh [T][B][3H] = (dr, dz, dnr) time-major OR only dnr [T][B][H] (exactly one of the two pointers is non-NULL).  The second
 * form stores dr, dz once: valid when dgi itself is time-major (gi_agents == 0), the caller then takes the first 2H columns of
 * dgh from dgi.
 * gi_agents: row order of gi and dgi.  0: time-major, row t B + b.  P > 0: the order of the encoder's output rows
 * (episode n, step t, agent p) with sequence b = n P + p, row ((b / P) T + t) P + b % P (B a multiple of P) -- the input
 * projection and its gradients then run on the embedding as it lies in memory, without the permuted copies around the GRU
 * (_sequence_features, DHGN/mappo_parallel.py:426-437).
 * All pointers 16-byte aligned.
 */
int64_t gru_seq_save_elems(int32_t T, int32_t B);
/* Several independent layers in ONE launch (the actor's and the critic's layer of the same depth, for every mini-batch of a group:
 * same T and gi_agents; own weights, inputs and B): at a data-parallel rank's share of the batch one layer is a few dozen workgroups
 * that run T sequential steps, and a launch takes as long as at full size; together the layers fill the chip in the same time.
 * Records of HOST memory holding DEVICE pointers, meaning as in gru_seq_fwd / gru_seq_bwd. */
#define MO_GRU_MAX_NETS 24
#define MO_GRU_CELL_MAX_NETS 4
/* B (per record): this layer's number of sequences, <= the launch's B argument (0: the launch's B) -- the layers of one launch may be
 * ragged (the last mini-batch of an epoch is smaller); gi / out / save / dgi / ... of a record are sized by ITS B. */
typedef struct mo_gru_seq_net { const float *gi, *w_hh, *b_hh, *h0; float *out, *save; int32_t B, pad0; } mo_gru_seq_net;
typedef struct mo_gru_seq_bwd_net {
    const float *dout, *save, *out, *h0, *w_hh;
    float *dgi, *dgh, *dnr, *dh0, *db_ih, *db_hh;
    void *workspace;   /* >= gru_seq_bwd_workspace(B) bytes when db_ih / db_hh are requested; one per record */
    int32_t B, pad0;
} mo_gru_seq_bwd_net;
int gru_seq_fwd_multi(int32_t n_nets, const mo_gru_seq_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream);
int gru_seq_bwd_multi(int32_t n_nets, const mo_gru_seq_bwd_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream);
/* The same recurrences in fp32 arithmetic on the bf16 matrix pipe (exact three-way operand splits, csrc/sb_gru_seq.hpp): same records,
 * same (interchangeable) save layout, results equal to the fp32-MFMA kernels' to fp32 rounding.  `runtime.matmul: split_bf16`. */
int gru_seq_split_fwd_multi(int32_t n_nets, const mo_gru_seq_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream);
int gru_seq_split_bwd_multi(int32_t n_nets, const mo_gru_seq_bwd_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream);
int gru_seq_fwd(int32_t T, int32_t B, int32_t H, const float *gi, const float *w_hh, const float *b_hh, const float *h0, float *out,
                float *save, int32_t gi_agents, void *stream);
int64_t gru_seq_bwd_workspace(int32_t B);
/* db_ih / db_hh [3H] (optional, both or none): column sums of dgi / dgh, reduced from per-workgroup partials in
 * `workspace` (>= gru_seq_bwd_workspace(B) bytes) by a deterministic second pass. */
int gru_seq_bwd(int32_t T, int32_t B, int32_t H, const float *dout, const float *save, const float *out, const float *h0,
                const float *w_hh, float *dgi, float *dgh, float *dnr, float *dh0, float *db_ih, float *db_hh, int32_t gi_agents,
                void *workspace, void *stream);

/*
 * One torch.nn.GRU layer step for a large batch without autograd (the rollout's choose_action / get_value,
 * DHGN/mappo_parallel.py:422-425 with seq_len 1): h_out = GRUCell(x, h_prev), both projections and the gate math in
 * one persistent launch (weights in registers as fp32 MFMA operands).  I = H = 128; x, h_prev, h_out [B][H] dense;
 * w_ih, w_hh [3H][H] (torch gate order r, z, n); h_out must not alias x or h_prev.
 */
int gru_cell_fwd(int32_t B, int32_t H, const float *x, const float *h_prev, const float *w_ih, const float *w_hh, const float *b_ih,
                 const float *b_hh, float *h_out, void *stream);
/* Several independent cells of one shape (actor and critic) in ONE launch; records of HOST memory holding DEVICE pointers, meaning as
 * in gru_cell_fwd (n_nets <= MO_GRU_CELL_MAX_NETS).  The persistent workgroups (one per CU) are divided between the cells. */
typedef struct mo_gru_cell_net { const float *x, *h_prev, *w_ih, *w_hh, *b_ih, *b_hh; float *h_out; } mo_gru_cell_net;
int gru_cell_fwd_multi(int32_t n_nets, const mo_gru_cell_net *nets, int32_t B, int32_t H, void *stream);
/* The same cells in fp32 ARITHMETIC ON THE bf16 MATRIX PIPE: every fp32 operand is split exactly into three bf16 pieces
 * (x = x1 + x2 + x3) and a product is the six piece products with i + j <= 4 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation --
 * the dropped terms are below one fp32 rounding of the product; inputs, outputs and the state stay fp32 and the error against f64
 * equals the fp32-MFMA kernel's (csrc/mappo_ops.hip k_gru_cell_sb).  Same records and results to fp32 rounding as
 * gru_cell_fwd_multi, EXCEPT that h_out must not alias h_prev or x (two workgroups share a row tile). */
int gru_cell_split_fwd_multi(int32_t n_nets, const mo_gru_cell_net *nets, int32_t B, int32_t H, void *stream);
/* Y[R][128] = act(X[R][K] W[128][K]^T + bias [+ C]) in the same split arithmetic (fp32 in, fp32 out, fp32 accumulation; K in {128, 256,
 * 384}): the rollout's Linear layers (torch.nn.Linear of DHGN/mappo_parallel.py:148-233).  bias, C may be NULL; C may be Y (accumulate in
 * place); relu != 0 applies max(., 0) last.  ldx, ldw, ldc, ldy: row strides in floats (multiples of 4: X, C, Y may be column blocks of
 * wider matrices); all pointers 16-byte aligned. */
int sb_gemm_n128(int64_t R, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu, const float *C,
                 int64_t ldc, float *Y, int64_t ldy, void *stream);
/* The same for N outputs: (N, K) in {(128, 128), (128, 256), (128, 384), (256, 128), (384, 128)} -- (384, 128) is a GRU input projection
 * x W_ih^T + b_ih, (256, 128) the input gradient of an FCRA layer. */
int sb_gemm(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu, const float *C,
            int64_t ldc, float *Y, int64_t ldy, void *stream);
/* sb_gemm (N = 128, K in {128, 256}) that also writes the sign bits of its result: y_sign_bits[row * ld_bytes + j], bit k = (Y[row][8 j + k] > 0),
 * N / 8 bytes per row (rows ld_bytes apart: Y may be a column block of a wider matrix and the bytes a column block of its bit matrix).
 * Behind the ReLU epilogue these are relu'(Y): sb_gemm_masked_bits reads them in the backward of the layer that consumes Y. */
int sb_gemm_signs(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu, const float *C,
                  int64_t ldc, float *Y, int64_t ldy, uint8_t *y_sign_bits, int64_t ld_bytes, void *stream);
/* The input gradient of a Linear layer whose INPUT is the output of a ReLU layer, with that ReLU's backward and the bias gradient of
 * the layer in front of it in the same pass (autograd's threshold_backward + sum(0) of DHGN/mappo_parallel.py:148-233's relu(Linear)
 * chains):  Y = (X W^T) * (M > 0) on the first mask_cols columns (a multiple of 128; the others are plain X W^T),  colsum[N] = column
 * sums of Y.  M [R][N] (row stride ldm) is the saved ReLU output; (N, K) in {(256, 128), (384, 128), (128, 384)}.  The column sums are
 * reduced from per-workgroup partials in `workspace` (>= sb_gemm_masked_workspace(N) bytes) in a fixed order: same bits every run. */
int64_t sb_gemm_masked_workspace(int32_t N);
int sb_gemm_masked(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *M, int64_t ldm,
                   int32_t mask_cols, float *Y, int64_t ldy, float *colsum, void *workspace, void *stream);
/* The same with the mask as sign bits (sb_gemm_signs' layout, N / 8 bytes per row, row stride ld_bytes): 1 bit per element read instead of 4 bytes. */
int sb_gemm_masked_bits(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const uint8_t *sign_bits,
                        int64_t ld_bytes, int32_t mask_cols, float *Y, int64_t ldy, float *colsum, void *workspace, void *stream);

/*
 * PPO clipped-surrogate policy loss and clipped value loss of one mini-batch with their gradients
 * (DHGN/mappo_parallel.py:692-706):
 *   ratio = exp(logp_now - logp_old); actor = -min(ratio adv, clamp(ratio, 1-eps, 1+eps) adv) - entropy_coef entropy
 *   critic = use_value_clip ? max((clamp(v_now - v_old, -eps, eps) + v_old - v_target)^2, (v_now - v_target)^2) : (v_now - v_target)^2
 *   losses[0] = sum(actor * active) / active_sum ; losses[1] = sum(critic * active) / active_sum
 * grad_* = d(losses[0] + losses[1]) / d(logp_now | entropy | values_now), autograd's tie rules for min / max / clamp.
 * n elements (= episodes x T x P) per tensor, dense fp32; active_sum: device scalar sum(active); workspace >=
 * ppo_loss_workspace() bytes.  Deterministic (f64 partial sums added in a fixed order).
 */
int64_t ppo_loss_workspace(void);
int ppo_loss_fwd_bwd(int64_t n, const float *logp_now, const float *entropy, const float *logp_old, const float *adv, const float *active,
                     const float *values_now, const float *values_old, const float *v_target, const float *active_sum, float epsilon,
                     float entropy_coef, int32_t use_value_clip, float *losses, float *grad_logp, float *grad_entropy, float *grad_values,
                     void *workspace, void *stream);
/* The same losses from the policy's probabilities prob[.., A] (A <= 16): torch.distributions.Categorical(prob).log_prob(action) and
 * .entropy() (DHGN/mappo_parallel.py:451-456: renormalisation, the clamp of probs_to_logits, the gather) are evaluated inside the pass
 * and the gradient with respect to prob comes out instead of grad_logp / grad_entropy.  The n = d0 d1 d2 rows are indexed (i0, i1, i2);
 * prob and grad_prob are read / written at i0 p_s0 + i1 p_s1 + i2 p_s2 (+ k, k < A), values_now at i0 v_s0 + i1 v_s1 + i2 v_s2 (strides in
 * floats: the heads' outputs are time-major views); action (float-valued indices) and the other operands are dense in row order. */
int ppo_loss_prob_fwd_bwd(int64_t n, int32_t A, const float *prob, float *grad_prob, int64_t d1, int64_t d2, int64_t p_s0, int64_t p_s1, int64_t p_s2,
                          const float *action, const float *logp_old, const float *adv, const float *active, const float *values_now, int64_t v_s0,
                          int64_t v_s1, int64_t v_s2, const float *values_old, const float *v_target, const float *active_sum, float epsilon,
                          float entropy_coef, int32_t use_value_clip, float *losses, float *grad_values, void *workspace, void *stream);

/*
 * Records one rollout tick into the replay buffer (MAPPO.run_episode's minibuffer.store_transition,
 * DHGN/mappo_parallel.py:783-805, for N environments at once): for every item, row n of the dense [N][row_bytes] source
 * goes to dst + n * dst_row_stride (slot [n, t] of an (N, T, ...) buffer tensor); i32_to_f32 converts int32 actions to
 * the buffer's float32.  If raw != NULL, episode_return[n] += sum_p raw[n][p] (the evaluator's episode reward).
 * The item array is read on the host at call time.
 */
#define MO_RECORD_MAX_ITEMS 16
typedef struct {
    const void *src;
    void *dst;
    int64_t dst_row_stride; /* bytes between consecutive environments in dst */
    int32_t row_bytes;      /* multiple of 4 */
    int32_t i32_to_f32;
} mo_record_item;
int rollout_record(int32_t N, int32_t n_items, const mo_record_item *items, const float *raw, float *episode_return, int32_t P, void *stream);

/*
 * Weight gradient of a Linear / GRU projection, C = A^T B reduced over all K rows of a minibatch:
 *   C [M][N] (dense) = (accumulate ? C : 0) + sum_k A[k][:]^T B[k][:] ;  A [K][M] (lda), B [K][N] (ldb), row-major fp32.
 * Replaces the `grad_output.t() @ input` GEMMs autograd runs for torch.nn.Linear / torch.nn.GRU weights in
 * MAPPO.train (DHGN/mappo_parallel.py:660-708, loss.backward()).  Split-K over one workgroup per CU, fp32 MFMA,
 * partial tiles in `workspace` (>= wgrad_tn_workspace(M, N) bytes) added in a fixed order: results are deterministic.
 * M, N multiples of 128 (<= 1024), lda/ldb multiples of 4, A/B 16-byte aligned; anything else -> MO_ERR_BAD_ARG
 * (the caller keeps such shapes on the BLAS library).
 */
int64_t wgrad_tn_workspace(int32_t M, int32_t N);
int wgrad_tn(int64_t K, int32_t M, int32_t N, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int32_t accumulate,
             void *workspace, void *stream);
/* The same product in fp32 arithmetic on the bf16 matrix pipe (exact three-way operand splits, see gru_cell_split_fwd_multi): (M, N) in
 * {128, 256, 384}^2 with M + N <= 512 and M N < 65536; workspace >= wgrad_split_workspace(M, N) bytes; deterministic. */
int64_t wgrad_split_workspace(int32_t M, int32_t N);
int wgrad_split_tn(int64_t K, int32_t M, int32_t N, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int32_t accumulate,
                   void *workspace, void *stream);
/* The same with A given as two column blocks [A1 (K x M1) | A2 (K x M2)], M = M1 + M2 (each with its own row stride): one pass over B for
 * a product whose left operand lives in two tensors -- the GRU's dW_hh = [dr dz | dnr]^T h_prev (dr, dz: the first 2H columns of dgi). */
int wgrad_split_tn2(int64_t K, int32_t M1, int32_t M2, int32_t N, const float *A1, int64_t lda1, const float *A2, int64_t lda2, const float *B, int64_t ldb,
                    float *C, int32_t accumulate, void *workspace, void *stream);
/*
 * Neighbour mean of DHGN.fcra (DHGN/mappo_parallel.py:204-233: `torch.matmul(F.normalize(adj, p=1, dim=-1), hist)`), R rows of P
 * agents and E features:  out[r][i][:] = act( sum_j abar[r][i][j] z[r][j][:] + bias ),  act = ReLU when relu != 0.
 *   actor (out_actor != NULL): abar = adj / max(sum_j |adj|, 1e-12), adj [R][P][P] (adj_row_stride elements between rows);
 *   critic (out_critic != NULL): abar = 1 / P (the normalised ones_like(adj) of AttributeDataset :64-65).
 * z_* are read in place from a (N, T', P, E) history buffer: row r = (n, t) = (r / T, r % T) starts at n * episode_stride +
 * t * step_stride (T = 1 and episode_stride = P E: a dense [R][P][E] tensor); bias [E] or NULL; out_*: [R][P] vectors of E floats,
 * out_stride (>= E) elements apart (E: dense; 2 E: the left half of the [agg | h] operand of the FCRA layer, :227-231).
 * Either output may be NULL; both together serve the rollout's paired actor / critic tick.
 */
int fcra_neighbour_mean(int32_t R, int32_t P, int32_t E, int32_t T, const float *z_actor, int64_t za_episode_stride, int64_t za_step_stride,
                        const float *z_critic, int64_t zc_episode_stride, int64_t zc_step_stride, const float *adj, int64_t adj_row_stride,
                        const float *bias, int32_t relu, float *out_actor, float *out_critic, int64_t out_stride, void *stream);
/* Several such aggregations of one shape in ONE launch (the FCRA hops of a rollout tick: every hop reads a stored history slot, none
 * reads another hop's result): same R, P, E, T, strides, adjacency, relu and out_stride; own inputs, bias and outputs per job. */
#define MO_NBR_MAX_JOBS 4
typedef struct mo_nbr_job { const float *z_actor, *z_critic, *bias; float *out_actor, *out_critic; } mo_nbr_job;
int fcra_neighbour_mean_multi(int32_t n_jobs, const mo_nbr_job *jobs, int32_t R, int32_t P, int32_t E, int32_t T, int64_t za_episode_stride,
                              int64_t za_step_stride, int64_t zc_episode_stride, int64_t zc_step_stride, const float *adj, int64_t adj_row_stride,
                              int32_t relu, int64_t out_stride, void *stream);

/*
 * ReLU backward and the bias gradient of the Linear in front of it in one pass (autograd: aten::threshold_backward, then
 * grad.sum(0) re-reading it; MAPPO.train's loss.backward(), DHGN/mappo_parallel.py:660-708):
 *   gin [R][F] = gout * [y > 0] (y = the saved ReLU output);  colsum [F] = sum_r gin[r][:].
 * Rows are F contiguous floats, g_row_stride / y_row_stride (>= F, multiples of 4) elements apart for gout / y (a column block of
 * a wider matrix), dense for gin; F a multiple of 4 with F/4 dividing 256; 16-byte aligned pointers.  Deterministic
 * (per-workgroup partials in `workspace` >= relu_bwd_colsum_workspace(F) bytes, reduced in a fixed order in f64).
 */
int64_t relu_bwd_colsum_workspace(int32_t F);
int relu_bwd_colsum(int64_t R, int32_t F, const float *gout, int64_t g_row_stride, const float *y, int64_t y_row_stride, float *gin, float *colsum,
                    void *workspace, void *stream);

/*
 * Weight gradient of a Linear layer with at most 16 inputs or outputs (the K = 4 position part of DHGN's semantic layer, the action
 * and value heads; autograd's `grad_output.t() @ input`, DHGN/mappo_parallel.py:660-708) as one streaming pass:
 *   C [NS][F] = sum_r S[r][:]^T X[r][:]   (transposed != 0: C [F][NS]);  S [R][NS] (lds), X [R][F] (ldx), F a multiple of 64.
 * colsum_x [F] = sum_r X[r][:], colsum_s [NS] = sum_r S[r][:] (NULL skips): the bias gradient, whichever operand is grad_output.
 * Deterministic (per-workgroup partials in `workspace` >= wgrad_skinny_workspace(NS, F) bytes, reduced in a fixed order in f64).
 */
int64_t wgrad_skinny_workspace(int32_t NS, int32_t F);
int wgrad_skinny(int64_t R, int32_t NS, int32_t F, const float *S, int64_t lds, const float *X, int64_t ldx, int32_t transposed, float *C,
                 float *colsum_x, float *colsum_s, void *workspace, void *stream);

const char *mappo_ops_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif
