/*
 * mcts_oracle.c -- TEST INFRASTRUCTURE ONLY.  Parity pinned: every function
 * below is checked against outputs of the reference itself (the .npz files under tests/golden,
 * produced by tests/golden/make_fixtures.py importing /root/reference).
 *
 * A plain-C, single-tree, scalar restatement of the search in trunghng/muzero's
 * mcts/ package.  It exists so the tests, the smoke check and the benchmark's
 * CPU-baseline leg have something to compare the HIP engine with; nothing in
 * the product path (muzero_amd/) may link, load or call it.
 *
 * Arithmetic follows the reference's NumPy semantics, including
 *   - dtype promotion (f64 values/rewards; f64 priors for PUCT, f32 for Gumbel),
 *   - np.sum's pairwise/8-accumulator summation order (NumPy 2.2.6,
 *     numpy/_core/src/umath/loops_utils.h.src, pairwise_sum),
 *   - np.exp on float32: NumPy's own SIMD kernel (a third-party dependency of
 *     the reference, NumPy 2.2.6, loops_exponent_log.dispatch.c.src:
 *     Cody-Waite reduction + P5/Q2 rational), restated in np_expf() and
 *     verified bit-exact against np.exp in tests/test_oracle_numpy_kernels.py.
 * float64 exp/log go through libm; NumPy's AVX512 kernels for those may differ
 * by 1 ulp, which never changes a tree field (only argmax on <=1ulp ties and
 * the last bit of the returned policy), so those outputs are compared with a
 * 1e-12 relative tolerance in the tests and everything else exactly.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_A 64
#define ORC_ROOT 0
#define ORC_UNVISITED (-1)

typedef struct {
    int32_t A;              /* n_actions */
    int32_t S;              /* simulations */
    int32_t players;        /* 1 or 2 */
    int32_t gumbel;         /* 0: PUCT/MuZero, 1: Gumbel MuZero */
    double discount;        /* config.gamma */
    double c_base, c_init;  /* PUCT */
    int32_t max_considered; /* Gumbel */
    int32_t _pad;
    double c_visit, c_scale;
    const double *c_table;  /* optional [S+2]: c_init + log((n+c_base+1)/c_base) as NumPy evaluated it */
} orc_cfg;

/* Same field set as reference mcts/tree.py:26-40, caller-allocated, N = S+1. */
typedef struct {
    int32_t *node_visits;         /* [N] */
    double *node_values;          /* [N] */
    double *node_raw_values;      /* [N] */
    int32_t *parents;             /* [N] */
    int32_t *action_from_parent;  /* [N] */
    int32_t *to_play;             /* [N] */
    int32_t *children_index;      /* [N][A] */
    void *children_prior_logits;  /* [N][A] double (PUCT) or float (Gumbel) */
    int32_t *children_visits;     /* [N][A] */
    double *children_rewards;     /* [N][A] */
    double *children_values;      /* [N][A] */
    float *hidden_states;         /* [N][H] (may be NULL) */
    const double *root_legal;     /* [A] 0./1. */
    const double *gumbel_noise;   /* [A] or NULL */
    int32_t H;
    int32_t _pad;
} orc_tree;

/* ---------------------------------------------------------------- NumPy kernels */

/* np.sum over a contiguous float64 row (n <= 128). */
double orc_np_sum_f64(const double *a, int n)
{
    if (n < 8) {
        double res = -0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

float orc_np_sum_f32(const float *a, int n)
{
    if (n < 8) {
        float res = -0.0f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    float r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* np.exp on float32 as NumPy 2.2.6's SIMD kernel computes it. */
float orc_np_expf(float x)
{
    const float xmax = 88.72283935546875f, xmin = -103.97208404541015625f;
    if (x != x) return x;
    if (x >= xmax) return INFINITY;
    if (x <= xmin) return 0.0f;
    const float log2e = 0x1.715476p+0f, magic = 0x1.800000p+23f;
    float q = x * log2e;
    volatile float t = q + magic; /* round to nearest integer */
    q = t - magic;
    float r = fmaf(q, -0x1.62e400p-1f, x);
    r = fmaf(q, -0x1.7f7d1cp-20f, r);
    float num = fmaf(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
    num = fmaf(num, r, 5.114512081637298353406e-02f);
    num = fmaf(num, r, 2.473615434895520810817e-01f);
    num = fmaf(num, r, 7.257664613233124478488e-01f);
    num = fmaf(num, r, 9.999999999980870924916e-01f);
    float den = fmaf(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
    den = fmaf(den, r, 1.0f);
    return ldexpf(num / den, (int)q);
}

void orc_np_expf_array(const float *x, float *y, int64_t n)
{
    for (int64_t i = 0; i < n; i++) y[i] = orc_np_expf(x[i]);
}

/* utils/utils.py:18-20 on a float32 row: exp(x) / sum(exp(x)), no max-subtraction. */
static void softmax_f32(const float *x, float *out, int n)
{
    float e[ORC_MAX_A];
    for (int i = 0; i < n; i++) e[i] = orc_np_expf(x[i]);
    float s = orc_np_sum_f32(e, n);
    for (int i = 0; i < n; i++) out[i] = e[i] / s;
}

static void softmax_f64(const double *x, double *out, int n)
{
    double e[ORC_MAX_A];
    for (int i = 0; i < n; i++) e[i] = exp(x[i]);
    double s = orc_np_sum_f64(e, n);
    for (int i = 0; i < n; i++) out[i] = e[i] / s;
}

/* np.argmax: first maximal index; a NaN wins (first NaN). */
static int np_argmax(const double *x, int n)
{
    int best = 0;
    double bv = x[0];
    if (bv != bv) return 0;
    for (int i = 1; i < n; i++) {
        if (x[i] != x[i]) return i;
        if (x[i] > bv) { bv = x[i]; best = i; }
    }
    return best;
}

/* ---------------------------------------------------------------- counter noise */

static uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

double orc_noise_u(uint64_t seed, uint64_t tree, uint64_t call, uint32_t action)
{
    uint64_t key = splitmix64(seed ^ (tree * 0xD6E8FEB86659FD93ull));
    uint64_t z = splitmix64(key + call * 64 + action);
    return (double)(z >> 11) * 0x1p-53;
}

/* ---------------------------------------------------------------- hash network */

typedef struct {
    int32_t logit_shift, reward_on, parent_predicts, _pad;
} orc_hashnet;

static uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
static uint32_t hash2(uint32_t code, uint32_t k) { return fmix32(code + 0x9E3779B9u * (k + 1u)); }

void orc_hash_logits(uint32_t code, const orc_hashnet *p, int A, float *out)
{
    float scale = ldexpf(1.0f, -p->logit_shift);
    for (int a = 0; a < A; a++)
        out[a] = (float)((int)(hash2(code, 1 + a) % 257u) - 128) * scale;
}
double orc_hash_value(uint32_t code) { return ((int)(hash2(code, 200) % 513u) - 256) / 256.0; }
double orc_hash_reward(uint32_t code, const orc_hashnet *p)
{
    return p->reward_on ? (double)(hash2(code, 201) % 65u) / 64.0 : 0.0;
}
uint32_t orc_hash_child(uint32_t code, int action) { return hash2(code, 300 + action); }
void orc_hash_hidden(uint32_t code, int H, float *out)
{
    for (int i = 0; i < H; i++) {
        if (i == 0) out[i] = (float)(code & 0xFFFFu);
        else if (i == 1) out[i] = (float)(code >> 16);
        else out[i] = (float)(hash2(code, 1000 + i) & 0xFFu) / 256.0f;
    }
}

/* ---------------------------------------------------------------- tree helpers */

/* mcts/tree.py:55-58  children_rewards[n] + discount * children_values[n] */
static void qvalues(const orc_cfg *c, const orc_tree *t, int node, double *q)
{
    const double *r = t->children_rewards + (size_t)node * c->A;
    const double *v = t->children_values + (size_t)node * c->A;
    for (int a = 0; a < c->A; a++) q[a] = r[a] + c->discount * v[a];
}

/* mcts/qtransforms.py:7-22 */
void orc_qtransform_by_parent_and_siblings(const orc_cfg *c, const orc_tree *t, int node, double *out)
{
    int A = c->A;
    double q[ORC_MAX_A];
    qvalues(c, t, node, q);
    const int32_t *cv = t->children_visits + (size_t)node * A;
    double node_value = t->node_values[node];
    double mn = node_value, mx = node_value;
    for (int a = 0; a < A; a++) {
        double s = cv[a] > 0 ? q[a] : node_value;
        if (s < mn || s != s) mn = s;   /* np.min / np.max: a NaN wins and stays */
        if (s > mx || s != s) mx = s;
    }
    double den = mx - mn;
    if (den < 1e-8) den = 1e-8; /* np.maximum(max - min, epsilon): a NaN stays a NaN */
    for (int a = 0; a < A; a++) {
        double cq = cv[a] > 0 ? q[a] : mn;
        out[a] = (cq - mn) / den;
    }
}

/* mcts/qtransforms.py:51-77; returns the per-action VECTOR the reference produces. */
void orc_compute_mixed_value(int A, double raw_value, const double *q, const int32_t *cv,
                             const float *prior_logits, double *out)
{
    int64_t visit_sum = 0;
    for (int a = 0; a < A; a++) visit_sum += cv[a];
    float p[ORC_MAX_A], pm[ORC_MAX_A];
    double pq[ORC_MAX_A];
    softmax_f32(prior_logits, p, A);
    const float tiny = 1.17549435e-38f; /* np.finfo(np.float32).tiny */
    for (int a = 0; a < A; a++) {
        if (!(p[a] > tiny)) p[a] = (p[a] != p[a]) ? p[a] : tiny; /* np.maximum(tiny, p) */
        pm[a] = cv[a] > 0 ? p[a] : 0.0f;
        pq[a] = cv[a] > 0 ? (double)p[a] * q[a] : 0.0;
    }
    float prob_sum = orc_np_sum_f32(pm, A);
    double sum_pq = orc_np_sum_f64(pq, A);
    for (int a = 0; a < A; a++) {
        float den = cv[a] > 0 ? prob_sum : 1.0f;
        double wq = sum_pq / (double)den;
        out[a] = (raw_value + (double)visit_sum * wq) / (double)(visit_sum + 1);
    }
}

/* mcts/qtransforms.py:25-48 (prior logits are float32 on the Gumbel path) */
void orc_qtransform_completed_by_mix_value(const orc_cfg *c, const orc_tree *t, int node, double *out)
{
    int A = c->A;
    double q[ORC_MAX_A], mixed[ORC_MAX_A], comp[ORC_MAX_A];
    qvalues(c, t, node, q);
    const int32_t *cv = t->children_visits + (size_t)node * A;
    const float *pl = (const float *)t->children_prior_logits + (size_t)node * A;
    orc_compute_mixed_value(A, t->node_raw_values[node], q, cv, pl, mixed);
    double mn = 0, mx = 0;
    int32_t max_visit = cv[0];
    for (int a = 0; a < A; a++) {
        comp[a] = cv[a] > 0 ? q[a] : mixed[a];
        if (a == 0 || comp[a] < mn || comp[a] != comp[a]) mn = comp[a];   /* np.min / np.max: a NaN wins and stays */
        if (a == 0 || comp[a] > mx || comp[a] != comp[a]) mx = comp[a];
        if (cv[a] > max_visit) max_visit = cv[a];
    }
    double den = mx - mn;
    if (den < 1e-8) den = 1e-8;   /* np.maximum: a NaN stays a NaN */
    double scale = (c->c_visit + (double)max_visit) * c->c_scale;
    for (int a = 0; a < A; a++) out[a] = scale * ((comp[a] - mn) / den);
}

/* mcts/action_selection.py:160-175 */
void orc_get_considered_visits(int m, int S, int32_t *out)
{
    if (m <= 1) { for (int i = 0; i < S; i++) out[i] = i; return; }
    int log2max = (int)ceil(log2((double)m));
    int32_t visits[ORC_MAX_A] = {0};
    int n_considered = m, len = 0;
    while (len < S) {
        int n_extra = (int)((double)S / (double)(log2max * n_considered));
        if (n_extra < 1) n_extra = 1;
        for (int e = 0; e < n_extra; e++) {
            for (int i = 0; i < n_considered; i++) { if (len < S) out[len] = visits[i]; len++; }
            for (int i = 0; i < n_considered; i++) visits[i]++;
        }
        n_considered = n_considered / 2 > 2 ? n_considered / 2 : 2;
    }
}

/* mcts/action_selection.py:147-157 (logits float32, everything else float64) */
static void gumbel_score(int A, int32_t considered_visit, const double *g, const float *logits,
                         const double *nq, const int32_t *cv, double *out)
{
    float lmax = logits[0];
    for (int a = 1; a < A; a++) if (logits[a] > lmax) lmax = logits[a];
    for (int a = 0; a < A; a++) {
        float l = logits[a] - lmax;
        double s = (g[a] + (double)l) + nq[a];
        if (!(s > -1e9)) s = (s != s) ? s : -1e9; /* np.maximum(-1e9, s) */
        out[a] = s + (cv[a] == considered_visit ? 0.0 : -INFINITY);
    }
}

/* mcts/action_selection.py:141-144 */
static int masked_argmax(int A, double *x, const double *legal)
{
    if (legal) for (int a = 0; a < A; a++) if (legal[a] == 0) x[a] = -INFINITY;
    return np_argmax(x, A);
}

/* mcts/action_selection.py:25-49.  `uniform` is the np.random.uniform(size=A) draw. */
int orc_puct_select(const orc_cfg *c, const orc_tree *t, int node, int depth, const double *uniform)
{
    int A = c->A;
    int32_t nv = t->node_visits[node];
    const int32_t *cv = t->children_visits + (size_t)node * A;
    const double *pl = (const double *)t->children_prior_logits + (size_t)node * A;
    double cc = c->c_table ? c->c_table[nv]
                           : c->c_init + log(((double)nv + c->c_base + 1.0) / c->c_base);
    double q[ORC_MAX_A], s[ORC_MAX_A];
    orc_qtransform_by_parent_and_siblings(c, t, node, q);
    for (int a = 0; a < A; a++) {
        double u = cc * pl[a] + sqrt((double)nv / (double)(1 + cv[a]));
        s[a] = (u + q[a]) + 1e-7 * uniform[a];
    }
    return masked_argmax(A, s, depth == 0 ? t->root_legal : NULL);
}

/* mcts/action_selection.py:73-96 */
int orc_gumbel_root_select(const orc_cfg *c, const orc_tree *t, int node)
{
    int A = c->A;
    double cq[ORC_MAX_A], s[ORC_MAX_A];
    orc_qtransform_completed_by_mix_value(c, t, node, cq);
    const int32_t *cv = t->children_visits + (size_t)node * A;
    double n_legal = orc_np_sum_f64(t->root_legal, A);
    double mc = (double)c->max_considered < n_legal ? (double)c->max_considered : n_legal;
    int m = (int)mc;
    int32_t table[4096];
    orc_get_considered_visits(m, c->S, table);
    int64_t sim_index = 0;
    for (int a = 0; a < A; a++) sim_index += cv[a];
    int32_t considered = table[sim_index];
    const float *pl = (const float *)t->children_prior_logits + (size_t)node * A;
    gumbel_score(A, considered, t->gumbel_noise, pl, cq, cv, s);
    return masked_argmax(A, s, t->root_legal);
}

/* mcts/action_selection.py:99-118 */
int orc_gumbel_interior_select(const orc_cfg *c, const orc_tree *t, int node)
{
    int A = c->A;
    double cq[ORC_MAX_A], x[ORC_MAX_A], p[ORC_MAX_A];
    orc_qtransform_completed_by_mix_value(c, t, node, cq);
    const int32_t *cv = t->children_visits + (size_t)node * A;
    const float *pl = (const float *)t->children_prior_logits + (size_t)node * A;
    int64_t vs = 0;
    for (int a = 0; a < A; a++) { x[a] = (double)pl[a] + cq[a]; vs += cv[a]; }
    softmax_f64(x, p, A);
    for (int a = 0; a < A; a++) x[a] = p[a] - (double)cv[a] / (double)(1 + vs);
    return np_argmax(x, A);
}

/* mcts/mcts.py:98-114 with the switching wrapper of action_selection.py:11-22.
 * noise_mode 0: counter-based (seed, tree_id, running call index);
 * noise_mode 1: rows of `noise_buf` ([max_calls][A] uniforms) in call order. */
typedef struct {
    int32_t mode, max_calls;
    uint64_t seed, tree_id;
    const double *buf;
    int64_t calls; /* running count of PUCT selection calls in this search */
} orc_noise;

static int simulate(const orc_cfg *c, const orc_tree *t, orc_noise *nz, int *out_action,
                    int *out_to_play, int *out_depth)
{
    int next = ORC_ROOT, node = 0, action = 0, ntp = 0, depth = 0;
    while (next != ORC_UNVISITED) {
        node = next;
        if (c->gumbel) {
            action = depth == 0 ? orc_gumbel_root_select(c, t, node)
                                : orc_gumbel_interior_select(c, t, node);
        } else {
            double u[ORC_MAX_A];
            if (nz->mode == 0) {
                for (int a = 0; a < c->A; a++) u[a] = orc_noise_u(nz->seed, nz->tree_id, nz->calls, a);
            } else {
                for (int a = 0; a < c->A; a++)
                    u[a] = nz->calls < nz->max_calls ? nz->buf[nz->calls * c->A + a] : 0.0;
            }
            nz->calls++;
            action = orc_puct_select(c, t, node, depth, u);
        }
        next = t->children_index[(size_t)node * c->A + action];
        ntp = c->players == 1 ? ntp : 1 - ntp; /* utils/game_utils.py:61-63 */
        depth++;
    }
    *out_action = action; *out_to_play = ntp; *out_depth = depth;
    return node;
}

/* mcts/mcts.py:59-96: fresh arrays + root row. `root_prior` is already noised/masked. */
void orc_tree_init(const orc_cfg *c, orc_tree *t, const void *root_prior, double root_value,
                   const float *root_hidden, int root_to_play)
{
    int N = c->S + 1, A = c->A;
    for (int n = 0; n < N; n++) {
        t->node_visits[n] = 0; t->node_values[n] = 0; t->node_raw_values[n] = 0;
        t->parents[n] = -1; t->action_from_parent[n] = -1; t->to_play[n] = 0;
    }
    for (size_t i = 0; i < (size_t)N * A; i++) {
        t->children_index[i] = -1; t->children_visits[i] = 0;
        t->children_rewards[i] = 0; t->children_values[i] = 0;
        if (c->gumbel) ((float *)t->children_prior_logits)[i] = 0;
        else ((double *)t->children_prior_logits)[i] = 0;
    }
    if (t->hidden_states) memset(t->hidden_states, 0, sizeof(float) * (size_t)N * t->H);
    if (c->gumbel) memcpy(t->children_prior_logits, root_prior, sizeof(float) * A);
    else memcpy(t->children_prior_logits, root_prior, sizeof(double) * A);
    t->node_visits[0] = 1;
    t->node_values[0] = root_value; t->node_raw_values[0] = root_value;
    if (t->hidden_states && root_hidden) memcpy(t->hidden_states, root_hidden, sizeof(float) * t->H);
    t->to_play[0] = root_to_play;
}

/* mcts/mcts.py:116-141 (+ update_tree_node :37-57) */
void orc_expand(const orc_cfg *c, orc_tree *t, int node, int action, int leaf, const float *logits,
                const float *hidden, double reward, double value, int next_to_play)
{
    int A = c->A;
    for (int a = 0; a < A; a++) {
        if (c->gumbel) ((float *)t->children_prior_logits)[(size_t)leaf * A + a] = logits[a];
        else ((double *)t->children_prior_logits)[(size_t)leaf * A + a] = (double)logits[a];
    }
    t->node_visits[leaf] += 1;
    t->node_values[leaf] = value; t->node_raw_values[leaf] = value;
    if (t->hidden_states && hidden) memcpy(t->hidden_states + (size_t)leaf * t->H, hidden, sizeof(float) * t->H);
    t->to_play[leaf] = next_to_play;
    t->parents[leaf] = node; t->action_from_parent[leaf] = action;
    t->children_index[(size_t)node * A + action] = leaf;
    t->children_rewards[(size_t)node * A + action] = reward;
}

/* mcts/mcts.py:143-164 */
void orc_backward(const orc_cfg *c, orc_tree *t, int leaf, int to_play)
{
    int A = c->A;
    double leaf_value = t->node_values[leaf];
    while (leaf != ORC_ROOT) {
        int p = t->parents[leaf], a = t->action_from_parent[leaf];
        double reward = t->children_rewards[(size_t)p * A + a];
        int32_t pv = t->node_visits[p];
        double signed_v = t->to_play[p] == to_play ? leaf_value : -leaf_value;
        double parent_value = (t->node_values[p] * (double)pv + signed_v) / (double)(pv + 1);
        t->node_values[p] = parent_value;
        t->node_visits[p] = pv + 1;
        t->children_values[(size_t)p * A + a] = t->node_values[leaf];
        t->children_visits[(size_t)p * A + a] += 1;
        leaf_value = reward + c->discount * leaf_value;
        leaf = p;
    }
}

/* Network for orc_search(): mode 0 hash network, mode 1 table of recorded outputs per simulation. */
typedef struct {
    int32_t mode, _pad;
    orc_hashnet hash;
    const float *logits;   /* [S][A] */
    const double *value;   /* [S] */
    const double *reward;  /* [S] */
    const float *hidden;   /* [S][H] or NULL */
} orc_net;

/* The simulation loop of mcts/mcts.py:236-248.  Records (parent, action, depth) per simulation. */
void orc_search(const orc_cfg *c, orc_tree *t, const orc_net *net, orc_noise *nz,
                int32_t *sel_parent, int32_t *sel_action, int32_t *sel_depth)
{
    float logits[ORC_MAX_A];
    float *hid = t->H > 0 ? (float *)malloc(sizeof(float) * t->H) : NULL;
    for (int sim = 0; sim < c->S; sim++) {
        int action, ntp, depth;
        int node = simulate(c, t, nz, &action, &ntp, &depth);
        int leaf = sim + 1;
        double value, reward;
        const float *lg, *hd;
        if (net->mode == 0) {
            uint32_t pcode = ((uint32_t)t->hidden_states[(size_t)node * t->H]) |
                             ((uint32_t)t->hidden_states[(size_t)node * t->H + 1] << 16);
            uint32_t ccode = orc_hash_child(pcode, action);
            uint32_t src = net->hash.parent_predicts ? pcode : ccode;
            orc_hash_logits(src, &net->hash, c->A, logits);
            value = orc_hash_value(src);
            reward = orc_hash_reward(ccode, &net->hash);
            orc_hash_hidden(ccode, t->H, hid);
            lg = logits; hd = hid;
        } else {
            lg = net->logits + (size_t)sim * c->A;
            value = net->value[sim]; reward = net->reward[sim];
            hd = net->hidden ? net->hidden + (size_t)sim * t->H : NULL;
        }
        orc_expand(c, t, node, action, leaf, lg, hd, reward, value, ntp);
        orc_backward(c, t, leaf, ntp);
        if (sel_parent) sel_parent[sim] = node;
        if (sel_action) sel_action[sim] = action;
        if (sel_depth) sel_depth[sim] = depth;
    }
    free(hid);
}

/* ---------------------------------------------------------------- root preparation */

/* mcts/mcts.py:22-31 then utils/game_utils.py:71-78 (PUCT): f32 softmax, f64 mix, log, shift, mask. */
void orc_puct_root_prior(int A, const float *logits, const double *dirichlet, double frac,
                         const double *legal, double *out)
{
    float p[ORC_MAX_A];
    softmax_f32(logits, p, A);
    float keep = (float)(1.0 - frac);
    double mx = 0;
    for (int a = 0; a < A; a++) {
        double noised = (double)(keep * p[a]) + frac * dirichlet[a];
        const double tiny = 2.2250738585072014e-308;
        if (!(noised > tiny)) noised = (noised != noised) ? noised : tiny;
        out[a] = log(noised);
        if (a == 0 || out[a] > mx || out[a] != out[a]) mx = out[a];   /* np.max: a NaN wins and stays */
    }
    for (int a = 0; a < A; a++) out[a] = legal[a] != 0 ? out[a] - mx : -1.7976931348623157e308;
}

/* utils/game_utils.py:71-78 on the raw float32 logits (Gumbel). */
void orc_gumbel_root_prior(int A, const float *logits, const double *legal, float *out)
{
    float mx = logits[0];
    for (int a = 1; a < A; a++) if (logits[a] > mx) mx = logits[a];
    for (int a = 0; a < A; a++) out[a] = legal[a] != 0 ? logits[a] - mx : -3.4028234663852886e38f;
}

/* ---------------------------------------------------------------- action to act */

/* mcts/tree.py:60-68 */
static void summary_probs(int A, const int32_t *counts, double *probs)
{
    int64_t total = 0;
    for (int a = 0; a < A; a++) total += counts[a];
    int64_t den = total > 1 ? total : 1;
    for (int a = 0; a < A; a++)
        probs[a] = total > 0 ? (double)counts[a] / (double)den : 1.0 / (double)A;
}

/* mcts/action_selection.py:52-70; `u` replaces the uniform np.random.choice draws. */
int orc_puct_to_act(const orc_cfg *c, const orc_tree *t, double temperature, double u,
                    double *probs_out, double *q_out)
{
    int A = c->A;
    double vp[ORC_MAX_A], lg[ORC_MAX_A], cdf[ORC_MAX_A];
    summary_probs(A, t->children_visits, vp);
    const double tiny = 2.2250738585072014e-308;
    double mx = 0;
    for (int a = 0; a < A; a++) {
        lg[a] = log(vp[a] > tiny ? vp[a] : tiny);
        if (a == 0 || lg[a] > mx) mx = lg[a];
    }
    double tt = temperature > 1e-27 ? temperature : 1e-27;
    for (int a = 0; a < A; a++) lg[a] = (lg[a] - mx) / tt;
    softmax_f64(lg, probs_out, A);
    if (q_out) qvalues(c, t, ORC_ROOT, q_out);
    /* legacy np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right') */
    double acc = 0;
    for (int a = 0; a < A; a++) { acc += probs_out[a]; cdf[a] = acc; }
    int idx = 0;
    for (int a = 0; a < A; a++) { cdf[a] /= acc; if (cdf[a] <= u) idx = a + 1; }
    return idx < A ? idx : A - 1;
}

/* mcts/action_selection.py:121-138 */
int orc_gumbel_to_act(const orc_cfg *c, const orc_tree *t, double *probs_out, double *cq_out)
{
    int A = c->A;
    const int32_t *cv = t->children_visits;
    int32_t considered = cv[0];
    for (int a = 1; a < A; a++) if (cv[a] > considered) considered = cv[a];
    double cq[ORC_MAX_A], s[ORC_MAX_A], x[ORC_MAX_A];
    orc_qtransform_completed_by_mix_value(c, t, ORC_ROOT, cq);
    const float *pl = (const float *)t->children_prior_logits;
    gumbel_score(A, considered, t->gumbel_noise, pl, cq, cv, s);
    int action = masked_argmax(A, s, t->root_legal);
    double mx = 0;
    for (int a = 0; a < A; a++) { x[a] = (double)pl[a] + cq[a]; if (a == 0 || x[a] > mx) mx = x[a]; }
    for (int a = 0; a < A; a++) x[a] = t->root_legal[a] != 0 ? x[a] - mx : -1.7976931348623157e308;
    softmax_f64(x, probs_out, A);
    if (cq_out) memcpy(cq_out, cq, sizeof(double) * A);
    return action;
}

/* ---------------------------------------------------------------- batched driver (CPU baseline) */

/*
 * B independent searches with the hash network, one tree per OpenMP thread.
 * Roots: code root_codes[b], legal[b][A], to_play[b]; PUCT takes dirichlet[b][A]
 * (+frac), Gumbel takes gumbel[b][A].  Noise is counter-based with tree id
 * tree_id0 + b.  Outputs per tree: root children_visits, node_values[0],
 * sum(node_visits) - 1, the Gumbel action / the PUCT argmax-of-visits action.
 */
void orc_search_batch_hash(const orc_cfg *c, const orc_hashnet *hn, int H, int64_t B,
                           const uint32_t *root_codes, const double *legal, const int32_t *to_play,
                           const double *dirichlet, double frac, const double *gumbel,
                           uint64_t seed, uint64_t tree_id0, int32_t *out_root_visits,
                           double *out_root_value, int64_t *out_node_visits, int32_t *out_action,
                           int n_threads)
{
    int N = c->S + 1, A = c->A;
#pragma omp parallel num_threads(n_threads)
    {
        orc_tree t;
        memset(&t, 0, sizeof t);
        t.H = H;
        t.node_visits = malloc(sizeof(int32_t) * N); t.node_values = malloc(sizeof(double) * N);
        t.node_raw_values = malloc(sizeof(double) * N); t.parents = malloc(sizeof(int32_t) * N);
        t.action_from_parent = malloc(sizeof(int32_t) * N); t.to_play = malloc(sizeof(int32_t) * N);
        t.children_index = malloc(sizeof(int32_t) * N * A);
        t.children_prior_logits = malloc(sizeof(double) * N * A);
        t.children_visits = malloc(sizeof(int32_t) * N * A);
        t.children_rewards = malloc(sizeof(double) * N * A);
        t.children_values = malloc(sizeof(double) * N * A);
        t.hidden_states = malloc(sizeof(float) * (size_t)N * H);
        float *rh = malloc(sizeof(float) * H);
#pragma omp for schedule(dynamic, 16)
        for (int64_t b = 0; b < B; b++) {
            float lg[ORC_MAX_A], pf[ORC_MAX_A];
            double pd[ORC_MAX_A], probs[ORC_MAX_A];
            orc_hash_logits(root_codes[b], hn, A, lg);
            orc_hash_hidden(root_codes[b], H, rh);
            t.root_legal = legal + b * A;
            t.gumbel_noise = c->gumbel ? gumbel + b * A : NULL;
            if (c->gumbel) {
                orc_gumbel_root_prior(A, lg, t.root_legal, pf);
                orc_tree_init(c, &t, pf, orc_hash_value(root_codes[b]), rh, to_play[b]);
            } else {
                orc_puct_root_prior(A, lg, dirichlet + b * A, frac, t.root_legal, pd);
                orc_tree_init(c, &t, pd, orc_hash_value(root_codes[b]), rh, to_play[b]);
            }
            orc_net net; memset(&net, 0, sizeof net); net.mode = 0; net.hash = *hn;
            orc_noise nz; memset(&nz, 0, sizeof nz); nz.mode = 0; nz.seed = seed; nz.tree_id = tree_id0 + b;
            orc_search(c, &t, &net, &nz, NULL, NULL, NULL);
            int64_t nvs = 0;
            for (int n = 0; n < N; n++) nvs += t.node_visits[n];
            out_node_visits[b] = nvs - 1;
            out_root_value[b] = t.node_values[0];
            for (int a = 0; a < A; a++) out_root_visits[b * A + a] = t.children_visits[a];
            if (c->gumbel) out_action[b] = orc_gumbel_to_act(c, &t, probs, NULL);
            else out_action[b] = orc_puct_to_act(c, &t, 0.0, 0.0, probs, NULL);
        }
        free(t.node_visits); free(t.node_values); free(t.node_raw_values); free(t.parents);
        free(t.action_from_parent); free(t.to_play); free(t.children_index);
        free(t.children_prior_logits); free(t.children_visits); free(t.children_rewards);
        free(t.children_values); free(t.hidden_states); free(rh);
    }
}

/* ---------------------------------------------------------------- network leaf evaluation
 * A lowered MuZero network is a "dense program" (muzero_amd/packing.py documents the flat
 * layout): levels of rows, each row = one output neuron  act(bias + sum_k W4[k/4][row][k%4] * x[in+k]
 * (+ skip)).  For the reference's fully connected network (network.py:405-479,
 * utils/network_utils.py:8-22: Linear + ELU stacks) this is exactly its forward pass; plain
 * float32 multiply-adds in k order.  Compared against the reference's own outputs at 1e-5. */
void orc_dense_run(const int32_t *meta, const float *data, float *scratch)
{
    int n_levels = meta[0];
    const int32_t *lv = meta + meta[5], *rows = meta + meta[6];
    const float *bias = data + meta[4];
    for (int l = 0; l < n_levels; l++) {
        int n_rows = lv[4 * l], kmax = lv[4 * l + 1], w_off = lv[4 * l + 2], row0 = lv[4 * l + 3];
        const float *WT = data + w_off;
        for (int r = 0; r < n_rows; r++) {
            const int32_t *d = rows + 4 * (row0 + r);
            float acc = bias[row0 + r];
            for (int k = 0; k < kmax; k++)
                acc += WT[((size_t)(k / 4) * n_rows + r) * 4 + (k % 4)] * scratch[d[0] + k];
            if (d[3] >= 0) acc += scratch[d[3]];
            if (d[2] == 1) acc = acc > 0 ? acc : 0;
            else if (d[2] == 2) acc = acc > 0 ? acc : expm1f(acc);
            scratch[d[1]] = acc;
        }
    }
}

/* utils/network_utils.py:132-158: (s - min) / (max - min + 1e-5) per group, float32. */
void orc_minmax_norm(const float *in, float *out, int n, int groups)
{
    int per = n / groups;
    for (int g = 0; g < groups; g++) {
        float mn = in[g * per], mx = in[g * per];
        for (int i = 1; i < per; i++) {
            float v = in[g * per + i];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        float den = (mx - mn) + 1e-5f;
        for (int i = 0; i < per; i++) out[g * per + i] = (in[g * per + i] - mn) / den;
    }
}

/* utils/network_utils.py:75-96: float32 softmax . arange(-L..L) in float64, inverse value transform. */
double orc_support_to_scalar(const float *logits, int n)
{
    int L = (n - 1) / 2;
    float mx = logits[0];
    for (int i = 1; i < n; i++) if (logits[i] > mx) mx = logits[i];
    float e[1024], s = 0;
    for (int i = 0; i < n; i++) { e[i] = expf(logits[i] - mx); s += e[i]; }
    double x = 0;
    for (int i = 0; i < n; i++) x += (double)(e[i] / s) * (double)(i - L);
    const double eps = 0.001;
    double sg = x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0);
    double t = (sqrt(1 + 4 * eps * (fabs(x) + 1 + eps)) - 1) / (2 * eps);
    return sg * (t * t - 1);
}

/* io = PackedNetwork.io(): {in_hidden, in_action, out_next, out_reward, out_policy, out_value, H, A,
 * action_size, support, norm_groups, obs_size}.  recurrent_inference of network.py:105-126. */
void orc_net_recurrent(const int32_t *meta, const float *data, const int32_t *io, const float *hidden,
                       int action, float *scratch, float *logits_out, float *next_hidden_out,
                       double *value_out, double *reward_out)
{
    int H = io[6], A = io[7], asz = io[8], sup = io[9];
    memset(scratch, 0, sizeof(float) * meta[1]);
    memcpy(scratch + io[0], hidden, sizeof(float) * H);
    for (int i = 0; i < asz; i++) scratch[io[1] + i] = i == action ? 1.0f : 0.0f;
    orc_dense_run(meta, data, scratch);
    orc_minmax_norm(scratch + io[2], next_hidden_out, H, io[10]);
    memcpy(logits_out, scratch + io[4], sizeof(float) * A);
    *value_out = orc_support_to_scalar(scratch + io[5], sup);
    *reward_out = orc_support_to_scalar(scratch + io[3], sup);
}

void orc_net_recurrent_batch(const int32_t *meta, const float *data, const int32_t *io, int64_t B,
                             const float *hidden, const int32_t *action, float *logits_out,
                             float *next_hidden_out, double *value_out, double *reward_out)
{
    int H = io[6], A = io[7];
    float *scratch = malloc(sizeof(float) * meta[1]);
    for (int64_t b = 0; b < B; b++)
        orc_net_recurrent(meta, data, io, hidden + b * H, action[b], scratch, logits_out + b * A,
                          next_hidden_out + b * H, value_out + b, reward_out + b);
    free(scratch);
}

/*
 * B independent searches with a packed network as the leaf evaluator (the CPU baseline of the
 * benchmark): roots are given already evaluated (root_logits f32[B][A], root_hidden f32[B][H],
 * root_value f64[B]) -- the one initial inference per search is outside the loop this times.
 */
void orc_search_batch_net(const orc_cfg *c, const int32_t *meta, const float *data, const int32_t *io,
                          int64_t B, const float *root_logits, const float *root_hidden,
                          const double *root_value, const double *legal, const int32_t *to_play,
                          const double *dirichlet, double frac, const double *gumbel, uint64_t seed,
                          uint64_t tree_id0, int32_t *out_root_visits, double *out_root_value,
                          int64_t *out_node_visits, int32_t *out_action, int n_threads)
{
    int N = c->S + 1, A = c->A, H = io[6];
#pragma omp parallel num_threads(n_threads)
    {
        orc_tree t;
        memset(&t, 0, sizeof t);
        t.H = H;
        t.node_visits = malloc(sizeof(int32_t) * N); t.node_values = malloc(sizeof(double) * N);
        t.node_raw_values = malloc(sizeof(double) * N); t.parents = malloc(sizeof(int32_t) * N);
        t.action_from_parent = malloc(sizeof(int32_t) * N); t.to_play = malloc(sizeof(int32_t) * N);
        t.children_index = malloc(sizeof(int32_t) * N * A);
        t.children_prior_logits = malloc(sizeof(double) * N * A);
        t.children_visits = malloc(sizeof(int32_t) * N * A);
        t.children_rewards = malloc(sizeof(double) * N * A);
        t.children_values = malloc(sizeof(double) * N * A);
        t.hidden_states = malloc(sizeof(float) * (size_t)N * H);
        float *scratch = malloc(sizeof(float) * meta[1]);
        float *nh = malloc(sizeof(float) * H);
#pragma omp for schedule(dynamic, 16)
        for (int64_t b = 0; b < B; b++) {
            float pf[ORC_MAX_A], lg[ORC_MAX_A];
            double pd[ORC_MAX_A], probs[ORC_MAX_A];
            t.root_legal = legal + b * A;
            t.gumbel_noise = c->gumbel ? gumbel + b * A : NULL;
            if (c->gumbel) {
                orc_gumbel_root_prior(A, root_logits + b * A, t.root_legal, pf);
                orc_tree_init(c, &t, pf, root_value[b], root_hidden + b * H, to_play[b]);
            } else {
                orc_puct_root_prior(A, root_logits + b * A, dirichlet + b * A, frac, t.root_legal, pd);
                orc_tree_init(c, &t, pd, root_value[b], root_hidden + b * H, to_play[b]);
            }
            orc_noise nz; memset(&nz, 0, sizeof nz); nz.seed = seed; nz.tree_id = tree_id0 + b;
            for (int sim = 0; sim < c->S; sim++) {
                int action, ntp, depth;
                int node = simulate(c, &t, &nz, &action, &ntp, &depth);
                double value, reward;
                orc_net_recurrent(meta, data, io, t.hidden_states + (size_t)node * H, action, scratch, lg, nh,
                                  &value, &reward);
                orc_expand(c, &t, node, action, sim + 1, lg, nh, reward, value, ntp);
                orc_backward(c, &t, sim + 1, ntp);
            }
            int64_t nvs = 0;
            for (int n = 0; n < N; n++) nvs += t.node_visits[n];
            out_node_visits[b] = nvs - 1;
            out_root_value[b] = t.node_values[0];
            for (int a = 0; a < A; a++) out_root_visits[b * A + a] = t.children_visits[a];
            if (c->gumbel) out_action[b] = orc_gumbel_to_act(c, &t, probs, NULL);
            else out_action[b] = orc_puct_to_act(c, &t, 0.0, 0.0, probs, NULL);
        }
        free(t.node_visits); free(t.node_values); free(t.node_raw_values); free(t.parents);
        free(t.action_from_parent); free(t.to_play); free(t.children_index);
        free(t.children_prior_logits); free(t.children_visits); free(t.children_rewards);
        free(t.children_values); free(t.hidden_states); free(scratch); free(nh);
    }
}

/* ---------------------------------------------------------------- environments
 * TicTacToe: games/tictactoe.py:14-81 + games/game.py:56-74.  board[9] in {-1 (X), 0, +1 (O)},
 * *mover = -1/+1 is the player about to move.  Returns terminated; *reward as step() returns it;
 * obs[27] = planes [board == -1, board == +1, full(next mover)]; legal[9] = board == 0.
 * `winner` persists inside a game exactly like the attribute: -2 stands for None. */
int orc_tictactoe_step(double *board, int *mover, int *winner, int action, double *reward, double *obs,
                       double *legal)
{
    board[action] = (double)*mover;      /* no occupancy check (games/tictactoe.py:54) */
    *mover = -*mover;
    int terminated = 0;
    double sums[8];
    for (int c = 0; c < 3; c++) sums[c] = board[c] + board[3 + c] + board[6 + c];          /* columns */
    for (int r = 0; r < 3; r++) sums[3 + r] = board[3 * r] + board[3 * r + 1] + board[3 * r + 2]; /* rows */
    sums[6] = board[0] + board[4] + board[8];
    sums[7] = board[2] + board[4] + board[6];
    for (int i = 0; i < 8; i++)
        if (fabs(sums[i]) == 3) { terminated = 1; *winner = (int)(sums[i] / 3); break; }
    double filled = 0;
    for (int i = 0; i < 9; i++) filled += fabs(board[i]);
    if (filled == 9) { terminated = 1; *winner = 0; }   /* a board-filling win is scored as a tie */
    *reward = (*winner != -2 && *winner != 0) ? 1.0 : 0.0;
    for (int i = 0; i < 9; i++) {
        obs[i] = board[i] == -1 ? 1.0 : 0.0;
        obs[9 + i] = board[i] == 1 ? 1.0 : 0.0;
        obs[18 + i] = (double)*mover;
        legal[i] = board[i] == 0 ? 1.0 : 0.0;
    }
    return terminated;
}

/* CartPole-v1 as the public gymnasium specification states it (games/cartpole.py:9-48 delegates to
 * gymnasium, which the reference does not pin: this is a SPEC restatement, parity unpinned).
 * Euler integration, tau 0.02, g 9.8, cart 1.0, pole 0.1, half-length 0.5, force 10; terminated when
 * |x| > 2.4 or |theta| > 12 degrees; reward 1 per step including the terminating one.  state f64[4]. */
int orc_cartpole_step(double *s, int action, double *reward, float *obs)
{
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, total_mass = masspole + masscart, length = 0.5;
    const double polemass_length = masspole * length, force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12 * 2 * 3.141592653589793 / 360, x_thr = 2.4;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = action == 1 ? force_mag : -force_mag;
    double costheta = cos(theta), sintheta = sin(theta);
    double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
    double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot; x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot; theta_dot = theta_dot + tau * thetaacc;
    s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
    for (int i = 0; i < 4; i++) obs[i] = (float)s[i];
    *reward = 1.0;
    return x < -x_thr || x > x_thr || theta < -theta_thr || theta > theta_thr;
}

/* ------------------------------------------------------------------------------------------------
 * Training targets of one sampled position (GameHistory.make_target, games/game.py:185-253), on the
 * game's own lists: rewards[T], to_plays[T], root_values[T] (the reanalysed ones when the game has
 * them, :223), probs[T][A] (likewise, :245).  gamma_pow[k] = gamma**k as Python evaluates it
 * (k <= td_steps).  Outputs are the float64 lists the reference returns, value/reward [U+1],
 * policy [U+1][A].  Python's sum() starts from 0 and adds left to right. */
void orc_make_target(int T, int A, const double *rewards, const int32_t *to_plays, const double *root_values,
                     const double *probs, int t, int td_steps, double gamma, const double *gamma_pow,
                     int unroll_steps, double *value_out, double *reward_out, double *policy_out)
{
    for (int step = t, j = 0; step <= t + unroll_steps; step++, j++) {
        double value = 0.0;
        if (gamma == 1.0) {                                    /* :214-220 */
            for (int i = step; i < T; i++)
                value += (to_plays[i] == to_plays[step]) ? rewards[i] : -rewards[i];
        } else {                                               /* :221-235 */
            int bootstrap_step = step + td_steps;
            double bootstrap = 0.0;
            if (bootstrap_step < T) {
                bootstrap = (to_plays[bootstrap_step] == to_plays[step]) ? root_values[bootstrap_step]
                                                                         : -root_values[bootstrap_step];
                bootstrap *= gamma_pow[td_steps];
            }
            int k = 0;                                         /* enumerate(rewards[step+1 : bootstrap_step+1]) */
            for (int i = step + 1; i < bootstrap_step + 1 && i < T; i++, k++) {
                double r = (to_plays[step + k] == to_plays[step]) ? rewards[i] : -rewards[i];
                value += r * gamma_pow[k];
            }
            value = value + bootstrap;
        }
        if (step < T) {                                        /* :241-247 */
            value_out[j] = value;
            reward_out[j] = rewards[step];
            for (int a = 0; a < A; a++) policy_out[j * A + a] = probs[step * A + a];
        } else {                                               /* :248-251 */
            value_out[j] = 0.0;
            reward_out[j] = 0.0;
            for (int a = 0; a < A; a++) policy_out[j * A + a] = 1.0 / (double)A;
        }
    }
}

/* GameHistory.stack_n_observations for a non-empty history (games/game.py:143-160): newest first,
 * optional action planes a / n_actions, zero planes when fewer than n observations exist.
 * observations: [T][obs] float64; out: [n * (1 + stack_action)][obs]. */
void orc_stack_observations(int T, int obs, const double *observations, const int32_t *actions, int t, int n,
                            int n_actions, int stack_action, double *out)
{
    t = ((t % T) + T) % T;
    int n_ = n < t + 1 ? n : t + 1, row = 0;
    for (int step = t; step >= t - n_ + 1; step--) {
        for (int i = 0; i < obs; i++) out[row * obs + i] = observations[step * obs + i];
        row++;
        if (stack_action) {
            for (int i = 0; i < obs; i++) out[row * obs + i] = (double)actions[step] / (double)n_actions;
            row++;
        }
    }
    for (int r = 0; r < (n - n_) * (stack_action ? 2 : 1); r++, row++)
        for (int i = 0; i < obs; i++) out[row * obs + i] = 0.0;
}

/* scalar_to_support with the default transform (utils/network_utils.py:70-71, 99-129), float32 like the
 * torch expressions: out[i][0 .. 2L] for each of the n scalars. */
void orc_scalar_to_support(const float *x, int64_t n, int L, float *out)
{
    const int bins = 2 * L + 1;
    for (int64_t i = 0; i < n; i++) {
        float v = x[i];
        float sg = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
        float root = sqrtf(fabsf(v) + 1.0f) - 1.0f;
        float scaled = sg * root;
        float eps_x = 0.001f * v;
        float h = scaled + eps_x;
        if (h < (float)-L) h = (float)-L;
        if (h > (float)L) h = (float)L;
        float fl = floorf(h);
        float prob = h - fl;
        int lo = (int)fl + L, hi = lo + 1;
        for (int j = 0; j < bins; j++) out[i * bins + j] = 0.0f;
        out[i * bins + lo] = 1.0f - prob;                    /* first scatter_ */
        if (hi > 2 * L) { prob = 0.0f; hi = 0; }             /* masked_fill_ of both tensors */
        out[i * bins + hi] = prob;                           /* second scatter_ */
    }
}
