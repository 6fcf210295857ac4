// C ABI of the batched search engine (gfx950): contexts, argument checks and kernel launches.
// See include/muzero_amd.h; the kernels live in the headers included below.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>
#include <unistd.h>
#include <vector>

#include <map>
#include <mutex>
#include <utility>
#include "../../include/muzero_amd.h"
#include "tree_kernels.h"  // device_common.h, tree_ops.h
#include "dense_net.h"
#include "mfma_launch.h"   // matrix-core leaf evaluation and searches, compiled in mfma_kernels.hip
#include "search_net.h"
#include "search_reg.h"
#include "env_step.h"
#include "targets.h"

using namespace mz;


// ============================================================================ host side

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail(MZ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct mz_ctx {
    mz_config cfg;
    Dev d;
    int lanes, rl;    // lanes per tree (group size) and action lanes (smallest power of two >= A, >= 2)
    int n_expanded;   // nodes 1..n_expanded exist (simulations done since reset)
    int selected;     // 1 between mz_select and mz_expand_backup
    int B;            // batch of the last reset
    std::vector<void *> allocs;
    double *node_tmp;   // device scratch for per-node ops [A]
    int32_t *node_act;  // device scratch
    NetDev net;         // uploaded dense-program network (mz_set_network)
    int net_scratch;    // activation floats per tree
    bool net_small;     // the program fits the register-cached evaluator and the whole-wave epilogue
    bool net_tiny;      // ... and its hidden states and second-level outputs can live in registers (search_reg.h)
    void *net_meta_d, *net_data_d;
    RootNetDev root;    // representation + prediction programs (mz_set_root_network)
    void *root_bufs[4];
    MfmaNet mnet;       // MFMA-layout network (mz_set_network_mfma)
    void *mnet_bufs[10];  // steps, per-wave schedules, (unused), wfrag, bias, (unused), global activations, ctinfo, kidx, block descriptors of the level kernel
    int mnet_max_k16;                  // widest layer input in blocks of 16 floats
    std::vector<int> mnet_level_job0;  // first block of each level in the job table (+ total at the end)
    bool mnet_tile_fits;               // a 16-tree activation tile fits in LDS (tile-resident kernels usable)
    // host copies of the uploaded program's tables, from which the chained search's work lists are made, and that
    // search's device buffers: segment table + ranges (for `chain_G` workgroups per cluster), tagged activations
    std::vector<int32_t> mnet_steps_h, mnet_ctinfo_h, mnet_kidx_h;
    int chain_G, chain_n_seg, chain_kk_max, chain_ok;   // chain_ok: -1 not examined, 0 not servable, 1 servable
    int chain_in_n, chain_out_phase[4];
    std::string chain_why;                              // why the chained search does not serve this network
    void *chain_tab, *chain_acts2;
    size_t chain_acts2_bytes;
    float *leaf_logits; // [max_batch][A] outputs of the batched leaf evaluation
    double *leaf_value, *leaf_reward;
    // representation and prediction programs of the initial inference in matrix-core layout (mz_set_root_network_mfma)
    struct RootMfma {
        MfmaNet mnet;
        void *bufs[10];
        int max_k16;
        std::vector<int> level_job0;
    } rootm[2];
};
using RootMfma = mz_ctx::RootMfma;

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }
static bool shape_supported(int lanes, int rl) { return lanes == rl || (lanes == 16 && rl < 16) || (lanes == 64 && rl < 64); }

static void considered_visits_host(int m, int S, int32_t *out)
{
    if (m <= 1) { for (int i = 0; i < S; i++) out[i] = i; return; }
    int log2max = (int)std::ceil(std::log2((double)m));
    std::vector<int32_t> visits(m, 0);
    int n_considered = m, len = 0;
    while (len < S) {
        int n_extra = (int)((double)S / (double)(log2max * n_considered));
        if (n_extra < 1) n_extra = 1;
        for (int e = 0; e < n_extra && len < S; e++) {
            for (int i = 0; i < n_considered && len < S; i++) out[len++] = visits[i];
            for (int i = 0; i < n_considered; i++) visits[i]++;
        }
        n_considered = n_considered / 2 > 2 ? n_considered / 2 : 2;
    }
}

extern "C" int mz_considered_visits(int32_t m, int32_t S, int32_t *out_h)
{
    if (S < 0 || m > MZ_MAX_ACTIONS || !out_h) return fail(MZ_ERR_ARG, "mz_considered_visits: bad argument");
    considered_visits_host(m, S, out_h);
    return MZ_OK;
}

// NumPy's legacy generator (numpy/random/src/mt19937/mt19937.c, src/legacy/legacy-distributions.c, 2.2.6), restated
// for the one sequence the drop-in draws per tree: a Dirichlet row with one shape below 1, then optionally a uniform.
namespace {
struct LegacyMT {
    uint32_t *key;
    int pos;
    void refill()
    {
        const int N = 624, M = 397;
        const uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
        int kk;
        uint32_t y;
        for (kk = 0; kk < N - M; kk++) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + M] ^ (y >> 1) ^ (-(y & 1) & MATRIX_A);
        }
        for (; kk < N - 1; kk++) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (M - N)] ^ (y >> 1) ^ (-(y & 1) & MATRIX_A);
        }
        y = (key[N - 1] & UPPER) | (key[0] & LOWER);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(y & 1) & MATRIX_A);
        pos = 0;
    }
    uint32_t next32()
    {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double()
    {
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    double standard_exponential() { return -std::log(1.0 - next_double()); }
    double standard_gamma_below_one(double shape)
    {
        for (;;) {
            const double U = next_double(), V = standard_exponential();
            if (U <= 1.0 - shape) {
                const double X = std::pow(U, 1. / shape);
                if (X <= V) return X;
            } else {
                const double Y = -std::log((1 - U) / shape);
                const double X = std::pow(1.0 - shape + shape * Y, 1. / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
};
}  // namespace

extern "C" int mz_legacy_root_noise(uint32_t *mt_key, int32_t *mt_pos, double alpha, int32_t A, int32_t B,
                                    int32_t with_uniform, double *out_dirichlet, double *out_u)
{
    if (!mt_key || !mt_pos || !out_dirichlet || A < 1 || B < 0 || *mt_pos < 0 || *mt_pos > 624 || (with_uniform && !out_u))
        return fail(MZ_ERR_ARG, "mz_legacy_root_noise: bad argument");
    if (!(alpha > 0.0 && alpha < 1.0)) return fail(MZ_ERR_ARG, "mz_legacy_root_noise: alpha must lie in (0, 1)");
    LegacyMT mt{mt_key, *mt_pos};
    for (int b = 0; b < B; b++) {
        double *row = out_dirichlet + (size_t)b * A, acc = 0.0;
        for (int j = 0; j < A; j++) { row[j] = mt.standard_gamma_below_one(alpha); acc = acc + row[j]; }
        const double invacc = 1 / acc;
        for (int j = 0; j < A; j++) row[j] = row[j] * invacc;
        if (with_uniform) out_u[b] = mt.next_double();
    }
    *mt_pos = mt.pos;
    return MZ_OK;
}

// np.random.gumbel(size=n) of the legacy generator (numpy/random/src/legacy/legacy-distributions.c legacy_gumbel with
// loc 0, scale 1): u = 1 - next_double(), value = -log(-log(u)), one uniform per value in order.  The uniforms come off
// the Mersenne Twister one after the other; the two logarithms per value - nearly all of the time - are taken by a few
// host threads side by side, each with the C library's log on its own stretch, so every value has the bits NumPy gives.
// Helper threads of the host-side draws, kept for the life of the process: starting sixteen threads costs as much as
// the draws of a 16 384-tree batch themselves.  start() hands fn(w) to helpers w = 0 .. n - 1 and returns, wait() returns
// when all of them have; one job at a time (`call`, held by the caller from start() to wait()).  A process forked from
// one that owns a pool has none of its threads: the pool is then made again (host_pool()).
class HostPool {
public:
    explicit HostPool(int n)
    {
        for (int w = 0; w < n; w++) threads_.emplace_back([this, w]() { work(w); });
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_work_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    int size() const { return (int)threads_.size(); }
    void start(int n, std::function<void(int)> fn)
    {
        { std::lock_guard<std::mutex> lk(m_); job_ = std::move(fn); active_ = std::min(n, size()); pending_ = active_; epoch_++; }
        cv_work_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [this]() { return pending_ == 0; });
    }
    std::mutex call;
private:
    void work(int w)
    {
        std::unique_lock<std::mutex> lk(m_);
        unsigned long long seen = 0;
        for (;;) {
            cv_work_.wait(lk, [&]() { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
            if (w >= active_) continue;
            lk.unlock();
            job_(w);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::function<void(int)> job_;
    unsigned long long epoch_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};

static HostPool &host_pool()
{
    static std::mutex m;
    static std::unique_ptr<HostPool> pool;
    static pid_t owner = 0;
    std::lock_guard<std::mutex> lk(m);
    if (pool && owner != getpid()) (void)pool.release();   // a forked child: the threads stayed with the parent
    if (!pool) {   // of one size for good: a caller may be inside a job while another asks for the pool
        const unsigned hw = std::thread::hardware_concurrency();
        pool.reset(new HostPool(std::max(1, (int)std::min<unsigned>(hw ? hw : 1u, 16u) - 1)));
        owner = getpid();
    }
    return *pool;
}

// The next n doubles of the stream (random_double: 27 + 26 bits of two consecutive words), each as 1 - u, in bulk: the words
// of a block are tempered in one loop and paired in another - both loops without a branch, unlike a call per value that
// asks "block used up?" twice - and `progress` (may be null) is told how many values stand in `out` after every block.
static void legacy_one_minus_doubles(LegacyMT &mt, double *out, int64_t n, std::atomic<int64_t> *progress)
{
    uint32_t tw[624];
    uint32_t first = 0;
    bool have_first = false;
    int64_t i = 0;
    while (i < n) {
        if (mt.pos == 624) mt.refill();
        const int64_t need = 2 * (n - i) - (have_first ? 1 : 0);
        const int take = (int)std::min<int64_t>(624 - mt.pos, need);
        const uint32_t *src = mt.key + mt.pos;
        for (int k = 0; k < take; k++) {
            uint32_t y = src[k];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            tw[k] = y;
        }
        mt.pos += take;
        auto value = [](uint32_t a, uint32_t b) {
            return 1.0 - ((double)(int32_t)(a >> 5) * 67108864.0 + (double)(int32_t)(b >> 6)) / 9007199254740992.0;
        };
        int k = 0;
        if (have_first) { out[i++] = value(first, tw[0]); k = 1; have_first = false; }
        const int pairs = (take - k) / 2;
        for (int j = 0; j < pairs; j++) out[i + j] = value(tw[k + 2 * j], tw[k + 2 * j + 1]);
        i += pairs;
        k += 2 * pairs;
        if (k < take) { first = tw[k]; have_first = true; }
        if (progress) progress->store(i, std::memory_order_release);
    }
}

extern "C" int mz_legacy_gumbel(uint32_t *mt_key, int32_t *mt_pos, int64_t n, double *out)
{
    if (!mt_key || !mt_pos || n < 0 || (n > 0 && !out) || *mt_pos < 0 || *mt_pos > 624)
        return fail(MZ_ERR_ARG, "mz_legacy_gumbel: bad argument");
    LegacyMT mt{mt_key, *mt_pos};
    // The generator is serial, the logarithms are not: helper threads take them chunk by chunk behind the generator
    // (a chunk is theirs once `progress` has passed its end), so that little more than the generator's time is left.
    constexpr int64_t CHUNK = 4096;
    unsigned hw = std::thread::hardware_concurrency();
    int64_t n_threads = std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, 16u), n / 8192);   // a thread per 8192 values at least
    if (const char *e = getenv("MZ_HOST_THREADS")) n_threads = std::min<int64_t>(n_threads, std::max(1, atoi(e)));
    auto transform = [out](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; i++) out[i] = -log(-log(out[i])); };
    if (n_threads <= 1) {
        legacy_one_minus_doubles(mt, out, n, nullptr);
        *mt_pos = mt.pos;
        transform(0, n);
        return MZ_OK;
    }
    std::atomic<int64_t> progress{0};
    HostPool &pool = host_pool();
    const int64_t n_chunks = (n + CHUNK - 1) / CHUNK, n_helpers = std::min<int64_t>(n_threads - 1, pool.size());
    std::lock_guard<std::mutex> one_job(pool.call);
    pool.start((int)n_helpers, [&](int w) {
        for (int64_t c = w; c < n_chunks; c += n_helpers) {
            const int64_t lo = c * CHUNK, hi = std::min(n, lo + CHUNK);
            while (progress.load(std::memory_order_acquire) < hi) std::this_thread::yield();
            transform(lo, hi);
        }
    });
    legacy_one_minus_doubles(mt, out, n, &progress);   // ends with progress = n: every helper comes to its end
    *mt_pos = mt.pos;
    pool.wait();
    return MZ_OK;
}

extern "C" const char *mz_last_error(void) { return g_err.c_str(); }
extern "C" int mz_version(void) { return 1; }

template <typename T>
static int dev_alloc(mz_ctx *c, T **p, size_t n)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, n * sizeof(T) ? n * sizeof(T) : 16));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return MZ_OK;
}

static int scratch_for_device(int device, void **out);  // fixed device array of mz_expand_backup_net_static, below

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

extern "C" mz_ctx *mz_create(const mz_config *cfg)
{
    if (!cfg) { fail(MZ_ERR_ARG, "mz_create: null config"); return nullptr; }
    const int A = cfg->n_actions, S = cfg->n_simulations;
    if (A < 1 || A > MZ_MAX_ACTIONS) { fail(MZ_ERR_ARG, "mz_create: n_actions must be in 1..64"); return nullptr; }
    if (S < 1 || S > (1 << 20)) { fail(MZ_ERR_ARG, "mz_create: n_simulations out of range"); return nullptr; }
    if (cfg->players != 1 && cfg->players != 2) { fail(MZ_ERR_ARG, "mz_create: players must be 1 or 2"); return nullptr; }
    if (cfg->max_batch < 1 || cfg->hidden_size < 0) { fail(MZ_ERR_ARG, "mz_create: bad max_batch/hidden_size"); return nullptr; }
    int lanes = cfg->lanes_per_tree;
    int l0 = next_pow2(A) < 2 ? 2 : next_pow2(A);
    if (lanes == 0) {
        // Small batches: give each tree a wider group (more lanes for leaf evaluation, more waves in flight).
        lanes = l0;
        if (cfg->max_batch <= 1024 && l0 < 64) lanes = 64;
        else if (cfg->max_batch <= 32768 && l0 < 16) lanes = 16;
    }
    if (!shape_supported(lanes, l0)) {
        fail(MZ_ERR_ARG, "mz_create: lanes_per_tree must be the smallest power of two >= n_actions (min 2), or 16, or 64");
        return nullptr;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) { fail(MZ_ERR_HIP, "mz_create: hipSetDevice failed"); return nullptr; }
    mz_ctx *c = new mz_ctx();
    c->cfg = *cfg;
    c->cfg.c_table_h = nullptr;
    c->lanes = lanes;
    c->n_expanded = 0; c->selected = 0; c->B = 0;
    memset(&c->net, 0, sizeof c->net); c->net_scratch = 0; c->net_small = false; c->net_tiny = false; c->net_meta_d = c->net_data_d = nullptr;
    memset(&c->root, 0, sizeof c->root); memset(c->root_bufs, 0, sizeof c->root_bufs);
    memset(&c->mnet, 0, sizeof c->mnet); memset(c->mnet_bufs, 0, sizeof c->mnet_bufs);
    c->chain_G = 0; c->chain_n_seg = 0; c->chain_kk_max = 0; c->chain_ok = -1; c->chain_in_n = 0;
    c->chain_tab = c->chain_acts2 = nullptr; c->chain_acts2_bytes = 0;
    Dev &d = c->d;
    memset(&d, 0, sizeof d);
    Layout &L = d.L;
    L.A = A; L.S = S; L.N = S + 1; L.H = cfg->hidden_size;
    L.prior_f32 = cfg->gumbel ? 1 : 0;
    L.off_cval = 32;
    L.off_crew = L.off_cval + 8 * A;
    L.off_prior = L.off_crew + 8 * A;
    L.off_cvis = L.off_prior + round_up((cfg->gumbel ? 4 : 8) * A, 8);
    // visit counts and node indices fit int16 for N < 32768; the narrow rows are used when they shorten the
    // 64-byte-granular record (A = 9 Gumbel: 320 -> 256 bytes, two full 128-byte lines instead of three)
    L.idx16 = L.N < 32768 && round_up(L.off_cvis + round_up(2 * A, 4) + 2 * A, 64) < round_up(L.off_cvis + 8 * A, 64);
    if (const char *e = getenv("MZ_IDX16")) L.idx16 = L.idx16 && atoi(e) != 0;
    const int isz = L.idx16 ? 2 : 4;
    L.off_cidx = L.off_cvis + round_up(isz * A, 4);
    L.rec = round_up(L.off_cidx + isz * A, 64);
    L.tree_stride = (long long)L.N * L.rec;
    const size_t B = (size_t)cfg->max_batch;
    int rc = MZ_OK;
    rc |= dev_alloc(c, &d.nodes, B * (size_t)L.tree_stride);
    rc |= dev_alloc(c, &d.hidden, B * L.N * (size_t)L.H);
    rc |= dev_alloc(c, &d.legal, B);
    rc |= dev_alloc(c, &d.gumbel, B * A);
    rc |= dev_alloc(c, &d.m_considered, B);
    rc |= dev_alloc(c, &d.sel_parent, B);
    rc |= dev_alloc(c, &d.sel_action, B);
    rc |= dev_alloc(c, &d.sel_ntp, B);
    rc |= dev_alloc(c, &d.sel_depth, B);
    rc |= dev_alloc(c, &d.noise_calls, B);
    rc |= dev_alloc(c, &d.visit_total, B);
    rc |= dev_alloc(c, &d.path, B * S);
    if (!cfg->gumbel) rc |= dev_alloc(c, &d.dec, B * L.N);
    rc |= dev_alloc(c, &d.err, (size_t)1);
    rc |= dev_alloc(c, &c->node_tmp, (size_t)MZ_MAX_ACTIONS);
    rc |= dev_alloc(c, &c->node_act, (size_t)1);
    rc |= dev_alloc(c, &c->leaf_logits, B * A);
    rc |= dev_alloc(c, &c->leaf_value, B);
    rc |= dev_alloc(c, &c->leaf_reward, B);
    int32_t *cvt = nullptr;
    double *ct = nullptr;
    rc |= dev_alloc(c, &cvt, (size_t)(A + 1) * S);
    rc |= dev_alloc(c, &ct, (size_t)S + 2);
    if (rc != MZ_OK) { mz_destroy(c); return nullptr; }
    std::vector<int32_t> tab((size_t)(A + 1) * S);
    for (int m = 0; m <= A; m++) considered_visits_host(m, S, tab.data() + (size_t)m * S);
    std::vector<double> ctab(S + 2);
    for (int n = 0; n < S + 2; n++)
        ctab[n] = cfg->c_table_h ? cfg->c_table_h[n]
                                 : cfg->c_init + std::log(((double)n + cfg->c_base + 1.0) / cfg->c_base);
    if (hipMemcpy(cvt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ct, ctab.data(), ctab.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d.err, 0, 4) != hipSuccess || hipMemset(d.sel_depth, 0xff, B * 4) != hipSuccess) {
        fail(MZ_ERR_HIP, "mz_create: table upload failed");
        mz_destroy(c);
        return nullptr;
    }
    d.cv_table = cvt; d.c_table = ct;
    if (!cfg->gumbel && (S + 2) * (S + 1) <= MZ_SQ_LDS) {
        std::vector<double> sq((size_t)(S + 2) * (S + 1));
        for (int n = 0; n < S + 2; n++)
            for (int v = 0; v < S + 1; v++) sq[(size_t)n * (S + 1) + v] = std::sqrt((double)n / (double)(1 + v));
        double *sqd = nullptr;
        if (dev_alloc(c, &sqd, sq.size()) != MZ_OK ||
            hipMemcpy(sqd, sq.data(), sq.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            fail(MZ_ERR_HIP, "mz_create: sqrt table upload failed");
            mz_destroy(c);
            return nullptr;
        }
        d.sqrt_table = sqd;
    }
    d.players = cfg->players; d.noise_mode = cfg->noise_mode;
    d.RL = l0;
    c->rl = l0;
    d.discount = cfg->discount; d.c_visit = cfg->c_visit; d.c_scale = cfg->c_scale;
    d.seed = cfg->seed;
    void *unused = nullptr;
    // set-up, once per device, while that device is current.  Best effort: a context whose callers never use
    // mz_expand_backup_net_static works without the array, and that entry point repeats the lookup and reports the failure.
    (void)scratch_for_device(cfg->device, &unused);
    return c;
}

extern "C" void mz_destroy(mz_ctx *c)
{
    if (!c) return;
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->net_meta_d) (void)hipFree(c->net_meta_d);
    if (c->net_data_d) (void)hipFree(c->net_data_d);
    for (void *p : c->root_bufs) if (p) (void)hipFree(p);
    for (void *p : c->mnet_bufs) if (p) (void)hipFree(p);
    if (c->chain_tab) (void)hipFree(c->chain_tab);
    if (c->chain_acts2) (void)hipFree(c->chain_acts2);
    for (auto &r : c->rootm)
        for (void *p : r.bufs) if (p) (void)hipFree(p);
    delete c;
}

extern "C" int mz_set_seed(mz_ctx *c, uint64_t seed)
{
    if (!c) return fail(MZ_ERR_ARG, "mz_set_seed: null context");
    c->cfg.seed = seed;
    c->d.seed = seed;   // kernels receive the context's device descriptor by value at every launch
    return MZ_OK;
}

// Calls f(lanes tag, action-lanes tag, gumbel tag) for the (lanes per tree, action lanes, rule) of a context.
// Supported shapes: a group is either exactly the action lanes, or 16 or 64 lanes.
template <int N> using ic = std::integral_constant<int, N>;
template <typename F>
static bool dispatch_shape(int lanes, int rl, bool gumbel, F &&f)
{
#define MZ_SHAPE(L, R)                                                     \
    if (lanes == L && rl == R) {                                           \
        if (gumbel) f(ic<L>(), ic<R>(), std::true_type());                 \
        else f(ic<L>(), ic<R>(), std::false_type());                       \
        return true;                                                       \
    }
    MZ_SHAPE(2, 2) MZ_SHAPE(16, 2) MZ_SHAPE(64, 2)
    MZ_SHAPE(4, 4) MZ_SHAPE(16, 4) MZ_SHAPE(64, 4)
    MZ_SHAPE(8, 8) MZ_SHAPE(16, 8) MZ_SHAPE(64, 8)
    MZ_SHAPE(16, 16) MZ_SHAPE(64, 16)
    MZ_SHAPE(32, 32) MZ_SHAPE(64, 32)
    MZ_SHAPE(64, 64)
#undef MZ_SHAPE
    return false;
}

static unsigned grid_for(long long groups, int lanes) { return (unsigned)((groups * lanes + MZ_BLOCK - 1) / MZ_BLOCK); }

// Heavy kernels run with the context's group size; light once-per-search kernels with group = action lanes.
#define MZ_LAUNCH(c, KERNEL, light, groups, lds_bytes, stream, ...)                                          \
    dispatch_shape((light) ? (c)->rl : (c)->lanes, (c)->rl, (c)->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) { \
        hipLaunchKernelGGL((KERNEL<decltype(L_)::value, decltype(R_)::value, decltype(G_)::value>),           \
                           dim3(grid_for((groups), decltype(L_)::value)), dim3(MZ_BLOCK), (lds_bytes),        \
                           (hipStream_t)(stream), __VA_ARGS__);                                               \
    })

static int check_ctx(mz_ctx *c, const char *fn)
{
    if (!c) return fail(MZ_ERR_ARG, std::string(fn) + ": null context");
    if (hipSetDevice(c->cfg.device) != hipSuccess) return fail(MZ_ERR_HIP, std::string(fn) + ": hipSetDevice failed");
    (void)hipGetLastError();  // a stale error left by other users of the runtime is not ours to report
    return MZ_OK;
}

// hipFuncSetAttribute costs microseconds of host time per call: the dynamic LDS limit of a kernel is raised only when
// a launch needs more than an earlier one was granted.  The limit belongs to (device, function), and contexts of several
// devices or host threads may launch at once, hence the key and the lock.
hipError_t mz::ensure_dynamic_lds(const void *kern, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &g = granted[{dev, kern}];
    if (bytes <= g) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) g = bytes;
    return e;
}

static int launch_status(const char *fn)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MZ_ERR_HIP, std::string(fn) + ": launch failed: " + hipGetErrorString(e));
    return MZ_OK;
}

extern "C" int mz_root_prepare(mz_ctx *c, int32_t B, const float *logits, const double *dirichlet,
                               double frac, const uint8_t *legal, int32_t shift_and_mask, void *out_prior,
                               void *stream)
{
    int rc = check_ctx(c, "mz_root_prepare");
    if (rc) return rc;
    if (B < 1 || !logits || !out_prior) return fail(MZ_ERR_ARG, "mz_root_prepare: null/empty argument");
    if (!legal && (shift_and_mask || c->cfg.gumbel)) return fail(MZ_ERR_ARG, "mz_root_prepare: legal mask required");
    if (!c->cfg.gumbel && !dirichlet) return fail(MZ_ERR_ARG, "mz_root_prepare: PUCT needs dirichlet noise");
    MZ_LAUNCH(c, k_root_prepare, true, B, 0, stream, (int)B, c->cfg.n_actions, logits, dirichlet, frac, legal,
                (int)shift_and_mask, out_prior);
    return launch_status("mz_root_prepare");
}

extern "C" int mz_tree_reset(mz_ctx *c, int32_t B, const void *root_prior, const double *root_value,
                             const float *root_hidden, const int32_t *root_to_play, const uint8_t *legal,
                             const double *gumbel, uint64_t tree_id0, void *stream)
{
    int rc = check_ctx(c, "mz_tree_reset");
    if (rc) return rc;
    if (B < 1 || B > c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_tree_reset: B outside 1..max_batch");
    if (!root_prior || !root_value || !root_to_play || !legal) return fail(MZ_ERR_ARG, "mz_tree_reset: null argument");
    if (c->cfg.gumbel && !gumbel) return fail(MZ_ERR_ARG, "mz_tree_reset: Gumbel search needs gumbel noise");
    c->B = B; c->d.B = B; c->d.tree_id0 = tree_id0;
    c->n_expanded = 0; c->selected = 0;
    MZ_LAUNCH(c, k_reset, true, B, 0, stream, c->d, root_prior, root_value, root_hidden, root_to_play, legal, gumbel,
                (int)c->cfg.max_considered_actions);
    return launch_status("mz_tree_reset");
}

static int check_noise(mz_ctx *c, const double *noise, int32_t max_calls, const char *fn)
{
    if (!c->cfg.gumbel && c->cfg.noise_mode == MZ_NOISE_BUFFER && (!noise || max_calls < 1))
        return fail(MZ_ERR_ARG, std::string(fn) + ": noise_mode is MZ_NOISE_BUFFER but no noise buffer given");
    return MZ_OK;
}

extern "C" int mz_select(mz_ctx *c, const double *noise, int32_t noise_max_calls, float *out_parent_hidden,
                         int32_t *out_parent, int32_t *out_action, void *stream)
{
    int rc = check_ctx(c, "mz_select");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_select: no tree (call mz_tree_reset first)");
    if (c->n_expanded >= c->cfg.n_simulations) return fail(MZ_ERR_STATE, "mz_select: all simulations already done");
    if ((rc = check_noise(c, noise, noise_max_calls, "mz_select"))) return rc;
    if (c->cfg.noise_mode != MZ_NOISE_BUFFER) noise = nullptr;
    MZ_LAUNCH(c, k_select, false, c->B, 0, stream, c->d, noise, (int)noise_max_calls, out_parent_hidden, out_parent,
                out_action);
    c->selected = 1;
    return launch_status("mz_select");
}

extern "C" int mz_select_outputs(mz_ctx *c, const int32_t **parent, const int32_t **action,
                                 const int32_t **ntp, const int32_t **depth)
{
    if (!c) return fail(MZ_ERR_ARG, "mz_select_outputs: null context");
    if (parent) *parent = c->d.sel_parent;
    if (action) *action = c->d.sel_action;
    if (ntp) *ntp = c->d.sel_ntp;
    if (depth) *depth = c->d.sel_depth;
    return MZ_OK;
}

extern "C" int mz_expand_backup(mz_ctx *c, const float *logits, const double *value, const double *reward,
                                const float *next_hidden, void *stream)
{
    int rc = check_ctx(c, "mz_expand_backup");
    if (rc) return rc;
    if (!c->selected) return fail(MZ_ERR_STATE, "mz_expand_backup: no pending selection (call mz_select first)");
    if (!logits || !value || !reward) return fail(MZ_ERR_ARG, "mz_expand_backup: null argument");
    int leaf = c->n_expanded + 1;
    MZ_LAUNCH(c, k_expand_backup, false, c->B, 0, stream, c->d, leaf, logits, value, reward, next_hidden);
    c->n_expanded = leaf;
    c->selected = 0;
    return launch_status("mz_expand_backup");
}

// Whole search, level by level: per simulation one launch per network level and one tree launch (epilogue, backup,
// next selection).  lvl_issue_first: the first selection and its network inputs, all trees.  lvl_issue_range: the
// chain of launches for the trees of range r of n_ranges (ranges are whole 32-tree tiles).
static int lvl_issue_first(mz_ctx *c, const double *noise, int32_t noise_max_calls, hipStream_t st)
{
    MZ_LAUNCH(c, k_select, false, c->B, 0, st, c->d, noise, (int)noise_max_calls, (float *)nullptr, (int32_t *)nullptr,
              (int32_t *)nullptr);
    launch_stage_inputs(st, c->d, c->mnet, (float *)c->mnet_bufs[6]);
    return launch_status("mz_search_run");
}

static int lvl_issue_range(mz_ctx *c, int n_sims, const double *noise, int32_t noise_max_calls, hipStream_t rs, int r,
                           int n_ranges)
{
    const MfmaNet &net = c->mnet;
    float *acts = (float *)c->mnet_bufs[6];
    const int32_t *desc = (const int32_t *)c->mnet_bufs[9];
    const int tiles = (c->B + 31) / 32;
    const int tile0 = (int)((long long)tiles * r / n_ranges), tile1 = (int)((long long)tiles * (r + 1) / n_ranges);
    const int t_first = tile0 * 32, t_end = std::min(tile1 * 32, c->B);
    const int first_leaf = c->n_expanded + 1;
    // PUCT: decisions that no tie-break draw can change are kept in a table and re-made only for the nodes a simulation
    // touched (tree_ops.h puct_decide).  The table describes nothing yet: other calls may have changed the trees.
    bool cached = !c->cfg.gumbel && c->d.dec != nullptr;
    if (const char *e = getenv("MZ_PUCT_CACHE")) cached = cached && atoi(e) != 0;
    if (cached)
        HIPCHK(hipMemsetAsync(c->d.dec + (size_t)t_first * c->d.L.N, 0, (size_t)(t_end - t_first) * c->d.L.N * 4, rs));
    for (int s = 0; s < n_sims; s++) {
        for (int l = 0; l < net.n_levels; l++) {
            const int j0 = c->mnet_level_job0[l], nb = c->mnet_level_job0[l + 1] - j0;
            if (nb < 1) continue;
            HIPCHK(launch_net_level_rows((unsigned)nb, (unsigned)(tile1 - tile0), c->mnet_max_k16, rs, net,
                                         desc + MZ_LVL_DESC * j0, acts, t_first, l));
        }
        if (!launch_leaf_backup_select(c->rl, c->cfg.gumbel != 0, rs, c->d, net, t_first, t_end, first_leaf + s, acts, noise,
                                       (int)noise_max_calls, (s + 1 < n_sims ? 1 : 0) | (cached ? 2 : 0)))
            return fail(MZ_ERR_ARG, "mz_search_run: unsupported action count");
    }
    return launch_status("mz_search_run");
}

static int search_by_level(mz_ctx *c, int n_sims, const double *noise, int32_t noise_max_calls, hipStream_t stream)
{
    if (!shape_supported(64, c->rl)) return fail(MZ_ERR_ARG, "mz_search_run: unsupported action count");
    int rc = lvl_issue_first(c, noise, noise_max_calls, stream);
    if (rc) return rc;
    // One range: splitting the batch into ranges on streams of their own (direct launches from one or several host
    // threads, or one captured graph per range) was measured and never won - a launch costs the host 7 us, about
    // half a kernel's duration, so two ranges already starve each other's streams.
    if ((rc = lvl_issue_range(c, n_sims, noise, noise_max_calls, stream, 0, 1))) return rc;
    c->n_expanded += n_sims;
    return MZ_OK;
}

// ---------------------------------------------------------------------------------------------------
// The same searches in ONE launch (search_chain.h): clusters of G workgroups run the network level by level for tiles
// of 16 trees while tree waves of the same workgroups finish and restart the trees; data crosses workgroups as tagged
// granules.  chain_examine decides once per uploaded program whether that kernel serves it and says why not.
static int chain_examine(mz_ctx *c)
{
    if (c->chain_ok >= 0) return c->chain_ok;
    c->chain_ok = 0;
    const MfmaNet &net = c->mnet;
    const std::vector<int32_t> &st = c->mnet_steps_h;
    auto no = [&](const char *why) { c->chain_why = why; return 0; };
    if (st.empty()) return no("no program");
    if (!search_chain_built(c->rl)) return no("the kernel is not built for this action count");
    if (net.n_levels > MZ_CH_MAXL) return no("too many network levels");
    const int n_steps = net.n_steps, H = net.io[6], asz = net.io[8];
    const int in_lo = net.io[0], in_n = (H + asz + 15) / 16 * 16;
    if (net.io[1] != in_lo + H) return no("the action does not follow the hidden state in the activation row");
    if (in_lo % 4 || in_lo + in_n > net.stride) return no("network inputs not 16-byte aligned inside the row");
    // every tensor in a place of its own, whole 16-float tiles: a column tile is written whole, and nothing orders a
    // writer behind the readers of whatever shared its place (a launch boundary does that for the level kernels)
    struct Span { int lo, hi, phase; };
    std::vector<Span> spans{{in_lo, in_lo + in_n, 0}};
    int kk_max = 16;
    std::vector<int> per_level(net.n_levels, 0);
    for (int s = 0; s < n_steps; s++) {
        const int32_t *q = &st[12 * s];
        if (q[4] % 16) return no("a layer's outputs do not start on a multiple of 16 floats");
        if (q[3] % 4) return no("a layer's inputs are not 16-byte aligned");
        if (q[5] >= 0 && q[5] % 16) return no("a residual input does not start on a multiple of 16 floats");
        spans.push_back({q[4], q[4] + q[9] * 16, q[0] + 1});
        kk_max = std::max(kk_max, q[2] * 16);
        if (q[2] > MZ_CH_MAXB) return no("a layer with more inputs than the kernel's weight ring serves");
        per_level[q[0]] += q[9];
    }
    for (size_t i = 0; i < spans.size(); i++)
        for (size_t j = i + 1; j < spans.size(); j++)
            if (spans[i].lo < spans[j].hi && spans[j].lo < spans[i].hi)
                return no("two tensors share a place in the activation row (pack the network without reuse)");
    for (int l = 0; l < net.n_levels; l++) if (per_level[l] < 1) return no("a level without layers");
    auto producer = [&](int off, int width) -> int {   // phase that writes [off, off + width)
        for (const Span &sp : spans) if (off >= sp.lo && off + width <= sp.hi) return sp.phase;
        return -1;
    };
    for (int s = 0; s < n_steps; s++) {
        const int32_t *q = &st[12 * s];
        const int pi = producer(q[3], q[2] * 16), ps = q[5] >= 0 ? producer(q[5], q[9] * 16) : 0;
        if (pi < 0 || pi > q[0] || ps < 0 || ps > q[0]) return no("a layer reads what no earlier phase writes whole");
    }
    const int outs[4] = {net.io[2], net.io[3], net.io[4], net.io[5]};
    const int outn[4] = {H, net.io[9], net.io[7], net.io[9]};
    for (int k = 0; k < 4; k++) {
        c->chain_out_phase[k] = producer(outs[k], (outn[k] + 15) / 16 * 16);
        if (c->chain_out_phase[k] < 1) return no("a network output is not a whole layer output");
    }
    if (search_chain_lds(net, kk_max, c->cfg.gumbel != 0) > 160 * 1024) return no("staging and tree rows exceed LDS");
    const size_t rows = ((size_t)c->cfg.max_batch + 31) / 32 * 32;
    if (rows * net.stride * 8 >= ((size_t)1 << 31)) return no("tagged activations exceed 2 GiB");
    c->chain_kk_max = kk_max; c->chain_in_n = in_n;
    c->chain_why.clear();
    return c->chain_ok = 1;
}

// Work lists for clusters of G workgroups: per level the column tiles of its layers, in order, are dealt in equal
// shares; a share is cut into segments of at most 8 column tiles of one layer.
static int chain_prepare(mz_ctx *c, int G)
{
    if (c->chain_G == G && c->chain_tab) return MZ_OK;
    const MfmaNet &net = c->mnet;
    const std::vector<int32_t> &st = c->mnet_steps_h, &cti = c->mnet_ctinfo_h, &kidx = c->mnet_kidx_h;
    const int H = net.io[6], asz = net.io[8], in_lo = net.io[0], in_n = c->chain_in_n;
    auto producer = [&](int off) -> int {
        if (off >= in_lo && off < in_lo + in_n) return 0;
        for (int s = 0; s < net.n_steps; s++)
            if (off >= st[12 * s + 4] && off < st[12 * s + 4] + st[12 * s + 9] * 16) return st[12 * s] + 1;
        return 0;
    };
    (void)H; (void)asz;
    std::vector<int32_t> seg, range;
    for (int l = 0; l < net.n_levels; l++) {
        // the level's layers cut into segments of at most 8 column tiles, dealt largest first to the workgroup with
        // the least work so far (a segment costs a staging of its inputs and one or two column tiles per wave)
        struct Piece { int s, ct0, n; };
        std::vector<Piece> pieces;
        int total = 0;
        for (int s = 0; s < net.n_steps; s++) if (st[12 * s] == l) total += st[12 * s + 9];
        const int cap = std::max(1, std::min(8, (total + G - 1) / G));   // so that every workgroup of the cluster gets some
        for (int s = 0; s < net.n_steps; s++)
            if (st[12 * s] == l)
                for (int ct = 0; ct < st[12 * s + 9]; ct += cap) pieces.push_back({s, ct, std::min(cap, st[12 * s + 9] - ct)});
        std::stable_sort(pieces.begin(), pieces.end(), [](const Piece &a, const Piece &b) { return a.n > b.n; });
        std::vector<std::vector<Piece>> mine(G);
        std::vector<int> load(G, 0);
        for (const Piece &pc : pieces) {
            int m = 0;
            for (int k = 1; k < G; k++) if (load[k] < load[m]) m = k;
            mine[m].push_back(pc);
            load[m] += 4 + pc.n;
        }
        for (int m = 0; m < G; m++) {
            range.push_back((int)(seg.size() / MZ_CH_SEG));
            for (const Piece &pc : mine[m]) {
                const int32_t *q = &st[12 * pc.s];
                int32_t d[MZ_CH_SEG] = {q[2] * 16, q[3], q[4] + 16 * pc.ct0, q[5] >= 0 ? q[5] + 16 * pc.ct0 : -1, q[6],
                                        q[8] + 16 * pc.ct0, pc.n, 1, producer(q[3]), q[5] >= 0 ? producer(q[5]) : 0, l + 1, 0};
                for (int w = 0; w < pc.n; w++) {
                    const int32_t *cj = &cti[3 * (q[10] + pc.ct0 + w)];
                    d[12 + 3 * w] = cj[0] * 4; d[13 + 3 * w] = cj[2]; d[14 + 3 * w] = cj[1];
                    for (int i = 0; i < cj[2]; i++) if (kidx[cj[1] + i] != i) d[7] = 0;
                }
                seg.insert(seg.end(), d, d + MZ_CH_SEG);
            }
        }
        range.push_back((int)(seg.size() / MZ_CH_SEG));
    }
    {   // a workgroup keeps its segments of all levels in LDS
        std::vector<int> per_block(G, 0);
        for (int l = 0; l < net.n_levels; l++)
            for (int m = 0; m < G; m++) per_block[m] += range[l * (G + 1) + m + 1] - range[l * (G + 1) + m];
        for (int m = 0; m < G; m++)
            if (per_block[m] > MZ_CH_MAXSEG) return fail(MZ_ERR_ARG, "mz_search_run: too many network segments per workgroup for the chained search");
    }
    if (c->chain_tab) { (void)hipFree(c->chain_tab); c->chain_tab = nullptr; }
    c->chain_G = 0;
    c->chain_n_seg = (int)(seg.size() / MZ_CH_SEG);
    std::vector<int32_t> tab(seg);
    tab.insert(tab.end(), range.begin(), range.end());
    HIPCHK(hipMalloc(&c->chain_tab, tab.size() * 4));
    HIPCHK(hipMemcpy(c->chain_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    if (!c->chain_acts2) {
        const size_t rows = ((size_t)c->cfg.max_batch + 31) / 32 * 32;
        c->chain_acts2_bytes = rows * net.stride * 8;
        HIPCHK(hipMalloc(&c->chain_acts2, c->chain_acts2_bytes));
    }
    c->chain_G = G;
    return MZ_OK;
}

static int search_chained(mz_ctx *c, int n_sims, const double *noise, int32_t noise_max_calls, hipStream_t stream)
{
    int n_cu = 0;
    HIPCHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->cfg.device));
    const int n_tiles = (c->B + MZ_CH_TM - 1) / MZ_CH_TM;
    // workgroups per cluster: 8 (their 32 tree waves serve one tile), more when the batch has fewer tiles than
    // clusters (every CU should have work)
    int G = 8;
    while (G < 32 && n_cu / (2 * G) >= n_tiles) G *= 2;
    if (const char *e = getenv("MZ_CHAIN_G")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32) G = v; }
    if (n_cu < G) return fail(MZ_ERR_HIP, "mz_search_run: fewer compute units than one cluster of the chained search needs");
    int rc = chain_prepare(c, G);
    if (rc) return rc;
    ChainArgs ca{};
    ca.seg = (const int32_t *)c->chain_tab;
    ca.blk_range = ca.seg + (size_t)c->chain_n_seg * MZ_CH_SEG;
    ca.acts2 = (float *)c->chain_acts2;
    ca.acts2_bytes = (unsigned)c->chain_acts2_bytes;
    ca.G = G; ca.n_clusters = std::min(n_cu / G, n_tiles); ca.n_tiles = n_tiles;
    ca.kk_max = c->chain_kk_max; ca.in_lo = c->mnet.io[0]; ca.in_n = c->chain_in_n;
    for (int k = 0; k < 4; k++) ca.out_phase[k] = c->chain_out_phase[k];
    ca.first_leaf = c->n_expanded + 1; ca.n_sims = n_sims;
    ca.noise = noise; ca.noise_max_calls = noise_max_calls;
    double ms = 250.0;
    if (const char *e = getenv("MZ_CHAIN_TIMEOUT_MS")) { const double v = atof(e); if (v >= 1.0 && v <= 10000.0) ms = v; }
    ca.timeout = (long long)(ms * 1e5);   // the clock the kernel reads ticks at 100 MHz
    // tags count from 1 within a launch: whatever the last search left must not look like data
    HIPCHK(hipMemsetAsync(c->chain_acts2, 0, (size_t)n_tiles * MZ_CH_TM * c->mnet.stride * 8, stream));
    if (launch_search_chain(c->rl, c->cfg.gumbel != 0, stream, c->d, c->mnet, ca, (unsigned)(n_cu / G * G)) != hipSuccess)
        return fail(MZ_ERR_HIP, "mz_search_run: could not raise the dynamic LDS limit");
    c->n_expanded += n_sims;
    return launch_status("mz_search_run");
}

// mz_search_run, optionally followed by the final action rule (`act->action` non-NULL).  *act_done tells
// whether the search kernel already ran the rule.
static int search_run(mz_ctx *c, int32_t net_kind, const void *net_params_h, const double *noise,
                      int32_t noise_max_calls, void *stream, const ActOut *act, bool *act_done)
{
    int rc = check_ctx(c, "mz_search_run");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_search_run: no tree (call mz_tree_reset first)");
    if (c->selected) return fail(MZ_ERR_STATE, "mz_search_run: a selection is pending");
    if (net_kind != MZ_NET_HASH && net_kind != MZ_NET_DENSE && net_kind != MZ_NET_MFMA)
        return fail(MZ_ERR_ARG, "mz_search_run: unsupported net_kind");
    if ((rc = check_noise(c, noise, noise_max_calls, "mz_search_run"))) return rc;
    if (c->cfg.noise_mode != MZ_NOISE_BUFFER) noise = nullptr;
    int n_sims = c->cfg.n_simulations - c->n_expanded;
    if (n_sims <= 0) return fail(MZ_ERR_STATE, "mz_search_run: all simulations already done");
    if (net_kind == MZ_NET_MFMA) {
        if (!c->mnet.steps) return fail(MZ_ERR_STATE, "mz_search_run: no network uploaded (call mz_set_network_mfma first)");
        // more than 16 actions, tiles that do not fit in LDS, or too few trees to give every CU a tile: level by level
        bool by_level = c->rl > 16 || !c->mnet.sched_ok8 ||
                        (size_t)c->mnet.stride * 32 * 4 + (size_t)32 * 16 * 8 > 160 * 1024;
        if (const char *e = getenv("MZ_SEARCH_LEVELS")) by_level = by_level || atoi(e) != 0;
        if (by_level) {
            // One launch for the whole search (search_chain.h) on request only: MZ_SEARCH_CHAIN = 1 where the kernel
            // serves the program, 2: demanded (tests that must not pass on the level-by-level path).  Measured slower
            // than the level-by-level launches on the shapes tried, hence not the default.
            bool chained = false;
            if (const char *e = getenv("MZ_SEARCH_CHAIN")) {
                const int want = atoi(e);
                chained = want != 0 && chain_examine(c) == 1;
                if (want == 2 && !chained)
                    return fail(MZ_ERR_ARG, "mz_search_run: the chained search does not serve this configuration: " + c->chain_why);
            }
            if (chained) return search_chained(c, n_sims, noise, noise_max_calls, (hipStream_t)stream);
            return search_by_level(c, n_sims, noise, noise_max_calls, (hipStream_t)stream);
        }
        // Gumbel trees of up to 16 actions walk down cached decisions and re-decide only the nodes a simulation changed
        // (search_cached.h): tiles of 32 trees, or of 16 where only those fit in LDS (longer searches, wider networks).
        int ctm = 32;
        if (const char *e = getenv("MZ_SEARCH_TM")) ctm = atoi(e) == 16 ? 16 : 32;
        else if (c->cfg.gumbel && !search_gcache_supported(c->d, c->mnet, 32) && search_gcache_supported(c->d, c->mnet, 16)) ctm = 16;
        bool cached = c->cfg.gumbel && search_gcache_supported(c->d, c->mnet, ctm);
        if (const char *e = getenv("MZ_SEARCH_CACHED")) {   // 0: never, 2: demanded (tests that must not pass vacuously)
            if (atoi(e) == 2 && !cached)
                return fail(MZ_ERR_ARG, "mz_search_run: the cached-decision kernel does not serve this configuration");
            cached = cached && atoi(e) != 0;
        }
        hipError_t attr_err;
        if (cached) {
            attr_err = launch_search_gcache(ctm, (hipStream_t)stream, c->d, c->mnet, c->n_expanded + 1, n_sims);
        } else {
            // 32-tree tiles: 16-tree tiles (two or three blocks per CU) were measured no faster
            if ((size_t)c->mnet.stride * 32 * 4 + (size_t)32 * 16 * 8 > 160 * 1024)
                return fail(MZ_ERR_ARG, "mz_search_run: activations of a 32-tree tile do not fit in LDS beside the backup staging; use the step API");
            attr_err = launch_search_mfma(c->rl, c->cfg.gumbel != 0, (hipStream_t)stream, c->d, c->mnet, c->n_expanded + 1, n_sims,
                                          noise, (int)noise_max_calls);
        }
        if (attr_err != hipSuccess) return fail(MZ_ERR_HIP, "mz_search_run: could not raise the dynamic LDS limit");
        c->n_expanded += n_sims;
        return launch_status("mz_search_run");
    }
    if (net_kind == MZ_NET_DENSE) {
        if (!c->net.meta) return fail(MZ_ERR_STATE, "mz_search_run: no network uploaded (call mz_set_network first)");
        // Groups per block: as many as fill a block, but small batches of whole-wave groups use smaller
        // blocks so that they spread over all CUs (256 trees -> 256 one-wave blocks).
        int gpb = MZ_BLOCK / c->lanes;
        if (c->lanes == 64)
            while (gpb > 1 && (c->B + gpb - 1) / gpb < 256) gpb >>= 1;
        if (const char *e = getenv("MZ_GROUPS_PER_BLOCK")) { int v = atoi(e); if (v >= 1 && v <= MZ_BLOCK / c->lanes) gpb = v; }
        // The kernel's static tables (backup staging, PUCT c(n) and sqrt tables) share the 160 KB with the
        // weights and the per-group activations: narrow groups give up groups per block until all fit.
        const size_t lds_static = (size_t)MZ_BLOCK * 8 + (c->cfg.gumbel ? 32 : (size_t)(MZ_CT_LDS + MZ_SQ_LDS) * 8);
        auto lds_for = [&](int g) { return ((size_t)c->net.n_data + c->net.n_meta + (size_t)g * c->net_scratch) * 4; };
        while (gpb * c->lanes > 64 && lds_for(gpb) + lds_static > 160 * 1024) gpb >>= 1;
        const unsigned n_blocks = (unsigned)((c->B + gpb - 1) / gpb), n_threads = (unsigned)(gpb * c->lanes);
        const size_t lds = lds_for(gpb);
        if (lds + lds_static > 160 * 1024)
            return fail(MZ_ERR_ARG, "mz_search_run: network does not fit in LDS for the fused search; use the step API");
        // Small batches are latency-bound: whole-wave groups keep their trees in LDS when one block per CU
        // covers the batch and everything fits beside the kernel's static tables.
        const Layout &L = c->d.L;
        const int tree_lds = ((L.N * L.rec + 15) & ~15) + ((L.N * L.H * 4 + 15) & ~15) + ((L.S * 4 + 15) & ~15) +
                             ((L.N * 4 + 15) & ~15);
        const size_t lds_tree = ((lds + 15) & ~(size_t)15) + (size_t)gpb * tree_lds;
        bool tree_in_lds = c->lanes == 64 && n_blocks <= 256 && lds_tree + lds_static <= 160 * 1024;
        if (const char *e = getenv("MZ_TREE_LDS")) tree_in_lds = tree_in_lds && atoi(e) != 0;
        hipError_t attr_err = hipSuccess;
        // 2-action PUCT trees of at most 64 nodes: one wave per tree with the tree in registers, whatever group
        // width the context was created with (measured faster than the lane groups at 256 and at 8192 trees)
        bool tiny = c->net_tiny;
        if (const char *e = getenv("MZ_NET_TINY")) tiny = tiny && atoi(e) != 0;
        // hidden states of the smallest networks: registers while a wave has its SIMD to itself, LDS when the chip is full
        bool hid_lds = tiny && c->B > 1024;
        if (const char *e = getenv("MZ_HID_LDS")) hid_lds = tiny && atoi(e) != 0;
        const int hid_bytes = (tiny && !hid_lds) ? 0 : (L.N * L.H * 4 + 15) & ~15;
        int wpb = MZ_BLOCK / 64;  // waves (= trees) per block: fewer for small batches, so that they spread over all CUs
        while (wpb > 1 && (c->B + wpb - 1) / wpb < 256) wpb >>= 1;
        auto lds_reg_for = [&](int g) { return ((lds_for(g) + 15) & ~(size_t)15) + (size_t)g * hid_bytes; };
        while (wpb > 1 && lds_reg_for(wpb) + (size_t)(MZ_CT_LDS + MZ_SQ_LDS) * 8 > 160 * 1024) wpb >>= 1;
        const size_t lds_reg = lds_reg_for(wpb);
        bool tree_in_regs = !c->cfg.gumbel && L.A == 2 && L.N <= 64 && c->net_small && !L.prior_f32 &&
                            lds_reg + (size_t)(MZ_CT_LDS + MZ_SQ_LDS) * 8 <= 160 * 1024;
        if (const char *e = getenv("MZ_TREE_REG")) tree_in_regs = tree_in_regs && atoi(e) != 0;
        if (tree_in_regs) {
            auto kern = !tiny ? k_search_reg<0> : L.H == 8 ? (hid_lds ? k_search_reg<8, true> : k_search_reg<8>)
                                                            : (hid_lds ? k_search_reg<4, true> : k_search_reg<4>);
            attr_err = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (size_t)lds_reg);
            if (attr_err != hipSuccess) return fail(MZ_ERR_HIP, "mz_search_run: could not raise the dynamic LDS limit");
            ActOut ao{};
            if (act && (size_t)c->net_scratch * 4 >= (size_t)L.rec) { ao = *act; *act_done = true; }
            hipLaunchKernelGGL(kern, dim3((unsigned)((c->B + wpb - 1) / wpb)), dim3((unsigned)(wpb * 64)), lds_reg,
                               (hipStream_t)stream, c->d, c->net, c->n_expanded + 1, n_sims, noise, (int)noise_max_calls,
                               c->net_scratch, hid_bytes, ao);
            c->n_expanded += n_sims;
            return launch_status("mz_search_run");
        }
        dispatch_shape(c->lanes, c->rl, c->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) {
            constexpr int LN = decltype(L_)::value, RN = decltype(R_)::value;
            constexpr bool GM = decltype(G_)::value;
            auto launch = [&](auto kern, size_t bytes, int tree_bytes) {
                attr_err = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (size_t)bytes);
                hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(n_threads), bytes, (hipStream_t)stream, c->d,
                                   c->net, c->n_expanded + 1, n_sims, noise, (int)noise_max_calls, c->net_scratch,
                                   tree_bytes);
            };
            if constexpr (LN == 64) {
                if (tree_in_lds) { launch(k_search_net<LN, RN, GM, true>, lds_tree, tree_lds); return; }
            }
            launch(k_search_net<LN, RN, GM, false>, lds, 0);
        });
        if (attr_err != hipSuccess) return fail(MZ_ERR_HIP, "mz_search_run: could not raise the dynamic LDS limit");
        c->n_expanded += n_sims;
        return launch_status("mz_search_run");
    }
    if (!net_params_h) return fail(MZ_ERR_ARG, "mz_search_run: null network parameters");
    if (c->cfg.hidden_size < 2) return fail(MZ_ERR_ARG, "mz_search_run: hash network needs hidden_size >= 2");
    const int32_t *p = reinterpret_cast<const int32_t *>(net_params_h);
    HashNet hn{p[0], p[1], p[2]};
    MZ_LAUNCH(c, k_search_hash, false, c->B, 0, stream, c->d, hn, c->n_expanded + 1, n_sims, noise, (int)noise_max_calls);
    c->n_expanded += n_sims;
    return launch_status("mz_search_run");
}

extern "C" int mz_search_run(mz_ctx *c, int32_t net_kind, const void *net_params_h, const double *noise,
                             int32_t noise_max_calls, void *stream)
{
    bool act_done = false;
    return search_run(c, net_kind, net_params_h, noise, noise_max_calls, stream, nullptr, &act_done);
}

extern "C" int mz_search_run_act(mz_ctx *c, int32_t net_kind, const void *net_params_h, const double *noise,
                                 int32_t noise_max_calls, double temperature, const double *u, int32_t *out_action,
                                 double *out_probs, double *out_root_value, double *out_q, void *stream)
{
    if (!out_action || !out_probs || !out_root_value) return fail(MZ_ERR_ARG, "mz_search_run_act: null output");
    const ActOut act{temperature, u, out_action, out_probs, out_root_value, out_q};
    bool act_done = false;
    int rc = search_run(c, net_kind, net_params_h, noise, noise_max_calls, stream, &act, &act_done);
    if (rc || act_done) return rc;
    return mz_to_act(c, temperature, u, out_action, out_probs, out_root_value, out_q, stream);
}

extern "C" int mz_search_move(mz_ctx *c, int32_t B, const float *obs, const uint8_t *legal, const int32_t *to_play,
                              const double *dirichlet, double frac, const double *gumbel, uint64_t tree_id0,
                              const double *noise, int32_t noise_max_calls, double temperature, const double *u,
                              int32_t *out_action, double *out_probs, double *out_root_value, double *out_q,
                              void *stream)
{
    int rc = check_ctx(c, "mz_search_move");
    if (rc) return rc;
    if (!c->root.rep_meta) return fail(MZ_ERR_STATE, "mz_search_move: no root network (call mz_set_root_network first)");
    if (!c->net.meta) return fail(MZ_ERR_STATE, "mz_search_move: no network uploaded (call mz_set_network first)");
    if (B < 1 || B > c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_search_move: B outside 1..max_batch");
    if (!obs || !legal || !to_play) return fail(MZ_ERR_ARG, "mz_search_move: null argument");
    if (c->cfg.gumbel ? !gumbel : !dirichlet) return fail(MZ_ERR_ARG, "mz_search_move: root noise missing");
    if (!out_action || !out_probs || !out_root_value) return fail(MZ_ERR_ARG, "mz_search_move: null output");
    // The roots stay a launch of their own: evaluated by the search kernel's single wave per tree they took longer
    // than k_begin_net's launch costs, and their arguments crowded the simulation loop's scalar registers.
    rc = mz_search_begin_net(c, B, obs, legal, to_play, dirichlet, frac, gumbel, tree_id0, nullptr, nullptr, stream);
    if (rc) return rc;
    const ActOut act{temperature, u, out_action, out_probs, out_root_value, out_q};
    bool act_done = false;
    rc = search_run(c, MZ_NET_DENSE, nullptr, noise, noise_max_calls, stream, &act, &act_done);
    if (rc || act_done) return rc;
    return mz_to_act(c, temperature, u, out_action, out_probs, out_root_value, out_q, stream);
}

extern "C" int mz_to_act(mz_ctx *c, double temperature, const double *u, int32_t *out_action,
                         double *out_probs, double *out_root_value, double *out_q, void *stream)
{
    int rc = check_ctx(c, "mz_to_act");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_to_act: no tree");
    if (!out_action || !out_probs || !out_root_value) return fail(MZ_ERR_ARG, "mz_to_act: null output");
    MZ_LAUNCH(c, k_to_act, true, c->B, 0, stream, c->d, temperature, u, out_action, out_probs, out_root_value, out_q);
    return launch_status("mz_to_act");
}

extern "C" int mz_make_targets(const mz_trajectories *tr, int32_t n_samples, const int32_t *game_idx, const int32_t *t,
                               const int32_t *absorbed, int32_t td_steps, int32_t unroll_steps, double gamma,
                               const double *gamma_pow, int32_t n_stack, int32_t stack_action, float *out_obs,
                               float *out_actions, float *out_value, float *out_reward, float *out_policy,
                               void *stream)
{
    (void)hipGetLastError();
    if (!tr || n_samples < 1 || !game_idx || !t || !absorbed || !out_obs || !out_actions || !out_value || !out_reward ||
        !out_policy)
        return fail(MZ_ERR_ARG, "mz_make_targets: null/empty argument");
    if (!tr->game_offset || !tr->observations || !tr->actions || !tr->rewards || !tr->to_plays || !tr->root_values ||
        !tr->probs || tr->n_games < 1 || tr->obs_size < 1 || tr->n_actions < 1)
        return fail(MZ_ERR_ARG, "mz_make_targets: incomplete trajectory store");
    if (td_steps < 0 || unroll_steps < 0 || n_stack < 1) return fail(MZ_ERR_ARG, "mz_make_targets: bad step counts");
    if (gamma != 1.0 && !gamma_pow) return fail(MZ_ERR_ARG, "mz_make_targets: gamma_pow table required when gamma != 1");
    TrajDev d;
    d.n_games = tr->n_games; d.obs_size = tr->obs_size; d.A = tr->n_actions;
    d.game_offset = reinterpret_cast<const long long *>(tr->game_offset);
    d.observations = tr->observations; d.actions = tr->actions; d.rewards = tr->rewards; d.to_plays = tr->to_plays;
    d.root_values = tr->root_values; d.probs = tr->probs;
    d.re_root_values = tr->re_root_values; d.re_probs = tr->re_probs; d.has_re = tr->has_reanalysed;
    d.game_base = reinterpret_cast<const long long *>(tr->game_base); d.stride = tr->position_stride;
    if (tr->game_base && tr->position_stride < 1) return fail(MZ_ERR_ARG, "mz_make_targets: position_stride must be positive");
    long long n = (long long)n_samples * (unroll_steps + 1);
    hipLaunchKernelGGL(k_make_targets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d,
                       (int)n_samples, game_idx, t, absorbed, (int)td_steps, (int)unroll_steps, gamma, gamma_pow,
                       (int)n_stack, (int)stack_action, out_obs, out_actions, out_value, out_reward, out_policy);
    return launch_status("mz_make_targets");
}

extern "C" int mz_scalar_to_support(const float *x, int64_t n, int32_t support_limit, float *out, void *stream)
{
    (void)hipGetLastError();
    if (!x || !out || n < 1 || support_limit < 1) return fail(MZ_ERR_ARG, "mz_scalar_to_support: null/empty argument");
    const long long total = (long long)n * (2 * support_limit + 1);
    hipLaunchKernelGGL(k_scalar_to_support, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)n, (int)support_limit, out);
    return launch_status("mz_scalar_to_support");
}

extern "C" int mz_stats(mz_ctx *c, int64_t *out_node_visits, int32_t *out_noise_calls, void *stream)
{
    int rc = check_ctx(c, "mz_stats");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_stats: no tree");
    hipLaunchKernelGGL(k_stats, dim3((c->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d,
                       reinterpret_cast<long long *>(out_node_visits), out_noise_calls);
    return launch_status("mz_stats");
}

static int device_error(mz_ctx *c, hipStream_t st, const char *fn)
{
    int32_t e = 0;
    HIPCHK(hipMemcpyAsync(&e, c->d.err, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (e) {
        (void)hipMemsetAsync(c->d.err, 0, 4, st);
        return fail(MZ_ERR_DEVICE, std::string(fn) + (e == 2 ? ": a search made more PUCT selection calls than the noise buffer holds rows"
                                                              : ": a kernel reported an inconsistent selection state"));
    }
    return MZ_OK;
}

extern "C" int mz_check(mz_ctx *c, void *stream)
{
    int rc = check_ctx(c, "mz_check");
    if (rc) return rc;
    return device_error(c, (hipStream_t)stream, "mz_check");
}

extern "C" int mz_tree_export(mz_ctx *c, int32_t b, mz_tree_host *o, void *stream)
{
    int rc = check_ctx(c, "mz_tree_export");
    if (rc) return rc;
    if (!o || b < 0 || b >= c->B) return fail(MZ_ERR_ARG, "mz_tree_export: tree index out of range");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = device_error(c, st, "mz_tree_export"))) return rc;
    const Layout &L = c->d.L;
    const int A = L.A, N = L.N, H = L.H;
    std::vector<char> buf((size_t)L.tree_stride);
    HIPCHK(hipMemcpyAsync(buf.data(), c->d.nodes + (long long)b * L.tree_stride, buf.size(), hipMemcpyDeviceToHost, st));
    if (o->hidden_states && H > 0)
        HIPCHK(hipMemcpyAsync(o->hidden_states, c->d.hidden + (long long)b * N * H, (size_t)N * H * 4,
                              hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int n = 0; n < N; n++) {
        const char *r = buf.data() + (size_t)n * L.rec;
        bool live = n <= c->n_expanded;
        NodeHdr h;
        memcpy(&h, r, sizeof h);
        o->node_visits[n] = live ? h.visits : 0;
        o->node_values[n] = live ? h.value : 0.0;
        o->node_raw_values[n] = live ? h.raw : 0.0;
        o->parents[n] = live ? h.parent : -1;
        o->action_from_parent[n] = live ? h.afp : -1;
        o->to_play[n] = live ? h.to_play : 0;
        for (int a = 0; a < A; a++) {
            size_t k = (size_t)n * A + a;
            if (live) {
                memcpy(&o->children_values[k], r + L.off_cval + 8 * a, 8);
                memcpy(&o->children_rewards[k], r + L.off_crew + 8 * a, 8);
                if (L.idx16) {
                    int16_t v16, i16;
                    memcpy(&v16, r + L.off_cvis + 2 * a, 2);
                    memcpy(&i16, r + L.off_cidx + 2 * a, 2);
                    o->children_visits[k] = v16; o->children_index[k] = i16;
                } else {
                    memcpy(&o->children_visits[k], r + L.off_cvis + 4 * a, 4);
                    memcpy(&o->children_index[k], r + L.off_cidx + 4 * a, 4);
                }
                if (L.prior_f32) memcpy((float *)o->children_prior_logits + k, r + L.off_prior + 4 * a, 4);
                else memcpy((double *)o->children_prior_logits + k, r + L.off_prior + 8 * a, 8);
            } else {
                o->children_values[k] = 0; o->children_rewards[k] = 0;
                o->children_visits[k] = 0; o->children_index[k] = -1;
                if (L.prior_f32) ((float *)o->children_prior_logits)[k] = 0;
                else ((double *)o->children_prior_logits)[k] = 0;
            }
        }
        if (!live && o->hidden_states && H > 0) memset(o->hidden_states + (size_t)n * H, 0, (size_t)H * 4);
    }
    return MZ_OK;
}

extern "C" int mz_tree_import(mz_ctx *c, int32_t b, const mz_tree_host *in, const double *root_legal_h,
                              const double *gumbel_h, void *stream)
{
    int rc = check_ctx(c, "mz_tree_import");
    if (rc) return rc;
    if (!in || !root_legal_h || b < 0 || b >= c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_tree_import: bad argument");
    if (c->cfg.gumbel && !gumbel_h) return fail(MZ_ERR_ARG, "mz_tree_import: Gumbel tree needs gumbel noise");
    hipStream_t st = (hipStream_t)stream;
    const Layout &L = c->d.L;
    const int A = L.A, N = L.N, H = L.H;
    std::vector<char> buf((size_t)L.tree_stride, 0);
    int live = 0;
    for (int n = 0; n < N; n++) {
        char *r = buf.data() + (size_t)n * L.rec;
        NodeHdr h;
        h.visits = in->node_visits[n]; h.parent = in->parents[n]; h.afp = in->action_from_parent[n];
        h.to_play = in->to_play[n]; h.value = in->node_values[n]; h.raw = in->node_raw_values[n];
        if (h.visits > 0) live = n;
        memcpy(r, &h, sizeof h);
        for (int a = 0; a < A; a++) {
            size_t k = (size_t)n * A + a;
            memcpy(r + L.off_cval + 8 * a, &in->children_values[k], 8);
            memcpy(r + L.off_crew + 8 * a, &in->children_rewards[k], 8);
            if (L.idx16) {
                int16_t v16 = (int16_t)in->children_visits[k], i16 = (int16_t)in->children_index[k];
                memcpy(r + L.off_cvis + 2 * a, &v16, 2);
                memcpy(r + L.off_cidx + 2 * a, &i16, 2);
            } else {
                memcpy(r + L.off_cvis + 4 * a, &in->children_visits[k], 4);
                memcpy(r + L.off_cidx + 4 * a, &in->children_index[k], 4);
            }
            if (L.prior_f32) memcpy(r + L.off_prior + 4 * a, (const float *)in->children_prior_logits + k, 4);
            else memcpy(r + L.off_prior + 8 * a, (const double *)in->children_prior_logits + k, 8);
        }
    }
    unsigned long long mask = 0;
    int n_legal = 0;
    for (int a = 0; a < A; a++) if (root_legal_h[a] != 0) { mask |= 1ull << a; n_legal++; }
    int32_t m = c->cfg.max_considered_actions < n_legal ? c->cfg.max_considered_actions : n_legal;
    HIPCHK(hipMemcpyAsync(c->d.nodes + (long long)b * L.tree_stride, buf.data(), buf.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d.legal + b, &mask, 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d.m_considered + b, &m, 4, hipMemcpyHostToDevice, st));
    if (gumbel_h) HIPCHK(hipMemcpyAsync(c->d.gumbel + (long long)b * A, gumbel_h, (size_t)A * 8, hipMemcpyHostToDevice, st));
    if (in->hidden_states && H > 0)
        HIPCHK(hipMemcpyAsync(c->d.hidden + (long long)b * N * H, in->hidden_states, (size_t)N * H * 4,
                              hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    if (b >= c->B) { c->B = b + 1; c->d.B = c->B; }
    if (live > c->n_expanded) c->n_expanded = live;
    return MZ_OK;
}

static int node_op(mz_ctx *c, int32_t b, int32_t node, int depth, int which, const double *uniform_h,
                   double *out_vec_h, int32_t *out_action_h, void *stream, const char *fn)
{
    int rc = check_ctx(c, fn);
    if (rc) return rc;
    if (b < 0 || b >= c->B || node < 0 || node >= c->d.L.N) return fail(MZ_ERR_ARG, std::string(fn) + ": tree/node out of range");
    hipStream_t st = (hipStream_t)stream;
    const int A = c->d.L.A;
    double *u_d = nullptr;
    if (uniform_h) {
        u_d = c->node_tmp;
        HIPCHK(hipMemcpyAsync(u_d, uniform_h, (size_t)A * 8, hipMemcpyHostToDevice, st));
    }
    double *vec_d = c->node_tmp;
    dispatch_shape(c->rl, c->rl, c->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) {
        hipLaunchKernelGGL((k_node_op<decltype(L_)::value, decltype(R_)::value, decltype(G_)::value>), dim3(1), dim3(64),
                           0, st, c->d, (int)b, (int)node, depth, which, u_d, vec_d, c->node_act);
    });
    if ((rc = launch_status(fn))) return rc;
    if (out_vec_h) HIPCHK(hipMemcpyAsync(out_vec_h, vec_d, (size_t)A * 8, hipMemcpyDeviceToHost, st));
    if (out_action_h) HIPCHK(hipMemcpyAsync(out_action_h, c->node_act, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MZ_OK;
}

extern "C" int mz_node_qtransform(mz_ctx *c, int32_t b, int32_t node, int32_t which, double *out_h, void *stream)
{
    if (which < 0 || which > 2 || !out_h) return fail(MZ_ERR_ARG, "mz_node_qtransform: bad argument");
    if (c && which > 0 && !c->cfg.gumbel)
        return fail(MZ_ERR_ARG, "mz_node_qtransform: completed-Q transforms need a Gumbel context (float32 priors)");
    return node_op(c, b, node, 1, which, nullptr, out_h, nullptr, stream, "mz_node_qtransform");
}

extern "C" int mz_node_select(mz_ctx *c, int32_t b, int32_t node, int32_t depth, const double *uniform_h,
                              int32_t *out_action_h, void *stream)
{
    if (!out_action_h) return fail(MZ_ERR_ARG, "mz_node_select: null output");
    return node_op(c, b, node, depth, 3, uniform_h, nullptr, out_action_h, stream, "mz_node_select");
}

extern "C" int mz_node_backward(mz_ctx *c, int32_t b, int32_t leaf, int32_t to_play, void *stream)
{
    int rc = check_ctx(c, "mz_node_backward");
    if (rc) return rc;
    if (b < 0 || b >= c->B || leaf < 0 || leaf >= c->d.L.N) return fail(MZ_ERR_ARG, "mz_node_backward: tree/node out of range");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_node_backward, dim3(1), dim3(64), 0, st, c->d, (int)b, (int)leaf, (int)to_play);
    if ((rc = launch_status("mz_node_backward"))) return rc;
    return device_error(c, st, "mz_node_backward");
}

extern "C" int mz_set_network(mz_ctx *c, const int32_t *meta_h, int32_t n_meta, const float *data_h,
                              int32_t n_data, const int32_t *io_h)
{
    int rc = check_ctx(c, "mz_set_network");
    if (rc) return rc;
    if (!meta_h || !data_h || !io_h || n_meta < 8 || n_data < 1) return fail(MZ_ERR_ARG, "mz_set_network: null/empty program");
    if (meta_h[3] != n_data || meta_h[6] + 4 * meta_h[2] != n_meta) return fail(MZ_ERR_ARG, "mz_set_network: inconsistent program header");
    if (io_h[6] != c->cfg.hidden_size || io_h[7] != c->cfg.n_actions)
        return fail(MZ_ERR_ARG, "mz_set_network: network hidden size / action count differ from the context's");
    if (io_h[10] < 1 || io_h[6] % io_h[10]) return fail(MZ_ERR_ARG, "mz_set_network: hidden size not divisible by normalisation groups");
    // every row must read and write inside the scratch
    const int32_t *lv = meta_h + meta_h[5], *rows = meta_h + meta_h[6];
    for (int l = 0; l < meta_h[0]; l++)
        for (int r = 0; r < lv[4 * l]; r++) {
            const int32_t *d = rows + 4 * (lv[4 * l + 3] + r);
            if (d[0] < 0 || d[0] + lv[4 * l + 1] > meta_h[1] || d[1] < 0 || d[1] >= meta_h[1] || d[3] >= meta_h[1])
                return fail(MZ_ERR_ARG, "mz_set_network: a row addresses outside the activation scratch");
        }
    if (c->net_meta_d) { (void)hipFree(c->net_meta_d); c->net_meta_d = nullptr; }
    if (c->net_data_d) { (void)hipFree(c->net_data_d); c->net_data_d = nullptr; }
    HIPCHK(hipMalloc(&c->net_meta_d, (size_t)n_meta * 4));
    HIPCHK(hipMalloc(&c->net_data_d, (size_t)n_data * 4));
    HIPCHK(hipMemcpy(c->net_meta_d, meta_h, (size_t)n_meta * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->net_data_d, data_h, (size_t)n_data * 4, hipMemcpyHostToDevice));
    c->net.meta = reinterpret_cast<const int32_t *>(c->net_meta_d);
    c->net.data = reinterpret_cast<const float *>(c->net_data_d);
    c->net.n_meta = n_meta; c->net.n_data = n_data;
    memcpy(c->net.io, io_h, sizeof c->net.io);
    c->net_scratch = meta_h[1];
    // at most 2 levels of at most 64 rows over at most 16 inputs, one normalisation group of at most 16
    // hidden values, supports of at most 32 bins: what dense_run_cached and epilogue_wave_small serve
    c->net_small = meta_h[0] <= 2 && io_h[10] == 1 && io_h[6] <= 16 && io_h[9] <= 32;
    for (int l = 0; l < meta_h[0]; l++)
        if (lv[4 * l] > 64 || lv[4 * l + 1] > 16) c->net_small = false;
    // hidden states in registers: two levels without residual inputs, every first-level row reads the network
    // input (hidden state, then the encoded action), the second level produces exactly the four outputs
    const int H = io_h[6], A = io_h[7], asz = io_h[8], sup = io_h[9];
    bool tiny = c->net_small && meta_h[0] == 2 && (H == 4 || H == 8) && A == 2 && asz == 2 && sup <= 24 && io_h[1] == io_h[0] + H &&
                lv[4] == H + 2 * sup + A;
    if (tiny) {
        for (int r = 0; r < lv[0]; r++) {
            const int32_t *d = rows + 4 * (lv[3] + r);
            if (d[0] != io_h[0] || d[3] >= 0) tiny = false;
        }
        int seen = 0;
        for (int r = 0; r < lv[4]; r++) {
            const int32_t *d = rows + 4 * (lv[7] + r);
            if (d[3] >= 0) tiny = false;
            const int y = d[1];
            seen += (y >= io_h[2] && y < io_h[2] + H) + (y >= io_h[3] && y < io_h[3] + sup) + (y >= io_h[4] && y < io_h[4] + A) +
                    (y >= io_h[5] && y < io_h[5] + sup);
        }
        if (seen != lv[4]) tiny = false;
    }
    c->net_tiny = tiny;
    return MZ_OK;
}

static int check_program(const int32_t *meta_h, int32_t n_meta, const float *data_h, int32_t n_data, const char *fn)
{
    if (!meta_h || !data_h || n_meta < 8 || n_data < 1) return fail(MZ_ERR_ARG, std::string(fn) + ": null/empty program");
    if (meta_h[3] != n_data || meta_h[6] + 4 * meta_h[2] != n_meta) return fail(MZ_ERR_ARG, std::string(fn) + ": inconsistent program header");
    const int32_t *lv = meta_h + meta_h[5], *rows = meta_h + meta_h[6];
    for (int l = 0; l < meta_h[0]; l++)
        for (int r = 0; r < lv[4 * l]; r++) {
            const int32_t *d = rows + 4 * (lv[4 * l + 3] + r);
            if (d[0] < 0 || d[0] + lv[4 * l + 1] > meta_h[1] || d[1] < 0 || d[1] >= meta_h[1] || d[3] >= meta_h[1])
                return fail(MZ_ERR_ARG, std::string(fn) + ": a row addresses outside the activation scratch");
        }
    return MZ_OK;
}

extern "C" int mz_set_root_network(mz_ctx *c, const int32_t *rep_meta_h, int32_t rep_n_meta, const float *rep_data_h,
                                   int32_t rep_n_data, const int32_t *pred_meta_h, int32_t pred_n_meta,
                                   const float *pred_data_h, int32_t pred_n_data, const int32_t *io_h)
{
    int rc = check_ctx(c, "mz_set_root_network");
    if (rc) return rc;
    if (!io_h) return fail(MZ_ERR_ARG, "mz_set_root_network: null io");
    if ((rc = check_program(rep_meta_h, rep_n_meta, rep_data_h, rep_n_data, "mz_set_root_network"))) return rc;
    if ((rc = check_program(pred_meta_h, pred_n_meta, pred_data_h, pred_n_data, "mz_set_root_network"))) return rc;
    if (io_h[7] < 1 || c->cfg.hidden_size % io_h[7]) return fail(MZ_ERR_ARG, "mz_set_root_network: bad normalisation groups");
    for (void *&p : c->root_bufs) if (p) { (void)hipFree(p); p = nullptr; }
    const void *src[4] = {rep_meta_h, rep_data_h, pred_meta_h, pred_data_h};
    size_t n[4] = {(size_t)rep_n_meta, (size_t)rep_n_data, (size_t)pred_n_meta, (size_t)pred_n_data};
    for (int i = 0; i < 4; i++) {
        HIPCHK(hipMalloc(&c->root_bufs[i], n[i] * 4));
        HIPCHK(hipMemcpy(c->root_bufs[i], src[i], n[i] * 4, hipMemcpyHostToDevice));
    }
    RootNetDev &r = c->root;
    r.rep_meta = (const int32_t *)c->root_bufs[0]; r.rep_data = (const float *)c->root_bufs[1];
    r.pred_meta = (const int32_t *)c->root_bufs[2]; r.pred_data = (const float *)c->root_bufs[3];
    r.rep_n_meta = rep_n_meta; r.rep_n_data = rep_n_data; r.pred_n_meta = pred_n_meta; r.pred_n_data = pred_n_data;
    r.rep_scratch = rep_meta_h[1]; r.pred_scratch = pred_meta_h[1];
    r.obs_off = io_h[0]; r.raw_off = io_h[1]; r.hid_off = io_h[2]; r.pol_off = io_h[3]; r.val_off = io_h[4];
    r.obs_size = io_h[5]; r.support = io_h[6]; r.ngroups = io_h[7];
    return MZ_OK;
}

extern "C" int mz_search_begin_net(mz_ctx *c, int32_t B, const float *obs, const uint8_t *legal,
                                   const int32_t *to_play, const double *dirichlet, double frac,
                                   const double *gumbel, uint64_t tree_id0, float *out_logits,
                                   double *out_value, void *stream)
{
    int rc = check_ctx(c, "mz_search_begin_net");
    if (rc) return rc;
    if (!c->root.rep_meta) return fail(MZ_ERR_STATE, "mz_search_begin_net: no root network (call mz_set_root_network first)");
    if (B < 1 || B > c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_search_begin_net: B outside 1..max_batch");
    if (!obs || !legal || !to_play) return fail(MZ_ERR_ARG, "mz_search_begin_net: null argument");
    if (c->cfg.gumbel ? !gumbel : !dirichlet) return fail(MZ_ERR_ARG, "mz_search_begin_net: root noise missing");
    const RootNetDev &r = c->root;
    size_t lds = ((size_t)r.rep_n_data + r.rep_n_meta + r.pred_n_data + r.pred_n_meta +
                  (size_t)(MZ_BLOCK / c->lanes) * (r.rep_scratch + r.pred_scratch)) * 4;
    if (lds > 160 * 1024) return fail(MZ_ERR_ARG, "mz_search_begin_net: root network does not fit in LDS");
    c->B = B; c->d.B = B; c->d.tree_id0 = tree_id0;
    c->n_expanded = 0; c->selected = 0;
    hipError_t attr_err = hipSuccess;
    dispatch_shape(c->lanes, c->rl, c->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) {
        auto kern = k_begin_net<decltype(L_)::value, decltype(R_)::value, decltype(G_)::value>;
        attr_err = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (size_t)lds);
        hipLaunchKernelGGL(kern, dim3(grid_for(B, decltype(L_)::value)), dim3(MZ_BLOCK), lds, (hipStream_t)stream,
                           c->d, r, obs, legal, to_play, dirichlet, frac, gumbel,
                           (int)c->cfg.max_considered_actions, out_logits, out_value);
    });
    if (attr_err != hipSuccess) return fail(MZ_ERR_HIP, "mz_search_begin_net: could not raise the dynamic LDS limit");
    return launch_status("mz_search_begin_net");
}

#ifdef MZ_STAMPS
// Diagnostic build only: copies the first n_bytes of the path buffer (where k_search_net parks its phase stamps).
extern "C" int mz_debug_read_path(mz_ctx *c, void *out_h, size_t n_bytes)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_h, c->d.path, n_bytes, hipMemcpyDeviceToHost));
    return MZ_OK;
}
#endif

// One program in matrix-core layout (muzero_amd/packing.py pack_mfma) checked and uploaded into `slot`: the recurrent
// program of the searches, or the representation / prediction program of the initial inference.
struct MfmaSlot {
    MfmaNet *mnet;
    void **bufs;                     // [10], see mz_ctx::mnet_bufs
    int *max_k16;
    std::vector<int> *level_job0;
    bool *tile_fits;
};

static int upload_mfma(mz_ctx *c, const char *fn, const MfmaSlot &slot, const int32_t *steps_h, int32_t n_steps, int32_t n_levels,
                       const int32_t *sched_h, int32_t n_sched, const float *wfrag_h, int64_t n_w,
                       const float *bias_h, int32_t n_b, const int32_t *ctinfo_h, int32_t n_jobs,
                       const int32_t *kidx_h, int32_t n_kidx, int32_t stride, int32_t plane_major,
                       const int32_t *io_h)
{
    if (!steps_h || !sched_h || !wfrag_h || !bias_h || !ctinfo_h || !kidx_h || !io_h || n_steps < 1 ||
        n_sched < 4 || n_levels < 1 || n_jobs < 1 || n_kidx < 1)
        return fail(MZ_ERR_ARG, std::string(fn) + ": null/empty program");
    if (io_h[6] != c->cfg.hidden_size || io_h[7] != c->cfg.n_actions)
        return fail(MZ_ERR_ARG, std::string(fn) + ": network hidden size / action count differ from the context's");
    if (io_h[10] < 1 || io_h[6] % io_h[10]) return fail(MZ_ERR_ARG, std::string(fn) + ": hidden size not divisible by normalisation groups");
    if (plane_major && io_h[10] < 8) return fail(MZ_ERR_ARG, std::string(fn) + ": plane-major storage needs at least 8 normalisation groups (channels)");
    if (stride % 8 != 4) return fail(MZ_ERR_ARG, std::string(fn) + ": stride must be 4 times an odd number");
    if (stride > (1 << 20)) return fail(MZ_ERR_ARG, std::string(fn) + ": unreasonable activation row");
    if (n_w < 1 || n_w > (1 << 28) || n_b < 1) return fail(MZ_ERR_ARG, std::string(fn) + ": weight or bias array out of range");
    for (int s = 0; s < n_steps; s++) {
        const int32_t *st = steps_h + 12 * s;
        int k16 = st[2], nct = st[9];
        if (st[3] < 0 || st[3] + k16 * 16 > stride || st[4] < 0 || st[4] + nct * 16 > stride || st[5] + nct * 16 > stride ||
            st[8] < 0 || st[8] + nct * 16 > n_b || st[10] < 0 || st[10] + nct > n_jobs)
            return fail(MZ_ERR_ARG, std::string(fn) + ": a step addresses outside its buffers");
        for (int ct = 0; ct < nct; ct++) {
            const int32_t *job = ctinfo_h + 3 * (st[10] + ct);
            if (job[0] < 0 || job[2] < 0 || (int64_t)job[0] + (int64_t)job[2] * 256 > n_w || job[1] < 0 || job[1] + job[2] > n_kidx)
                return fail(MZ_ERR_ARG, std::string(fn) + ": a column tile addresses outside the weight blocks");
            // a tile's blocks: each inside the layer's inputs, strictly ascending (hence at most k16 of them) - the level
            // kernel sizes its rounds of weight loads by the widest layer and would drop the blocks of a longer list
            for (int i = 0; i < job[2]; i++)
                if (kidx_h[job[1] + i] < 0 || kidx_h[job[1] + i] >= k16 || (i > 0 && kidx_h[job[1] + i] <= kidx_h[job[1] + i - 1]))
                    return fail(MZ_ERR_ARG, std::string(fn) + ": a column tile's weight blocks must be distinct, ascending and inside its layer's inputs");
        }
    }
    // the two per-wave schedules: job ranges in order, at most 64 jobs per wave, every descriptor inside its buffers
    bool sched_ok[2] = {false, false};   // an empty schedule = "this tile size is not served" (muzero_amd/packing.py)
    for (int k = 0; k < 2; k++) {
        const int nw = k == 0 ? 8 : 4;
        const int64_t o_start = sched_h[2 * k], o_desc = sched_h[2 * k + 1];
        const int64_t lim_desc = k == 0 ? sched_h[2] : n_sched;
        if (o_start < 4 || o_start + (int64_t)nw * (n_levels + 1) > o_desc || o_desc > lim_desc || lim_desc > n_sched)
            return fail(MZ_ERR_ARG, std::string(fn) + ": inconsistent schedule offsets");
        const int32_t *ws = sched_h + o_start, *wd = sched_h + o_desc;
        const int64_t n_wd = (lim_desc - o_desc) / MZ_DESC_INTS;
        sched_ok[k] = n_wd > 0;
        int64_t prev = 0;
        for (int w = 0; w < nw; w++) {
            for (int l = 0; l <= n_levels; l++) {
                const int64_t v = ws[w * (n_levels + 1) + l];
                if (v < prev || v > n_wd) return fail(MZ_ERR_ARG, std::string(fn) + ": schedule job ranges out of order");
                prev = v;
            }
            if (ws[w * (n_levels + 1) + n_levels] - ws[w * (n_levels + 1)] > 64)
                return fail(MZ_ERR_ARG, std::string(fn) + ": more than 64 jobs for one wave");
        }
        for (int64_t j = 0; j < n_wd; j++) {
            const int32_t *q = wd + MZ_DESC_INTS * j;
            const bool ok = q[0] >= 0 && q[0] % 4 == 0 && q[1] >= 0 && (int64_t)q[0] / 4 + (int64_t)q[1] * 256 <= n_w &&
                            q[2] >= 0 && q[2] <= 2 && (k == 0 || q[2] == 1) && q[3] >= 0 && q[3] + 16 <= stride &&
                            q[4] >= 0 && q[4] + 16 <= stride && q[5] + 16 <= stride && q[6] >= 0 && q[6] <= 2 &&
                            q[7] >= 0 && q[8] >= 1 && q[8] <= 16 && q[7] + q[8] <= n_b && q[9] >= 0 && q[9] + q[1] <= n_kidx;
            if (!ok) return fail(MZ_ERR_ARG, std::string(fn) + ": bad schedule entry");
            for (int i = 0; i < q[1] && i < 12; i++) {
                const int kb = (q[10 + i / 4] >> (8 * (i % 4))) & 0xff;
                if (q[3] + kb * 16 + 16 > stride || kb != kidx_h[q[9] + i])
                    return fail(MZ_ERR_ARG, std::string(fn) + ": a schedule entry reads outside the activation row");
            }
        }
    }
    // Block descriptors of the level kernel, level by level: a block takes up to MZ_LVL_WAVES column tiles of one step.
    // Workgroups go to the compute units in turn by their linear index (tree tile fast, block slow), the blocks of one
    // tile to the units of one group of units/T (T tiles of trees, when T divides the units): blocks j and j + units/T
    // of a level share a unit, and a level lasts as long as its most loaded unit.  Where a level has more blocks than
    // such a group has units but fewer than twice as many (13 on 8: five units hold 16 column tiles, three hold 8 or
    // fewer) its steps are cut into two blocks a unit, as evenly as the steps allow, and ordered so that a heavy block
    // shares its unit with a light one (16 blocks of 5..7 tiles: 13 per unit at most) - the matrix phase of such a
    // level is as long as the most loaded unit's.  With more than two blocks a unit the plain cut stays: every further
    // block copies its tile's inputs once more, and that was measured to cost more than the evening-out returns
    // (13 blocks on 4 units cut into 16: 2.5 % slower; on 2 units into 14: 0.6 %).
    std::vector<int32_t> desc;
    std::vector<int> job0(1, 0);
    int n_cus = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (n_cus < 1) n_cus = 256;
    }
    const int n_tiles_max = (c->cfg.max_batch + 31) / 32;
    const int groups = (n_tiles_max <= n_cus && n_cus % n_tiles_max == 0) ? n_cus / n_tiles_max : 1;
    struct LvlJob { int32_t q[MZ_LVL_DESC]; long long cost; };
    for (int l = 0; l < n_levels; l++) {
        std::vector<const int32_t *> lsteps;
        int plain_jobs = 0;
        for (int s = 0; s < n_steps; s++) {
            const int32_t *st = steps_h + 12 * s;
            if (st[0] != l) continue;
            if (l == 0 && (st[3] < io_h[0] || st[3] + st[2] * 16 > stride))
                return fail(MZ_ERR_ARG, std::string(fn) + ": a first-level step reads outside the network inputs");
            if ((size_t)(st[2] * 16 + 4) * 32 * 4 > 160 * 1024)
                return fail(MZ_ERR_ARG, std::string(fn) + ": a layer's inputs for 32 trees exceed LDS");
            lsteps.push_back(st);
            plain_jobs += (st[9] + MZ_LVL_WAVES - 1) / MZ_LVL_WAVES;
        }
        // blocks per step: the plain count, or - see above - shares of two blocks for each of the group's units
        std::vector<int> njobs(lsteps.size());
        for (size_t i = 0; i < lsteps.size(); i++) njobs[i] = (lsteps[i][9] + MZ_LVL_WAVES - 1) / MZ_LVL_WAVES;
        const bool even_out = groups > 1 && plain_jobs > groups && plain_jobs < 2 * groups && !getenv("MZ_LVL_PLAIN");
        if (even_out) {
            const int target = 2 * groups;
            long long tiles_total = 0;
            for (const int32_t *st : lsteps) tiles_total += st[9];
            int given = 0;
            std::vector<double> rest(lsteps.size());
            for (size_t i = 0; i < lsteps.size(); i++) {
                const double share = (double)lsteps[i][9] * target / (double)tiles_total;
                njobs[i] = std::max(njobs[i], std::max(1, (int)share));      // never fewer than the plain cut (8 tiles a block at most)
                njobs[i] = std::min(njobs[i], (int)lsteps[i][9]);
                rest[i] = share - njobs[i];
                given += njobs[i];
            }
            while (given < target) {   // largest remainder first
                size_t best = lsteps.size();
                for (size_t i = 0; i < lsteps.size(); i++)
                    if (njobs[i] < lsteps[i][9] && (best == lsteps.size() || rest[i] > rest[best])) best = i;
                if (best == lsteps.size()) break;
                njobs[best]++; rest[best] -= 1.0; given++;
            }
        }
        std::vector<LvlJob> jobs;
        for (size_t i = 0; i < lsteps.size(); i++) {
            const int32_t *st = lsteps[i];
            for (int k = 0; k < njobs[i]; k++) {
                const int ct = (int)((long long)st[9] * k / njobs[i]), ct_end = (int)((long long)st[9] * (k + 1) / njobs[i]);
                const int n = ct_end - ct;
                if (n < 1 || n > MZ_LVL_WAVES) return fail(MZ_ERR_STATE, std::string(fn) + ": internal error (level block width)");
                // everything a block needs, flattened: step fields + per-wave weight block lists
                LvlJob j{{st[2] * 16, st[3], st[1], st[4], st[5], st[6], st[8], n, ct, 1}, 0};
                for (int w = 0; w < n; w++) {
                    if (st[10] + ct + w >= n_jobs) return fail(MZ_ERR_ARG, std::string(fn) + ": column tile table too short");
                    const int32_t *cj = ctinfo_h + 3 * (st[10] + ct + w);
                    if (cj[1] < 0 || cj[2] < 0 || cj[1] + cj[2] > n_kidx)
                        return fail(MZ_ERR_ARG, std::string(fn) + ": column tile reads outside the block index list");
                    j.q[10 + 3 * w] = cj[0] * 4; j.q[11 + 3 * w] = cj[2]; j.q[12 + 3 * w] = cj[1];
                    for (int i2 = 0; i2 < cj[2]; i2++) if (kidx_h[cj[1] + i2] != i2) j.q[9] = 0;
                    j.cost += cj[2];
                }
                jobs.push_back(j);
            }
        }
        if (even_out) {   // heaviest first, row by row over the group's units, every other row backwards
            std::stable_sort(jobs.begin(), jobs.end(), [](const LvlJob &x, const LvlJob &y) { return x.cost > y.cost; });
            for (size_t r = 1; r * groups < jobs.size(); r += 2)
                std::reverse(jobs.begin() + r * groups, jobs.begin() + std::min(jobs.size(), (r + 1) * groups));
        }
        for (const LvlJob &j : jobs) desc.insert(desc.end(), j.q, j.q + MZ_LVL_DESC);
        job0.push_back((int)(desc.size() / MZ_LVL_DESC));
    }
    for (int s2 = 0; s2 < n_steps; s2++)
        if (steps_h[12 * s2] < 0 || steps_h[12 * s2] >= n_levels) return fail(MZ_ERR_ARG, std::string(fn) + ": bad level");
    for (int i = 0; i < 10; i++) if (slot.bufs[i]) { (void)hipFree(slot.bufs[i]); slot.bufs[i] = nullptr; }
    // the slot describes no program until every buffer of the new one is in place: a failure below leaves it empty
    // (every user checks `steps`), not pointing at freed memory
    memset(slot.mnet, 0, sizeof *slot.mnet);
    slot.level_job0->clear();
    *slot.tile_fits = false;
    *slot.max_k16 = 1;
    for (int s2 = 0; s2 < n_steps; s2++) *slot.max_k16 = std::max(*slot.max_k16, (int)steps_h[12 * s2 + 2]);
    slot.mnet->sched_ok8 = sched_ok[0]; slot.mnet->sched_ok4 = sched_ok[1];
    *slot.tile_fits = ((size_t)stride * 16 * 4 <= 160 * 1024 && sched_ok[1]) ||
                        ((size_t)stride * 32 * 4 <= 160 * 1024 && sched_ok[0]);
    HIPCHK(hipMalloc(&slot.bufs[9], desc.size() * 4));
    HIPCHK(hipMemcpy(slot.bufs[9], desc.data(), desc.size() * 4, hipMemcpyHostToDevice));
    const size_t act_rows = ((size_t)c->cfg.max_batch + 31) / 32 * 32;
    HIPCHK(hipMalloc(&slot.bufs[6], act_rows * stride * 4));
    HIPCHK(hipMemset(slot.bufs[6], 0, act_rows * stride * 4));
    const void *src[4] = {steps_h, sched_h, wfrag_h, bias_h};
    size_t n[4] = {(size_t)n_steps * 12, (size_t)n_sched, (size_t)n_w, (size_t)n_b};
    const int which[4] = {0, 1, 3, 4};
    for (int i = 0; i < 4; i++) {
        HIPCHK(hipMalloc(&slot.bufs[which[i]], n[i] * 4));
        HIPCHK(hipMemcpy(slot.bufs[which[i]], src[i], n[i] * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc(&slot.bufs[7], (size_t)n_jobs * 12));
    HIPCHK(hipMemcpy(slot.bufs[7], ctinfo_h, (size_t)n_jobs * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&slot.bufs[8], (size_t)n_kidx * 4));
    HIPCHK(hipMemcpy(slot.bufs[8], kidx_h, (size_t)n_kidx * 4, hipMemcpyHostToDevice));
    MfmaNet &m = *slot.mnet;
    m.steps = (const int32_t *)slot.bufs[0];
    const int32_t *sched_d = (const int32_t *)slot.bufs[1];
    m.wstart8 = sched_d + sched_h[0]; m.wdesc8 = sched_d + sched_h[1];
    m.wstart4 = sched_d + sched_h[2]; m.wdesc4 = sched_d + sched_h[3];
    m.w_bytes = (int)(n_w * 4); m.b_bytes = n_b * 4;
    m.wfrag = (const float *)slot.bufs[3];
    m.bias = (const float *)slot.bufs[4];
    m.ctinfo = (const int32_t *)slot.bufs[7]; m.kidx = (const int32_t *)slot.bufs[8];
    m.plane_major = plane_major ? 1 : 0;
    m.n_steps = n_steps; m.n_levels = n_levels; m.stride = stride;
    memcpy(m.io, io_h, sizeof m.io);
    *slot.level_job0 = job0;
    return MZ_OK;
}


extern "C" int mz_set_network_mfma(mz_ctx *c, const int32_t *steps_h, int32_t n_steps, int32_t n_levels,
                                   const int32_t *sched_h, int32_t n_sched, const float *wfrag_h, int64_t n_w,
                                   const float *bias_h, int32_t n_b, const int32_t *ctinfo_h, int32_t n_jobs,
                                   const int32_t *kidx_h, int32_t n_kidx, int32_t stride, int32_t plane_major,
                                   const int32_t *io_h)
{
    int rc = check_ctx(c, "mz_set_network_mfma");
    if (rc) return rc;
    const MfmaSlot slot{&c->mnet, c->mnet_bufs, &c->mnet_max_k16, &c->mnet_level_job0, &c->mnet_tile_fits};
    rc = upload_mfma(c, "mz_set_network_mfma", slot, steps_h, n_steps, n_levels, sched_h, n_sched, wfrag_h, n_w, bias_h, n_b, ctinfo_h, n_jobs,
                     kidx_h, n_kidx, stride, plane_major, io_h);
    c->chain_ok = -1; c->chain_G = 0;   // the chained search's work lists belong to the program just replaced
    if (rc == MZ_OK) {
        c->mnet_steps_h.assign(steps_h, steps_h + (size_t)n_steps * 12);
        c->mnet_ctinfo_h.assign(ctinfo_h, ctinfo_h + (size_t)n_jobs * 3);
        c->mnet_kidx_h.assign(kidx_h, kidx_h + n_kidx);
    } else if (!c->mnet.steps) {
        c->mnet_steps_h.clear();   // the upload failed after the old program was released
    }
    return rc;
}

extern "C" int mz_set_root_network_mfma(mz_ctx *c, int32_t part, const int32_t *steps_h, int32_t n_steps, int32_t n_levels,
                                        const int32_t *sched_h, int32_t n_sched, const float *wfrag_h, int64_t n_w,
                                        const float *bias_h, int32_t n_b, const int32_t *ctinfo_h, int32_t n_jobs,
                                        const int32_t *kidx_h, int32_t n_kidx, int32_t stride, int32_t plane_major,
                                        const int32_t *io_h)
{
    int rc = check_ctx(c, "mz_set_root_network_mfma");
    if (rc) return rc;
    if (part != 0 && part != 1) return fail(MZ_ERR_ARG, "mz_set_root_network_mfma: part is 0 (representation) or 1 (prediction)");
    RootMfma &r = c->rootm[part];
    bool unused = false;
    const MfmaSlot slot{&r.mnet, r.bufs, &r.max_k16, &r.level_job0, &unused};
    return upload_mfma(c, "mz_set_root_network_mfma", slot, steps_h, n_steps, n_levels, sched_h, n_sched, wfrag_h, n_w, bias_h, n_b, ctinfo_h, n_jobs,
                       kidx_h, n_kidx, stride, plane_major, io_h);
}

// network.py:86-103 for B observations through the two uploaded root programs, level by level on the matrix cores.
extern "C" int mz_initial_inference_mfma(mz_ctx *c, int32_t B, const float *obs, float *out_logits, float *out_hidden,
                                         double *out_value, void *stream)
{
    int rc = check_ctx(c, "mz_initial_inference_mfma");
    if (rc) return rc;
    if (B < 1 || B > c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_initial_inference_mfma: B outside 1..max_batch");
    if (!obs || !out_logits || !out_hidden || !out_value) return fail(MZ_ERR_ARG, "mz_initial_inference_mfma: null argument");
    RootMfma &rep = c->rootm[0], &pred = c->rootm[1];
    if (!rep.mnet.steps || !pred.mnet.steps)
        return fail(MZ_ERR_STATE, "mz_initial_inference_mfma: upload both root programs first (mz_set_root_network_mfma)");
    const hipStream_t st = (hipStream_t)stream;
    auto run_levels = [&](RootMfma &r) -> int {
        float *acts = (float *)r.bufs[6];
        const int32_t *desc = (const int32_t *)r.bufs[9];
        for (int l = 0; l < r.mnet.n_levels; l++) {
            const int j0 = r.level_job0[l], nb = r.level_job0[l + 1] - j0;
            if (nb < 1) continue;
            HIPCHK(launch_net_level_rows((unsigned)nb, (unsigned)((B + 31) / 32), r.max_k16, st, r.mnet,
                                         desc + MZ_LVL_DESC * j0, acts, 0, l));
        }
        return MZ_OK;
    };
    launch_root_stage_obs(st, (int)B, rep.mnet, obs, (float *)rep.bufs[6]);
    if ((rc = run_levels(rep))) return rc;
    launch_root_hidden(st, (int)B, rep.mnet, (const float *)rep.bufs[6], out_hidden);
    launch_root_stage_hidden(st, (int)B, pred.mnet, out_hidden, (float *)pred.bufs[6]);
    if ((rc = run_levels(pred))) return rc;
    launch_root_outputs(st, (int)B, pred.mnet, (const float *)pred.bufs[6], out_logits, out_value);
    return launch_status("mz_initial_inference_mfma");
}

// Staging of one batched leaf evaluation: activations of the per-level path ([rows][stride] floats, rows = the batch
// rounded up to 32) and the network outputs k_expand_backup / k_backup_select consume.
struct LeafStaging {
    float *acts;
    float *logits;   // [B][A]
    double *value;   // [B]
    double *reward;  // [B]
};

static int expand_backup_net_impl(mz_ctx *c, int32_t select_next, const double *noise, int32_t noise_max_calls,
                                  void *stream, const LeafStaging &stg, bool force_by_level)
{
    int rc = MZ_OK;
    if (!c->selected) return fail(MZ_ERR_STATE, "mz_expand_backup_net: no pending selection (call mz_select first)");
    if (!c->mnet.steps) return fail(MZ_ERR_STATE, "mz_expand_backup_net: no network uploaded (call mz_set_network_mfma first)");
    int leaf = c->n_expanded + 1;
    select_next = select_next && leaf < c->cfg.n_simulations;
    if (select_next) {   // checked before anything is launched: a rejected call leaves the pending selection as it was
        if ((rc = check_noise(c, noise, noise_max_calls, "mz_expand_backup_net"))) return rc;
        if (c->cfg.noise_mode != MZ_NOISE_BUFFER) noise = nullptr;
    }
    // 32 trees per block while their activations fit in LDS, else 16; batches that would leave most CUs
    // without a tile (and networks whose tile does not fit at all) go level by level through global memory
    const int tm = ((size_t)c->mnet.stride * 32 * 4 <= 160 * 1024 && c->mnet.sched_ok8) ? 32 : 16;
    bool by_level = force_by_level || !c->mnet_tile_fits || (c->B + tm - 1) / tm < 128;
    if (const char *e = getenv("MZ_NET_PATH")) by_level = force_by_level || !c->mnet_tile_fits || strcmp(e, "level") == 0;
    if (by_level) {
        float *acts = stg.acts;
        const int32_t *desc = (const int32_t *)c->mnet_bufs[9];
        launch_stage_inputs((hipStream_t)stream, c->d, c->mnet, acts);   // the pending selection's network inputs
        for (int l = 0; l < c->mnet.n_levels; l++) {
            const int j0 = c->mnet_level_job0[l], nb = c->mnet_level_job0[l + 1] - j0;
            if (nb < 1) continue;
            HIPCHK(launch_net_level_rows((unsigned)nb, (unsigned)((c->B + 31) / 32), c->mnet_max_k16, (hipStream_t)stream,
                                         c->mnet, desc + MZ_LVL_DESC * j0, acts, 0, l));
        }
        // a whole wave per tree when the supports or the hidden state are long
        const int ep_lanes = (c->mnet.io[9] > 64 || c->mnet.io[6] / c->mnet.io[10] > 64) ? 64 : 16;
        launch_net_epilogue(ep_lanes, (hipStream_t)stream, c->d, c->mnet, leaf, acts, stg.logits, stg.value, stg.reward);
    } else {
        HIPCHK(launch_net_step(tm, (hipStream_t)stream, c->d, c->mnet, leaf, stg.logits, stg.value, stg.reward));
    }
    if ((rc = launch_status("mz_expand_backup_net"))) return rc;
    if (select_next) {
        MZ_LAUNCH(c, k_backup_select, false, c->B, 0, stream, c->d, leaf, (const float *)stg.logits,
                  (const double *)stg.value, (const double *)stg.reward, noise, (int)noise_max_calls);
    } else {
        MZ_LAUNCH(c, k_expand_backup, false, c->B, 0, stream, c->d, leaf, (const float *)stg.logits,
                  (const double *)stg.value, (const double *)stg.reward, (const float *)nullptr);
    }
    c->n_expanded = leaf;
    c->selected = select_next ? 1 : 0;
    return launch_status("mz_expand_backup_net");
}

extern "C" int mz_expand_backup_net(mz_ctx *c, int32_t select_next, const double *noise, int32_t noise_max_calls,
                                    void *stream)
{
    int rc = check_ctx(c, "mz_expand_backup_net");
    if (rc) return rc;
    const LeafStaging stg{(float *)c->mnet_bufs[6], c->leaf_logits, c->leaf_value, c->leaf_reward};
    return expand_backup_net_impl(c, select_next, noise, noise_max_calls, stream, stg, false);
}

// ---------------------------------------------------------------------------------------------------
// The same step with its staging in a fixed array of the library's device image instead of buffers from hipMalloc.
// `scratch` below is only a host-side handle of that array: its device address differs per GPU and is looked up
// once per device (scratch_for_device); host code uses the looked-up pointer everywhere.
#define MZ_STATIC_SCRATCH_BYTES (16u << 20)
static __device__ __align__(256) unsigned char scratch[MZ_STATIC_SCRATCH_BYTES];

#define MZ_MAX_DEVICES 64
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[MZ_MAX_DEVICES];

// Device address of `scratch` on the current device (set by check_ctx / mz_create), resolved on first use.
static int scratch_for_device(int device, void **out)
{
    if (device < 0 || device >= MZ_MAX_DEVICES) return fail(MZ_ERR_ARG, "static scratch: device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(MZ_ERR_HIP, std::string("static scratch: hipGetSymbolAddress failed: ") + hipGetErrorString(e));
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return MZ_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" int mz_expand_backup_net_static(mz_ctx *c, int32_t select_next, const double *noise, int32_t noise_max_calls,
                                           void *stream)
{
    int rc = check_ctx(c, "mz_expand_backup_net_static");
    if (rc) return rc;
    if (!c->selected) return fail(MZ_ERR_STATE, "mz_expand_backup_net_static: no pending selection (call mz_select first)");
    if (!c->mnet.steps) return fail(MZ_ERR_STATE, "mz_expand_backup_net_static: no network uploaded (call mz_set_network_mfma first)");
    const size_t rows = ((size_t)c->B + 31) / 32 * 32;
    const size_t acts_bytes = align_up(rows * (size_t)c->mnet.stride * 4, 256);
    const size_t logits_bytes = align_up((size_t)c->B * c->cfg.n_actions * 4, 256);
    const size_t scalar_bytes = align_up((size_t)c->B * 8, 256);
    const size_t need = acts_bytes + logits_bytes + 2 * scalar_bytes;
    if (need > MZ_STATIC_SCRATCH_BYTES)
        return fail(MZ_ERR_ARG, "mz_expand_backup_net_static: " + std::to_string(c->B) + " trees x " + std::to_string(c->mnet.stride) +
                                    " activation floats need " + std::to_string(need) + " bytes of staging, the fixed scratch holds " +
                                    std::to_string((size_t)MZ_STATIC_SCRATCH_BYTES) + "; use mz_expand_backup_net");
    void *base = nullptr;
    if ((rc = scratch_for_device(c->cfg.device, &base))) return rc;
    unsigned char *p = reinterpret_cast<unsigned char *>(base);
    const LeafStaging stg{reinterpret_cast<float *>(p), reinterpret_cast<float *>(p + acts_bytes),
                          reinterpret_cast<double *>(p + acts_bytes + logits_bytes),
                          reinterpret_cast<double *>(p + acts_bytes + logits_bytes + scalar_bytes)};
    // The array keeps whatever the last call on this GPU left (another batch size, another network's layout): layers read
    // their inputs in blocks of 16 floats, past the end of a tensor into padding that no layer writes, and that padding
    // must be finite (its weights are zero).  Start every call from zeroed activations.
    HIPCHK(hipMemsetAsync(stg.acts, 0, acts_bytes, (hipStream_t)stream));
    return expand_backup_net_impl(c, select_next, noise, noise_max_calls, stream, stg, true);
}

static int act_env_step(mz_ctx *c, const mz_env_state *env, const mz_move_log *log, int32_t row, double temperature,
                        const double *u, int32_t *out_action, double *out_probs, double *out_root_value,
                        float *out_obs_before, float *out_reward, uint8_t *out_done, int32_t *out_to_play, void *stream,
                        const char *fn)
{
    int rc = check_ctx(c, fn);
    if (rc) return rc;
    const std::string f(fn);
    if (c->B < 1) return fail(MZ_ERR_STATE, f + ": no tree");
    if (!env) return fail(MZ_ERR_ARG, f + ": null argument");
    if (!log && (!out_action || !out_probs || !out_root_value || !out_reward || !out_done || !out_to_play))
        return fail(MZ_ERR_ARG, f + ": null argument");
    if (env->kind != MZ_ENV_TICTACTOE && env->kind != MZ_ENV_CARTPOLE) return fail(MZ_ERR_ARG, f + ": unknown environment kind");
    const int want_A = env->kind == MZ_ENV_TICTACTOE ? 9 : 2, want_obs = env->kind == MZ_ENV_TICTACTOE ? 27 : 4;
    if (c->cfg.n_actions != want_A || env->obs_size != want_obs)
        return fail(MZ_ERR_ARG, f + ": action count / observation size do not match the environment");
    if (!env->state || !env->to_play || !env->legal || !env->cur_obs || !env->search_obs || !env->moves ||
        (env->kind == MZ_ENV_TICTACTOE && !env->mover) || (env->kind == MZ_ENV_CARTPOLE && !env->reset_state))
        return fail(MZ_ERR_ARG, f + ": incomplete environment state");
    LogDev lg{};
    if (log) {
        if (!log->observations || !log->actions || !log->rewards || !log->to_plays || !log->root_values || !log->probs || !log->done)
            return fail(MZ_ERR_ARG, f + ": incomplete move log");
        if (log->n_envs != c->B || log->obs_size != env->obs_size || log->n_actions != c->cfg.n_actions)
            return fail(MZ_ERR_ARG, f + ": move log shaped for another batch / environment");
        if (row < 0 || row >= log->capacity) return fail(MZ_ERR_ARG, f + ": the move log is full (row outside 0..capacity-1)");
        lg = LogDev{log->observations, log->actions, log->to_plays, log->rewards, log->root_values, log->probs, log->done,
                    (long long)row * log->n_envs};
    }
    const int n_stack = env->n_stacked_observations > 0 ? env->n_stacked_observations : 1;
    if (n_stack > 1024 || (env->stack_action != 0 && env->stack_action != 1))
        return fail(MZ_ERR_ARG, f + ": n_stacked_observations / stack_action out of range");
    if ((n_stack > 1 || env->stack_action) && !env->stale_observation)
        return fail(MZ_ERR_ARG, f + ": stacked observations follow the history saved so far (GameHistory.stack_n_observations), "
                                    "which is the stale rule: set stale_observation");
    EnvDev e;
    e.kind = env->kind; e.max_moves = env->max_moves; e.stale = env->stale_observation; e.obs_size = env->obs_size;
    e.n_stack = n_stack; e.planes = env->stack_action ? 2 : 1;
    e.state = env->state; e.mover = env->mover; e.to_play = env->to_play; e.legal = env->legal;
    e.cur_obs = env->cur_obs; e.search_obs = env->search_obs; e.moves = env->moves; e.reset_state = env->reset_state;
    MZ_LAUNCH(c, k_act_env, true, c->B, 0, stream, c->d, e, lg, temperature, u, out_action, out_probs, out_root_value,
              out_obs_before, out_reward, out_done, out_to_play);
    return launch_status(fn);
}

extern "C" int mz_act_env_step(mz_ctx *c, const mz_env_state *env, double temperature, const double *u,
                               int32_t *out_action, double *out_probs, double *out_root_value,
                               float *out_obs_before, float *out_reward, uint8_t *out_done,
                               int32_t *out_to_play, void *stream)
{
    return act_env_step(c, env, nullptr, 0, temperature, u, out_action, out_probs, out_root_value, out_obs_before, out_reward,
                        out_done, out_to_play, stream, "mz_act_env_step");
}

extern "C" int mz_act_env_step_log(mz_ctx *c, const mz_env_state *env, const mz_move_log *log, int32_t row,
                                   double temperature, const double *u, int32_t *out_action, double *out_probs,
                                   double *out_root_value, float *out_obs_before, float *out_reward, uint8_t *out_done,
                                   int32_t *out_to_play, void *stream)
{
    if (!log) return fail(MZ_ERR_ARG, "mz_act_env_step_log: null move log");
    return act_env_step(c, env, log, row, temperature, u, out_action, out_probs, out_root_value, out_obs_before, out_reward,
                        out_done, out_to_play, stream, "mz_act_env_step_log");
}

static int check_log(const mz_move_log *log, int32_t n_rows, const char *fn)
{
    if (!log || !log->done || log->n_envs < 1 || n_rows < 0 || n_rows > log->capacity)
        return fail(MZ_ERR_ARG, std::string(fn) + ": bad move log or row count");
    return MZ_OK;
}

extern "C" int mz_log_count_games(const mz_move_log *log, int32_t n_rows, int32_t *out_count, void *stream)
{
    (void)hipGetLastError();
    int rc = check_log(log, n_rows, "mz_log_count_games");
    if (rc) return rc;
    if (!out_count) return fail(MZ_ERR_ARG, "mz_log_count_games: null output");
    hipLaunchKernelGGL(k_log_games<false>, dim3((log->n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, log->done,
                       log->open_at_start, (int)log->n_envs, (int)n_rows, out_count, (const long long *)nullptr, (long long *)nullptr,
                       (int32_t *)nullptr);
    return launch_status("mz_log_count_games");
}

extern "C" int mz_log_list_games(const mz_move_log *log, int32_t n_rows, const int64_t *first_game, int64_t *out_base,
                                 int32_t *out_len, void *stream)
{
    (void)hipGetLastError();
    int rc = check_log(log, n_rows, "mz_log_list_games");
    if (rc) return rc;
    if (!first_game || !out_base || !out_len) return fail(MZ_ERR_ARG, "mz_log_list_games: null argument");
    hipLaunchKernelGGL(k_log_games<true>, dim3((log->n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, log->done,
                       log->open_at_start, (int)log->n_envs, (int)n_rows, (int32_t *)nullptr, reinterpret_cast<const long long *>(first_game),
                       reinterpret_cast<long long *>(out_base), out_len);
    return launch_status("mz_log_list_games");
}
