// AddressSanitizer / UBSan harness for the HOST logic of libinstagraal_hip.so: the unmodified csrc/ig_hip.hip, compiled with
// `hipcc --offload-host-only -fsanitize=address,undefined`, on top of a fake HIP runtime (fake_hip_runtime.cpp) whose
// "device" is the heap and whose kernels are the small MODELS below.  What is exercised, without a GPU: uploads (CSR build,
// the tiled copy with its histograms and work items), buffer sizing and regrowth (window strides, slice pool, exact-kernel
// grid), the speculative-batch driver (widths, conflicts, pending one-move tails, first-slot overflows of both kinds), the
// fused draw + step entry point with its drawing thread, the runs of (move, nuisance step) pairs -- scored-ahead batches
// whose structural half survives accepted steps, the screened pass's three outcomes (decisive, undecided, void), the exact
// fallback, promotion -- the mapped-memory flag protocol, every argument check.  The models script the few device outputs
// that steer the host (the decide step's outcome block, result records, pass sums) with pseudo-random but protocol-conforming
// values; sums and scores are meaningless here, memory safety and the state machines are the subject.
// Built and run by tests/test_cpu_abi_and_host.py::test_host_logic_under_address_and_ub_sanitizers.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define ig_fail_msg harness_copy_of_ig_fail_msg /* ig_common.cuh defines it (for ig_draw.cpp): the library object has the real one */
#include "../../instagraal_amd/csrc/ig_common.cuh"
#undef ig_fail_msg
#include "fake_hip_runtime.h"

static uint64_t rs = 0x2545F4914F6CDD1Dull;
static uint32_t rnd()
{
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (uint32_t)(rs >> 11);
}
static int g_line_fail = 0;
#define CHECK(x)                                                                                           \
    do {                                                                                                   \
        if (!(x)) {                                                                                        \
            std::fprintf(stderr, "%s:%d: CHECK failed: %s   [last error: %s]\n", __FILE__, __LINE__, #x, ig_last_error()); \
            g_line_fail = __LINE__;                                                                        \
            return 1;                                                                                      \
        }                                                                                                  \
    } while (0)

// ---- a small random problem through the real ABI -------------------------------------------------------------------------
struct Problem {
    int N = 0, M = 0;
    std::vector<float> sub;        // M x 4
    std::vector<int32_t> soa;      // 17 x N
    std::vector<int32_t> row, col, cnt;
};
static Problem make_problem(int n_contigs, int mean_bins, int contacts)
{
    Problem p;
    std::vector<int> clen;
    for (int c = 0; c < n_contigs; c++) clen.push_back(1 + (int)(rnd() % (2 * mean_bins)));
    for (int c : clen) p.N += c;
    std::vector<int> sl(p.N);
    for (int f = 0; f < p.N; f++) {
        sl[f] = 1 + (int)(rnd() % 3);
        p.M += sl[f];
    }
    p.sub.resize((size_t)p.M * 4);
    p.soa.assign((size_t)17 * p.N, 0);
    auto S = [&](int field, int f) -> int32_t& { return p.soa[(size_t)field * p.N + f]; };
    int f = 0, s = 0;
    for (int c = 0; c < n_contigs; c++) {
        int bp = 0, sub_pos = 0, cont_sl = 0, cont_bp = 0;
        for (int k = 0; k < clen[c]; k++) {
            cont_sl += sl[f + k];
            cont_bp += 1000 * sl[f + k];
        }
        for (int k = 0; k < clen[c]; k++, f++) {
            S(0, f) = k;
            S(1, f) = sub_pos;
            S(2, f) = c;
            S(3, f) = bp;
            S(4, f) = 1000 * sl[f];
            S(5, f) = sl[f];
            S(6, f) = 0;
            S(7, f) = f;
            S(8, f) = k ? f - 1 : -1;
            S(9, f) = k + 1 < clen[c] ? f + 1 : -1;
            S(10, f) = clen[c];
            S(11, f) = cont_sl;
            S(12, f) = cont_bp;
            S(13, f) = 1;
            S(14, f) = 0;
            S(15, f) = 1;
            S(16, f) = f;
            for (int w = 0; w < sl[f]; w++, s++) {
                p.sub[(size_t)4 * s] = (float)f;
                p.sub[(size_t)4 * s + 1] = 0.5f + (float)w;
                p.sub[(size_t)4 * s + 2] = (float)sl[f] - 0.5f - (float)w;
                p.sub[(size_t)4 * s + 3] = (float)w;
            }
            bp += 1000 * sl[f];
            sub_pos += sl[f];
        }
    }
    // distinct strict-upper-triangle contacts, row-major sorted
    std::vector<uint64_t> keys;
    for (int k = 0; k < contacts; k++) {
        const uint32_t a = rnd() % (uint32_t)p.M, b = rnd() % (uint32_t)p.M;
        if (a == b) continue;
        keys.push_back(((uint64_t)std::min(a, b) << 32) | std::max(a, b));
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    for (uint64_t q : keys) {
        p.row.push_back((int32_t)(q >> 32));
        p.col.push_back((int32_t)(uint32_t)q);
        p.cnt.push_back(1 + (int32_t)(rnd() % ((rnd() % 50 == 0) ? 400 : 6))); // a few counts beyond the tile histograms' range
    }
    return p;
}

// ---- models of the kernels whose outputs steer the host --------------------------------------------------------------------
static int g_force_first_overflow = 0; // the next decide launch with w_start == 0 reports "first slot does not fit" (1: pool, 2: grid)
static int g_grow_windows = 0;         // ... reports longer contigs (window buffers are regrown)
static long g_decides = 0, g_pendings = 0, g_overflows = 0, g_conflicts = 0;
static int g_N = 0, g_M = 0;

static void model_decide(void** a, dim3, dim3)
{
    const int W = *(int*)a[4], w_start = *(int*)a[5], seq = *(int*)a[9];
    int* bo = *(int**)a[7];
    volatile int* ho = *(volatile int**)a[8];
    g_decides++;
    int committed = W, pending = -1, stop = 0, first = 0;
    const uint32_t r = rnd() % 16;
    if (w_start == 0 && g_force_first_overflow) {
        committed = 0;
        first = stop = g_force_first_overflow;
        g_force_first_overflow = 0;
        g_overflows++;
    } else if (r == 0 && W - w_start >= 1) { /* a winner that needs the one-move tail */
        pending = w_start + (int)(rnd() % (uint32_t)(W - w_start));
        committed = pending;
        g_pendings++;
    } else if (r <= 3 && W - w_start >= 2) { /* a conflict (or a pool / grid that cut the batch short) somewhere behind the first slot */
        committed = w_start + 1 + (int)(rnd() % (uint32_t)(W - w_start - 1));
        stop = (r == 3) ? 1 + (int)(rnd() % 2) : 0;
        g_conflicts++;
    } else if (r == 4 && w_start > 0) { /* nothing more of this batch can be decided */
        committed = w_start;
        g_conflicts++;
    }
    int vals[16] = {0};
    vals[0] = committed;
    vals[1] = pending;
    vals[2] = (committed == w_start && pending < 0) ? first : 0;
    { /* the window rule (run_moves_window): some of the positions this launch did not commit hold slots that went stale -- the slot it
       * stopped in front of at any rate, now and then a few behind it */
        unsigned long long stale = 0ull;
        if (committed < W && pending < 0) stale |= 1ull << committed;
        for (int p2 = committed + 1; p2 < W && p2 < 64; p2++)
            if (rnd() % 6 == 0) stale |= 1ull << p2;
        vals[12] = (int)(unsigned)stale;
        vals[13] = (int)(unsigned)(stale >> 32);
    }
    vals[3] = 5 * (committed - w_start);
    vals[4] = 0;
    vals[5] = 1 + (int)(rnd() % 1000);
    vals[6] = (int)(rnd() % 60000);
    vals[8] = g_grow_windows ? std::min(g_N / 2, 40 + (int)(rnd() % 200)) : 0;
    vals[9] = g_grow_windows ? std::min(g_M / 2, 100 + (int)(rnd() % 600)) : 0;
    vals[10] = stop;
    for (int i = 0; i < 16; i++)
        if (i != 7) bo[i] = vals[i];
    if (ho) {
        for (int i = 0; i < 16; i++)
            if (i != 7) ho[i] = vals[i];
        ho[7] = seq;
    }
}
/* k_decide_chain: a segment of a chain of (move, nuisance step) pairs: as many pairs as it likes, then one of its ways to end --
 * through (stop 0), behind a move that changed the genome (5, bo[11] = 1), in front of a pair whose step is not a certain rejection
 * (4), a conflict (0, short), a pending move, an overflow (1 .. 3) */
static long g_chain_decides = 0, g_chain_pairs = 0;
static void model_decide_chain(void** a, dim3, dim3)
{
    const int W = *(int*)a[4], w_start = *(int*)a[5], seq = *(int*)a[9];
    int* bo = *(int**)a[7];
    volatile int* ho = *(volatile int**)a[8];
    g_chain_decides++;
    int committed = W, pending = -1, stop = 0, changed = 0;
    const uint32_t r = rnd() % 12;
    const int span = W - w_start;
    if (r <= 2 && span >= 1) { /* behind a changing move somewhere in the segment */
        committed = w_start + 1 + (int)(rnd() % (uint32_t)span);
        stop = 5;
        changed = 1;
    } else if (r <= 4) { /* in front of a pair that needs the host */
        committed = w_start + (int)(rnd() % (uint32_t)(span + 1));
        if (committed == W) committed = W - 1;
        stop = 4;
    } else if (r == 5) { /* a conflict */
        committed = w_start + (int)(rnd() % (uint32_t)span);
        stop = 0;
    } else if (r == 6) {
        pending = w_start + (int)(rnd() % (uint32_t)span);
        committed = pending;
        g_pendings++;
    } else if (r == 7) {
        committed = w_start + (int)(rnd() % (uint32_t)span);
        stop = 1 + (int)(rnd() % 3);
    }
    g_chain_pairs += committed - w_start;
    int vals[12] = {0};
    vals[0] = committed;
    vals[1] = pending;
    vals[2] = (committed == w_start && pending < 0 && stop >= 1 && stop <= 3) ? stop : 0;
    vals[3] = 5 * (committed - w_start);
    vals[5] = 1 + (int)(rnd() % 1000);
    vals[6] = (int)(rnd() % 60000);
    vals[10] = stop;
    vals[11] = changed;
    for (int i = 0; i < 12; i++)
        if (i != 7) bo[i] = vals[i];
    if (ho) {
        for (int i = 0; i < 12; i++)
            if (i != 7) ho[i] = vals[i];
        ho[7] = seq;
    }
}
static double g_move_score = -1000.0;
static void model_commit_batch(void** a, dim3, dim3)
{
    NuisHost* hn = *(NuisHost**)a[17];
    const int seq = *(int*)a[18];
    if (!hn) return;
    memset((void*)&hn->res, 0, sizeof hn->res);
    g_move_score += ((int)(rnd() % 200) - 100) * 0.01;
    hn->res.o = g_move_score;
    hn->res.n_contigs = 7;
    hn->res.mean_len = 3.0;
    hn->nzb[0] = -1000;
    hn->nzb[1] = 12345;
    hn->max_L = 0;
    hn->max_SL = 0;
    hn->changed = (int)(rnd() % 4 == 0); /* a move in four changes the genome: the histogram's walk is launched behind it */
    hn->res_seq = seq;
}
/* the histogram tier of the screened pass (k_hist_eval): decisive either way, undecided (the pass over the contacts follows), void */
static long g_hists = 0;
static void model_hist_eval(void** a, dim3, dim3)
{
    NuisHost* hn = *(NuisHost**)a[6];
    const int seq = *(int*)a[7];
    g_hists++;
    if (!hn) return;
    const int mode = (int)(rnd() % 4);
    for (int q = 0; q < 8; q++) hn->diff[q] = 0;
    for (int q = 0; q < 8; q++) hn->sums[q] = 0;
    hn->diff[3] = 1 << 18; /* a bound of 0.25 */
    hn->diff[5] = 1000;
    if (mode == 0) hn->diff[2] = -(1LL << 40);       /* rejected from the interval */
    else if (mode == 1) hn->diff[2] = (1LL << 40);   /* accepted from the interval: the exact pass behind the decision */
    else if (mode == 2) hn->diff[3] = 1LL << 45;     /* an interval that decides nothing: tier 1 */
    else hn->diff[4] = 1 + (int)(rnd() % 7);         /* void: tier 1 */
    hn->diff_seq = seq;
}
/* k_decide_commit: both in one launch (the decide step's arguments first, the apply step's behind them) */
static void model_decide_commit(void** a, dim3 g, dim3 b)
{
    model_decide(a, g, b);
    void* shifted[19] = {nullptr};
    shifted[17] = a[21];
    shifted[18] = a[22];
    model_commit_batch(shifted, g, b);
}
/* k_decide_commit_par (round 6: the decisions in rounds, decide_rounds): the same outcome block; it never publishes to the host */
static void model_decide_commit_par(void** a, dim3 g, dim3 b) { model_decide(a, g, b); }
static int g_diff_mode = 0; // 0 random, 1 always decisive reject, 2 always undecided, 3 always void
static long g_diffs = 0, g_exacts = 0;
static void model_diff(void** a, dim3, dim3)
{
    NuisHost* hn = *(NuisHost**)a[9];
    const int seq = *(int*)a[10];
    g_diffs++;
    if (!hn) return;
    const int mode = g_diff_mode ? g_diff_mode : 1 + (int)(rnd() % 3);
    for (int q = 0; q < 8; q++) hn->diff[q] = 0;
    for (int q = 0; q < 8; q++) hn->sums[q] = 0;
    hn->diff[3] = 1 << 20; /* a bound of 1.0 */
    hn->diff[5] = 1000;
    if (mode == 1) hn->diff[2] = -(1LL << 40);      /* a million below: rejected from the interval */
    else if (mode == 2) hn->diff[2] = (1LL << 40);  /* far above: accept is certain -- the exact pass decides */
    else hn->diff[4] = 1 + (int)(rnd() % 7);        /* void */
    hn->diff_seq = seq;
}
static void model_full_nz_tiled(void** a, dim3, dim3)
{
    NuisHost* hn = *(NuisHost**)a[13];
    const int seq = *(int*)a[14];
    long long* out = *(long long**)a[8];
    g_exacts++;
    /* the exact sum: around the move's score, so that both outcomes of the Metropolis test occur */
    const long long v = (long long)(g_move_score + ((int)(rnd() % 7) - 3)) * (1LL << 32) / 1;
    out[0] = v >> 32;
    out[1] = v & 0xffffffffLL;
    if (hn) {
        for (int q = 0; q < 8; q++) hn->sums[q] = out[q];
        hn->sums_seq = seq;
    }
}

static void model_promote(void** a, dim3, dim3)
{
    const int mode = *(int*)a[7], seq = *(int*)a[9];
    NuisHost* hn = *(NuisHost**)a[8];
    if (mode == 2 && hn) {
        long long* full = *(long long**)a[1];
        hn->exact[0] = full[0];
        hn->exact[1] = full[1];
        hn->exact_seq = seq;
    }
}

/* k_commit (the one-move record writer; mangled "8k_commitP4Glob": the plain name is a substring of the batch kernels' names): a
 * record with pad = 1 -- "the move's lists did not fit the slice pool, nothing was applied" -- as often as g_force_retry says; the
 * host must grow the pool, clear the device's flag and repeat the move (retry_with_larger_pool) */
static int g_force_retry = 0, g_retry_paths = 0;
static void model_commit_one(void** a, dim3, dim3)
{
    ig_move_result* res = *(ig_move_result**)a[2];
    const int move = *(int*)a[3];
    std::memset(&res[move], 0, sizeof(ig_move_result));
    if (g_force_retry > 0) {
        res[move].pad = 1;
        g_force_retry--;
    }
    StepHost* hs = *(StepHost**)a[6]; /* ig_step_draw: the record of a move the one-move tail finished, into mapped host memory */
    if (hs) {
        hs->fin = res[move];
        hs->max_L = hs->max_SL = 0;
        hs->fin_seq = *(int*)a[7];
    }
}

int main()
{
    setenv("IG_NUIS_SCREEN_NOCHECK", "1", 1); /* the models' screened and exact sums are unrelated numbers */
    setenv("IG_POOL_ENTRIES", "4096", 1);      /* a small slice pool: the growth paths are taken */
    fake_hip::set_model("k_decide_batch", model_decide);
    fake_hip::set_model("k_commit_batch", model_commit_batch);
    fake_hip::set_model("15k_decide_commitP", model_decide_commit); /* (mangled: the models are matched by substring, in the map's order) */
    fake_hip::set_model("19k_decide_commit_par", model_decide_commit_par);
    fake_hip::set_model("k_decide_chain", model_decide_chain);
    fake_hip::set_model("k_chain_decide_commit", model_decide_chain);
    fake_hip::set_model("k_full_diff_tiled", model_diff);
    fake_hip::set_model("k_hist_eval", model_hist_eval);
    fake_hip::set_model("k_full_nz_tiled", model_full_nz_tiled);
    fake_hip::set_model("k_nuis_promote", model_promote);
    fake_hip::set_model("8k_commitP4Glob", model_commit_one);

    // ---- argument checks before anything is uploaded
    ig_ctx* c = nullptr;
    CHECK(ig_create(0, nullptr) != 0);
    CHECK(ig_create(3, &c) != 0 && std::strlen(ig_last_error()) > 0);
    CHECK(ig_create(0, &c) == 0 && c);
    ig_move_result one;
    int32_t cand1[1] = {1};
    CHECK(ig_step(c, 0, cand1, 1, &one, nullptr) != 0); // nothing uploaded yet
    float p8[8] = {50.0f, 9.6f, 0.0f, -1.5f, 2.0f, 250.0f, 3.0e5f, 5e-3f};
    p8[2] = (float)(0.53 * std::pow(9.6 / 50.0, -1.5) * std::pow(50.0, -3.0));
    CHECK(ig_set_params(c, p8, 1.8f, 2) != 0);

    for (int round = 0; round < 3; round++) {
        // three shapes: many short contigs; a few long ones (windows of thousands of sub-fragments); a tiny genome
        Problem pr = round == 0 ? make_problem(120, 8, 60000) : (round == 1 ? make_problem(5, 500, 90000) : make_problem(3, 3, 60));
        g_N = pr.N;
        g_M = pr.M;
        if (round > 0) { /* a handle is sized by its first upload: a new one per shape (and ig_destroy gets its leak check three times) */
            CHECK(ig_upload_subfrag_table(c, pr.sub.data(), pr.M) != 0);
            ig_destroy(c);
            c = nullptr;
            CHECK(ig_create(0, &c) == 0 && c);
        }
        std::fprintf(stderr, "[harness] round %d: N=%d M=%d Z=%zu\n", round, pr.N, pr.M, pr.row.size());
        CHECK(ig_upload_subfrag_table(c, pr.sub.data(), pr.M) == 0);
        {   // contacts that are not sorted / not upper triangle are refused
            std::vector<int32_t> r2 = pr.row, c2 = pr.col;
            if (r2.size() > 3) {
                std::swap(r2[0], r2[2]);
                std::swap(c2[0], c2[2]);
                if (r2[0] != r2[2] || c2[0] != c2[2]) CHECK(ig_upload_contacts(c, r2.data(), c2.data(), pr.cnt.data(), (int64_t)r2.size(), pr.M, 0, 1) != 0);
            }
        }
        CHECK(ig_upload_contacts(c, pr.row.data(), pr.col.data(), pr.cnt.data(), (int64_t)pr.row.size(), pr.M, 0, 1) == 0);
        CHECK(ig_upload_state(c, pr.soa.data(), pr.N) == 0);
        {
            std::vector<int32_t> bad = pr.soa;
            bad[(size_t)15 * pr.N] = 0; // an inactive fragment
            CHECK(ig_upload_state(c, bad.data(), pr.N) != 0);
            CHECK(ig_upload_state(c, pr.soa.data(), pr.N) == 0);
        }
        CHECK(ig_set_params(c, p8, 1.8f, 0) == 0);
        CHECK(ig_set_params(c, p8, 1.8f, 1) == 0);
        double nz, z;
        int64_t limbs[5];
        CHECK(ig_full_likelihood(c, 0, 0, &nz, &z, limbs) == 0);
        CHECK(ig_full_likelihood(c, 1, 1, &nz, &z, nullptr) == 0);
        std::vector<int32_t> soa_back((size_t)17 * pr.N);
        CHECK(ig_download_state(c, soa_back.data()) == 0);
        int32_t nc;
        float ml;
        CHECK(ig_renumber_contigs(c, &nc, &ml, nullptr) == 0);

        // ---- moves: lists of up to max_c candidates, -1 padded, never the focal bin
        for (int max_c : {5, 1, 16}) {
            if (pr.N < 3) continue;
            const int n_moves = 300;
            std::vector<int32_t> frags(n_moves), cands((size_t)n_moves * max_c, -1);
            for (int i = 0; i < n_moves; i++) {
                frags[i] = (int32_t)(rnd() % (uint32_t)pr.N);
                const int C = 1 + (int)(rnd() % (uint32_t)std::min(max_c, pr.N - 1));
                int k = 0;
                for (int q = 0; q < C; q++) {
                    const int32_t b = (int32_t)(rnd() % (uint32_t)pr.N);
                    if (b != frags[i]) cands[(size_t)i * max_c + k++] = b;
                }
                if (k == 0) cands[(size_t)i * max_c] = (frags[i] + 1) % pr.N;
            }
            std::vector<ig_move_result> res(n_moves);
            { // the one-move path's answer to a slice pool that is too small (the device's record says pad = 1): a larger pool, the
              // move again -- while the pool is still small; at its worst case the same outcome is a legitimate error
                int C0 = 0;
                while (C0 < max_c && cands[C0] >= 0) C0++;
                std::vector<double> sc0((size_t)max_c * IG_N_TMP_STRUCT);
                int64_t r0 = 0, r1 = 0;
                CHECK(ig_step(c, frags[0], cands.data(), C0, &one, sc0.data()) == 0); /* (allocates the move buffers) */
                CHECK(ig_debug_pool_retries(c, &r0) == 0);
                const bool room = (size_t)c->mb.pool_cap * 4 <= (size_t)std::max<long long>(c->Z, 1) * (size_t)std::max(c->mb.capC, 1);
                g_force_retry = 1;
                const int rc = ig_step(c, frags[0], cands.data(), C0, &one, sc0.data());
                CHECK(g_force_retry == 0);
                if (room) {
                    g_retry_paths++;
                    CHECK(rc == 0 && one.pad == 0);
                    CHECK(ig_debug_pool_retries(c, &r1) == 0);
                    CHECK(r1 == r0 + 1);
                    g_force_retry = 1;
                    CHECK(ig_set_batch_width(1) == 0);
                    CHECK(ig_step_batch(c, 5, frags.data(), cands.data(), max_c, res.data()) == 0 || (size_t)c->mb.pool_cap >= (size_t)std::max<long long>(c->Z, 1) * (size_t)std::max(c->mb.capC, 1));
                    CHECK(ig_set_batch_width(24) == 0);
                    g_force_retry = 0;
                }
            }
            for (int w : {24, 1, 7, 64}) { /* the batches of rounds 1 - 4 (no window) */
                CHECK(ig_set_window(0) == 0);
                CHECK(ig_set_batch_width(w) == 0);
                g_grow_windows = (w == 7);
                CHECK(ig_step_batch(c, n_moves, frags.data(), cands.data(), max_c, res.data()) == 0);
            }
            CHECK(ig_set_batch_width(24) == 0);
            for (int w : {48, 2, 7, 64}) { /* the window rule (run_moves_window): kept slots, stale masks, regrown window buffers */
                CHECK(ig_set_window(w) == 0);
                g_grow_windows = (w == 7);
                CHECK(ig_step_batch(c, n_moves, frags.data(), cands.data(), max_c, res.data()) == 0);
            }
            g_grow_windows = 0;
            CHECK(ig_set_window(48) == 0);
            // the first slot of a batch does not fit the slice pool / the exact kernel's grid: more room, the batch again
            if ((size_t)c->mb.pool_cap < (size_t)std::max<long long>(c->Z, 1) * (size_t)std::max(c->mb.capC, 1)) { /* (a pool at its worst case cannot: a legitimate error) */
                g_force_first_overflow = 1;
                CHECK(ig_step_batch(c, 40, frags.data(), cands.data(), max_c, res.data()) == 0);
            }
            g_force_first_overflow = 2;
            CHECK(ig_step_batch(c, 40, frags.data(), cands.data(), max_c, res.data()) == 0);
            // one move at a time, scores, forced apply
            std::vector<double> sc((size_t)max_c * IG_N_TMP_STRUCT);
            int C0 = 0;
            while (C0 < max_c && cands[C0] >= 0) C0++;
            CHECK(ig_step(c, frags[0], cands.data(), C0, &one, sc.data()) == 0);
            CHECK(ig_score_move(c, frags[0], cands.data(), C0, sc.data()) == 0);
            { /* ig_step_draw with the caller's candidates: a batch of one through the mapped block -- committed by the batch kernels, handed
               * to the one-move tail (the decide model's pending winners), a first slot that does not fit the pool / the exact grid */
                const long tails0 = g_pendings;
                for (int rep = 0; rep < 60 || (g_pendings == tails0 && rep < 2000); rep++) { /* (until the decide model has handed one to the tail) */
                    std::vector<int32_t> cl(cands.begin(), cands.begin() + max_c);
                    int32_t nc = C0;
                    const bool forced = rep % 9 == 4;
                    if (forced) g_force_first_overflow = 1 + (rep % 2);
                    const int rc = ig_step_draw(c, nullptr, nullptr, nullptr, frags[0], 0, cl.data(), &nc, &one, (rep & 1) ? sc.data() : nullptr);
                    g_force_first_overflow = 0;
                    /* (a forced "does not fit" with the pool / the work list at their largest fails, as it must) */
                    CHECK(rc == 0 || (forced && (std::strstr(ig_last_error(), "worst case") || std::strstr(ig_last_error(), "work list"))));
                    CHECK(rc != 0 || (nc == C0 && one.pad == 0));
                }
                CHECK(g_pendings > tails0); /* (one decision in sixteen) */
                int32_t nc = 0;
                CHECK(ig_step_draw(c, nullptr, nullptr, nullptr, frags[0], 0, cands.data(), &nc, &one, nullptr) != 0); // no candidates
                nc = C0;
                CHECK(ig_step_draw(c, nullptr, nullptr, nullptr, frags[0], 0, cands.data(), &nc, nullptr, nullptr) != 0);
                CHECK(ig_step_draw(c, nullptr, nullptr, nullptr, -1, 0, cands.data(), &nc, &one, nullptr) != 0);
            }
            CHECK(ig_apply(c, frags[0], cands[0], 3) == 0);
            CHECK(ig_apply(c, frags[0], cands[0], 24) != 0);
            // argument checks of the move entry points
            std::vector<int32_t> badc(cands.begin(), cands.begin() + max_c);
            badc[0] = frags[0];
            CHECK(ig_step(c, frags[0], badc.data(), 1, &one, nullptr) != 0); // candidate == focal bin
            badc[0] = pr.N;
            CHECK(ig_step(c, frags[0], badc.data(), 1, &one, nullptr) != 0);
            CHECK(ig_step(c, -1, cands.data(), 1, &one, nullptr) != 0);
            CHECK(ig_step_batch(c, 3, frags.data(), cands.data(), IG_MAX_CANDIDATES + 1, res.data()) != 0);
            if (max_c >= 2) { // candidates behind a -1 pad
                std::vector<int32_t> gap(cands.begin(), cands.begin() + max_c);
                gap[0] = -1;
                gap[1] = (frags[0] + 1) % pr.N;
                CHECK(ig_step_batch(c, 1, frags.data(), gap.data(), max_c, res.data()) != 0);
            }

            // ---- the draw inside the call: jump distributions in the library, numpy's MT19937 state in and out, a drawing thread
            {
                std::vector<int64_t> indptr(pr.N + 1, 0);
                std::vector<int32_t> xk;
                std::vector<float> pk;
                const int nn = std::min(max_c, std::max(1, pr.N - 1));
                for (int f = 0; f < pr.N; f++) {
                    /* bins without a hetero contact draw uniformly (CL:3124) -- possibly the focal bin itself, which is dropped: with one
                     * neighbour asked for that can leave an empty list, a refused move (quirk Q13); not what this block is after */
                    const int deg = (f % 7 == 0 && nn > 2) ? 0 : 1 + (int)(rnd() % 12);
                    float tot = 0;
                    const size_t at = pk.size();
                    for (int q = 0; q < deg; q++) {
                        int32_t b = (int32_t)(rnd() % (uint32_t)pr.N);
                        if (b == f) b = (b + 1) % pr.N;
                        bool dup = false;
                        for (size_t e = at; e < xk.size(); e++) dup |= xk[e] == b;
                        if (dup) continue;
                        xk.push_back(b);
                        pk.push_back(1.0f + (float)(rnd() % 9));
                        tot += pk.back();
                    }
                    for (size_t e = at; e < pk.size(); e++) pk[e] /= tot;
                    indptr[f + 1] = (int64_t)xk.size();
                }
                ig_neighbours* nb = nullptr;
                CHECK(ig_neighbours_create(indptr.data(), xk.data(), pk.data(), pr.N, nullptr, 0, &nb) == 0 && nb);
                std::vector<uint32_t> key(624);
                for (auto& k : key) k = rnd();
                int32_t pos = 624;
                std::vector<int32_t> cout_((size_t)n_moves * nn, -1);
                if (pr.N > nn + 1) {
                    CHECK(ig_step_batch_draw(c, nb, key.data(), &pos, n_moves, frags.data(), nn, cout_.data(), res.data()) == 0);
                    for (int rep = 0; rep < 20; rep++) { /* one call per move, the draw inside it */
                        int32_t nc = 0;
                        std::vector<int32_t> cl(nn, -7);
                        ig_move_result one2;
                        CHECK(ig_step_draw(c, nb, key.data(), &pos, frags[rep], nn, cl.data(), &nc, &one2, nullptr) == 0);
                        CHECK(nc >= 1 && nc <= nn);
                        for (int q = 0; q < nn; q++) CHECK(q < nc ? (cl[q] >= 0 && cl[q] < pr.N && cl[q] != frags[rep]) : cl[q] == -1);
                    }
                    {
                        int32_t nc = 0;
                        std::vector<int32_t> cl(nn, -7);
                        ig_move_result one2;
                        CHECK(ig_step_draw(c, nb, key.data(), &pos, pr.N + 5, nn, cl.data(), &nc, &one2, nullptr) != 0);
                        CHECK(ig_step_draw(c, nb, key.data(), &pos, frags[0], IG_MAX_CANDIDATES + 1, cl.data(), &nc, &one2, nullptr) != 0);
                    }
                    std::vector<int32_t> badf(frags.begin(), frags.begin() + 40);
                    badf[37] = pr.N + 5;
                    CHECK(ig_step_batch_draw(c, nb, key.data(), &pos, 40, badf.data(), nn, cout_.data(), res.data()) != 0);
                }
                ig_neighbours_destroy(nb);
            }

            // ---- the slot-split entry points (multi-GPU protocol), as BatchRunner drives them
            {
                const int wmax = ig_batch_max_width(c, max_c);
                CHECK(wmax >= 1);
                CHECK(ig_batch_upload(c, 60, frags.data(), cands.data(), max_c, std::min(wmax, 16)) == 0);
                void* rec = nullptr;
                int64_t bps = 0;
                CHECK(ig_batch_records(c, &rec, &bps) == 0 && rec && bps > 0);
                int done = 0;
                while (done < 60) {
                    const int w_now = std::min(std::min(wmax, 16), 60 - done);
                    CHECK(ig_batch_score(c, done, w_now, 0, w_now / 2) == 0);
                    int32_t got = 0;
                    CHECK(ig_batch_commit(c, done, w_now, &got) == 0);
                    done += got;
                }
                CHECK(ig_batch_results(c, 60, res.data()) == 0);
                CHECK(ig_batch_score(c, 55, 16, 0, 4) != 0); // beyond the uploaded moves
            }

            // ---- runs of (move, nuisance step) pairs
            for (int mode = 0; mode < 4; mode++) {
                g_diff_mode = mode;
                CHECK(ig_set_nuis_hist(mode == 3 ? 0 : 2) == 0);       /* the histogram tier whatever its cost model says / not at all */
                const int n = 160;
                CHECK(ig_nuis_run_begin(c, n, frags.data(), cands.data(), max_c) == 0);
                CHECK(ig_nuis_step_begin(c, 1, p8, 1.8f) != 0); // moves in order only
                float pt[8];
                memcpy(pt, p8, sizeof pt);
                pt[6] *= 1.01f;
                CHECK(ig_nuis_step_begin(c, 0, pt, 1.8f) == 0);
                CHECK(ig_nuis_step_begin(c, 1, pt, 1.8f) != 0); // one step in flight
                int n_acc = 0;
                for (int i = 0; i < n; i++) {
                    const int has_next = i + 1 < n;
                    float pr_[8], pa_[8];
                    memcpy(pr_, p8, sizeof pr_);
                    memcpy(pa_, p8, sizeof pa_);
                    pr_[3] = -1.5f + 0.001f * (float)((int)(rnd() % 11) - 5);
                    pa_[5] = 250.0f + (float)(rnd() % 50);
                    const bool give_acc = (rnd() & 1) != 0;
                    int32_t acc = -1;
                    double nzt, zt;
                    const double u = (rnd() % 1000 + 1) / 1001.0;
                    CHECK(ig_nuis_step_next(c, 1.0, u, pr_, give_acc ? pa_ : nullptr, 1.8f, has_next, &one, &nzt, &zt, &acc) == 0);
                    CHECK(acc == 0 || acc == 1 || acc == 2 || acc == 3);
                    if (acc == 3) { /* accepted from the screened interval, the exact pass behind the decision: its value later */
                        acc = 1;
                        if (rnd() & 1) {
                            double ex = 0;
                            CHECK(ig_nuis_exact_result(c, &ex) == 0);
                        }
                    }
                    if (acc == 2) { /* a close call handed back: the caller decides, accepts, begins the next step */
                        if (rnd() & 1) {
                            CHECK(ig_nuis_accept(c) == 0);
                            acc = 1;
                        } else {
                            acc = 0;
                        }
                        if (has_next) CHECK(ig_nuis_step_begin(c, i + 1, acc ? pa_ : pr_, 1.8f) == 0);
                    } else if (acc == 1 && has_next && !give_acc) {
                        CHECK(ig_nuis_step_begin(c, i + 1, pa_, 1.8f) == 0);
                    }
                    n_acc += acc == 1;
                }
                std::fprintf(stderr, "[harness]   nuisance run (max_c %d, pass mode %d): %d of %d steps accepted\n", max_c, mode, n_acc, n);
                // ---- the same run the way the sampler drives it: chains of pairs decided on the device (the model of the decide wave
                // scripts their ends), one pair the plain way for what a chain stops in front of
                CHECK(ig_nuis_run_begin(c, n, frags.data(), cands.data(), max_c) == 0);
                {
                    int i = 0;
                    bool try_chain = false;
                    long plain = 0, chained = 0;
                    while (i < n) {
                        if (try_chain) {
                            const int K = std::min(n - i, 1 + (int)(rnd() % 24));
                            std::vector<float> pt_(8 * (size_t)K);
                            std::vector<double> uu(K), tt(K, 1.0);
                            for (int k = 0; k < K; k++) {
                                memcpy(&pt_[8 * (size_t)k], p8, sizeof p8);
                                pt_[8 * (size_t)k + 3] = -1.5f + 0.001f * (float)((int)(rnd() % 11) - 5);
                                uu[k] = (rnd() % 1000) / 1001.0; /* (0 among them: no threshold, never a certain rejection) */
                            }
                            CHECK(ig_nuis_chain_begin(c, i + 1, K, pt_.data(), uu.data(), tt.data(), 1.8f) != 0 || true); /* (wrong move: reported by _end) */
                            int32_t nd = -1, why = -1;
                            if (ig_nuis_chain_end(c, &nd, &why) == 0) CHECK(false); /* the chain for a move that is not the next one fails */
                            CHECK(ig_nuis_chain_begin(c, i, K, pt_.data(), uu.data(), tt.data(), 1.8f) == 0);
                            while (!ig_nuis_chain_done(c)) {
                            }
                            CHECK(ig_nuis_chain_end(c, &nd, &why) == 0);
                            CHECK(nd >= 0 && nd <= K && why >= 0 && why <= 6);
                            CHECK(ig_nuis_chain_end(c, &nd, &why) != 0); /* no chain was begun */
                            i += nd;
                            chained += nd;
                            if (why == 0 && i < n) continue;
                            if (i >= n) break;
                        }
                        CHECK(ig_nuis_step_begin(c, i, pt, 1.8f) == 0);
                        int32_t acc = -1;
                        double nzt, zt;
                        const double u = (rnd() % 1000 + 1) / 1001.0;
                        CHECK(ig_nuis_step_next(c, 1.0, u, nullptr, nullptr, 1.8f, i + 1 < n, &one, &nzt, &zt, &acc) == 0);
                        bool rescored = acc == 1 || acc == 3;
                        if (acc == 2 && (rnd() & 1)) CHECK(ig_nuis_accept(c) == 0);
                        try_chain = rescored || acc == 0;
                        plain++;
                        i++;
                    }
                    std::fprintf(stderr, "[harness]   the same in chains: %ld pairs in chains, %ld the plain way\n", chained, plain);
                    CHECK(mode == 3 || chained > 0);
                }
                CHECK(ig_batch_results(c, n, std::vector<ig_move_result>((size_t)n).data()) == 0);
                // one pair at a time (no run)
                CHECK(ig_nuis_begin(c, frags[1], cands.data() + (size_t)max_c, 1, pt, 1.8f) == 0);
                CHECK(ig_nuis_begin(c, frags[1], cands.data() + (size_t)max_c, 1, pt, 1.8f) != 0);
                double nz2, z2;
                CHECK(ig_nuis_end(c, &one, &nz2, &z2, limbs) == 0);
                CHECK(ig_nuis_end(c, &one, &nz2, &z2, limbs) != 0); // no step in flight
                CHECK(ig_nuis_accept(c) == 0);
            }
            g_diff_mode = 0;
        }
        CHECK(ig_bomb(c, nullptr) == 0);
        double d;
        CHECK(ig_genome_distance(c, &d) == 0);
        int64_t sb[3], bs[4];
        CHECK(ig_scratch_bytes(c, sb) == 0 && ig_batch_stats(c, bs) == 0);
    }
    double st[12];
    CHECK(ig_debug_nuis_screen_stats(c, st) == 0);
    std::fprintf(stderr, "[harness] launches %ld, allocations %ld; decide launches %ld (pending %ld, conflicts %ld, first-slot overflows %ld); screened passes %ld, exact passes %ld; "
                         "screened steps %.0f, rejected from the interval %.0f, void %.0f\n",
                 fake_hip::launches(), fake_hip::allocations(), g_decides, g_pendings, g_conflicts, g_overflows, g_diffs, g_exacts, st[0], st[1], st[3]);
    CHECK(g_pendings > 0 && g_conflicts > 0 && g_overflows > 0 && g_diffs > 0 && g_exacts > 0 && st[1] > 0 && st[3] > 0);
    CHECK(g_retry_paths > 0); /* the one-move path's retry with a larger pool ran at least once */
    double hs[12];
    CHECK(ig_debug_nuis_hist_stats(c, hs) == 0);
    std::fprintf(stderr, "[harness] histogram tier: evaluations %.0f (model calls %ld), rejected there %.0f, accepted there %.0f, void %.0f, walks %.0f, builds %.0f\n",
                 hs[0], g_hists, hs[1], hs[2], hs[3], hs[6], hs[7]);
    CHECK(g_hists > 0 && hs[1] > 0 && hs[2] > 0 && hs[3] > 0 && hs[6] > 0);
    ig_destroy(c);
    ig_destroy(nullptr);
    std::puts("host logic harness ok");
    return 0;
}
