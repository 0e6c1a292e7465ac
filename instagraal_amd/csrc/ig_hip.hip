/*
 * ig_hip.hip -- MI355X (gfx950) implementation of instaGRAAL's per-move scoring path behind the
 * C ABI of include/instagraal_hip.h.  Design notes: DESIGN.md.  Reference being replaced:
 *   /root/reference/src/instagraal/kernels/kernel_sparse_adapt.cu  ("KA", 38 CUDA kernels)
 *   /root/reference/src/instagraal/cuda_lib_gl_single.py           ("CL", pycuda host code)
 *
 * One translation unit; the device code lives in the parts included below, this file is the host side
 * (handles, uploads, the launch sequences, the extern "C" entry points).
 *
 * A batch of W moves (CL:1401-1465 each) is one launch sequence on one stream, no host round trip inside:
 *   k_gather        O(N)            local fragment lists of the touched contigs, uniq-mutation lists, flags
 *   k_mutate        W x C x 25 WGs  one candidate genome per workgroup, operators applied in place on the local window,
 *                                   coordinate columns + zero-pixel sums
 *   k_offsets       1 workgroup     slice-list starts in the pool
 *   k_slice         row-parallel    CSR rows of the touched contigs -> compacted slice lists (the focal contig's rows once per move)
 *   k_screen_tail   the hot kernel  two tiers, first: every (contact, column) term in float with a rigorous bound; its first
 *                                   workgroups: quirk Q5, the last S_c mod 64 contacts of every list (k_tail on a second
 *                                   stream where a batch is not screened)
 *   k_contend, k_worklist           the columns that can still win, and their work items
 *   k_score_list    exact           one exact Rippe/Poisson term per (contact, column), 64-bit integer sums: contenders only
 *   k_records, k_predict, k_delta   slot-major score records; predicted windowed winners and their exact deltas
 *   k_decide_batch  1 wave          in-order decisions with the live scalars
 *   k_commit_batch  1 workgroup     winners applied together, exact genome-distance deltas, result records
 * plus the one-move kernels k_scores / k_delta / k_apply / k_post / k_commit (single moves, windowed winners), and the
 * from-scratch pass over all contacts: k_pack_tab_sig / k_nuis_prepare, k_tile_trans, k_full_nz_tiled; the
 * nuisance step's screened pass: k_hist_build / k_hist_walk / k_hist_eval (tier 0), k_full_diff_tiled (tier 1).
 *
 * Environment knobs (tuning and tests only): IG_BATCH_W (moves per batch, default 24), IG_POOL_ENTRIES (slice pool size: a small
 * one forces the overflow / re-run path), IG_WIDE_LISTS=1 (12-byte slice entries even where the packed 8-byte form fits),
 * IG_NO_HOST_FLAG=1 (outcomes by copy + synchronise instead of the polled mapped copies), IG_SCREEN=0 / IG_SCREEN_VERIFY=1 (every
 * column exact / exact and screened, bounds checked), IG_FUSE_TAIL=0, IG_SLICE_SHARE=0, IG_FULL_TILED=0, IG_FULL_HIST=0 (the
 * earlier forms of those kernels), IG_FULL_GRID / IG_FULL_GRID_SIDE (persistent grid of the from-scratch pass), IG_NUIS_W /
 * IG_NUIS_WMAX (moves scored ahead in the nuisance-on loop), IG_NUIS_SCREEN / IG_NUIS_HIST / IG_NUIS_SCREEN_VERIFY (the screened
 * nuisance pass and its tiers), IG_NUIS_ASYNC=0 (no helper thread), IG_NUIS_BG=1 (the next batch in the background), IG_ABLATE
 * (bit 1: every column of k_score_list through the checked path).
 */
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "ig_common.cuh"
#include "ig_model.cuh"
#include "ig_kernels_setup.cuh"
#include "ig_kernels_score.cuh"
#include "ig_kernels_screen.cuh"
#include "ig_kernels_commit.cuh"
#include "ig_kernels_nuis.cuh"

/* ================================================================== host side */
static void flush_pending_sums(ig_ctx* c); /* behind a decisively accepted nuisance step: see k_nuis_promote */

/* ---- the launches of a run's NEXT step on a helper thread ------------------------------------------------------------------
 * Once a step's Metropolis test reads 600 KB instead of 160 MB (ig_kernels_nuis.cuh, tier 0) a (move, nuisance step) pair is
 * bound by the host: the caller's proposal arithmetic and the half-dozen launches of the next step, one after the other on one
 * thread.  ig_nuis_step_next hands the launches to this thread and returns; the caller computes its next proposal meanwhile.
 * EVERY entry point that takes a handle waits for the thread first (IG_JOIN): outside the task nothing is shared.  An error of
 * the deferred launches is reported by the entry point that waits for them.  IG_NUIS_ASYNC=0: no helper thread. */
struct NuisWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> state{0}; /* 0 idle, 1 a task is waiting / running, 2 quit */
    int rc = 0;
    std::string err;
    int kind = 0; /* the task: 0 ig_nuis_step_begin, 1 a chain of pairs (ig_nuis_chain_begin) */
    int move = 0;
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mean_kb = 0;
    int n_sets = 0;
    std::thread::id tid;
};
static int nuis_step_begin_impl(ig_ctx* c, int32_t move, const float p_test[8], float mean_subfrag_kb);
static int nuis_chain_impl(ig_ctx* c, int32_t move, int32_t n_sets, float mean_subfrag_kb);
static int nuis_join(ig_ctx* c)
{
    NuisWorker* w = c->worker;
    if (!w || std::this_thread::get_id() == w->tid) return 0;
    if (w->state.load(std::memory_order_acquire) == 1) {
        for (unsigned spin = 0; w->state.load(std::memory_order_acquire) == 1; spin++)
            if ((spin & 0x3ff) == 0x3ff) std::this_thread::yield();
    }
    if (w->rc) {
        const int rc = w->rc;
        w->rc = 0;
        (void)rc;
        return fail("%s", w->err.c_str());
    }
    return 0;
}
static void nuis_worker_main(ig_ctx* c)
{
    NuisWorker* w = c->worker;
    hipSetDevice(c->device);
    for (;;) {
        int st = w->state.load(std::memory_order_acquire);
        if (st == 0) { /* spin for a while (a step of a run is tens of microseconds away), then sleep */
            bool got = false;
            for (unsigned spin = 0; spin < 200000 && !got; spin++) got = w->state.load(std::memory_order_acquire) != 0;
            if (!got) {
                std::unique_lock<std::mutex> lk(w->mu);
                w->cv.wait(lk, [&] { return w->state.load(std::memory_order_acquire) != 0; });
            }
            continue;
        }
        if (st == 2) return;
        const int rc = w->kind == 1 ? nuis_chain_impl(c, w->move, w->n_sets, w->mean_kb) : nuis_step_begin_impl(c, w->move, w->p, w->mean_kb);
        if (rc) w->err = g_err;
        w->rc = rc;
        w->state.store(0, std::memory_order_release);
    }
}
/* kind 0: p_test = the step's test parameters; kind 1: a chain over the n_sets sets staged in chain_in_host */
static int nuis_defer(ig_ctx* c, int kind, int32_t move, const float* p_test, float mean_subfrag_kb, int n_sets)
{
    static const int s_async = getenv("IG_NUIS_ASYNC") ? atoi(getenv("IG_NUIS_ASYNC")) : 1;
    if (!s_async) return kind == 1 ? nuis_chain_impl(c, move, n_sets, mean_subfrag_kb) : nuis_step_begin_impl(c, move, p_test, mean_subfrag_kb);
    if (!c->worker) {
        c->worker = new NuisWorker();
        c->worker->th = std::thread(nuis_worker_main, c);
        c->worker->tid = c->worker->th.get_id();
    }
    NuisWorker* w = c->worker;
    if (nuis_join(c)) return -1;
    w->kind = kind;
    w->move = move;
    if (p_test) memcpy(w->p, p_test, sizeof w->p);
    w->mean_kb = mean_subfrag_kb;
    w->n_sets = n_sets;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->state.store(1, std::memory_order_release);
    }
    w->cv.notify_one();
    return 0;
}
static int nuis_defer_step_begin(ig_ctx* c, int32_t move, const float p_test[8], float mean_subfrag_kb)
{
    return nuis_defer(c, 0, move, p_test, mean_subfrag_kb, 0);
}
static void nuis_worker_stop(ig_ctx* c)
{
    NuisWorker* w = c->worker;
    if (!w) return;
    (void)nuis_join(c);
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->state.store(2, std::memory_order_release);
    }
    w->cv.notify_one();
    w->th.join();
    delete w;
    c->worker = nullptr;
}
#define IG_JOIN(c)                    \
    do {                              \
        if (nuis_join(c)) return -1;  \
    } while (0)

template <class T>
static int dalloc(T** p, size_t n)
{
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) return fail("hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    return 0;
}
#define DALLOC(p, n)                     \
    do {                                 \
        if (dalloc(&(p), (n))) return -1; \
    } while (0)

enum { T_GATHER = 0, T_MUTATE, T_SCORE, T_FINALIZE, T_DELTA, T_APPLY, T_POST, T_COMMIT, T_SLICE, T_ARGMAX, T_SCREEN, T_DIFF, T_PROBE, T_COUNT };
static const char* kTimerNames[T_COUNT] = {"gather", "mutate", "score", "finalize", "delta", "apply", "post", "commit", "slice", "argmax", "screen", "diff", "probe"};

struct TimedLaunch {
    ig_ctx* c;
    int id;
    hipEvent_t a, b;
    hipStream_t st;
    TimedLaunch(ig_ctx* ctx, int which, hipStream_t stream = nullptr) : c(ctx), id(which), a(nullptr), b(nullptr), st(stream ? stream : ctx->stream)
    {
        if (c->timing && !((c->timing_mask >> id) & 1u)) return;
        /* every timing_every-th launch only: an event record between two kernels of a stream costs ~6 us of idle queue */
        if (c->timing && c->timing_every > 1 && (c->timers[id].seen++ % c->timing_every) != 0) return;
        if (c->timing) { /* events are recycled (drain_timers): creating a pair costs microseconds the timed call should not pay */
            if (c->ev_pool.size() >= 2) {
                a = c->ev_pool.back();
                c->ev_pool.pop_back();
                b = c->ev_pool.back();
                c->ev_pool.pop_back();
            } else {
                hipEventCreate(&a);
                hipEventCreate(&b);
            }
            hipEventRecord(a, st);
        }
    }
    ~TimedLaunch()
    {
        if (c->timing && a) {
            hipEventRecord(b, st);
            c->timers[id].ev.emplace_back(a, b);
        }
    }
};

static void drain_timers(ig_ctx* c)
{
    for (int i = 0; i < T_COUNT; i++) {
        for (auto& pr : c->timers[i].ev) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                c->timers[i].total_ms += ms;
                c->timers[i].n++;
            }
            c->ev_pool.push_back(pr.first);
            c->ev_pool.push_back(pr.second);
        }
        c->timers[i].ev.clear();
    }
}

extern "C" const char* ig_last_error(void) { return g_err.c_str(); }

extern "C" int ig_create(int device_id, ig_ctx** out)
{
    if (!out) return fail("ig_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail("ig_create: no HIP device (the MI355X path has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail("ig_create: device %d out of range (%d devices)", device_id, n);
    HIPCK(hipSetDevice(device_id));
    ig_ctx* c = new ig_ctx();
    memset((void*)&c->st, 0, sizeof c->st);
    memset((void*)&c->tab, 0, sizeof c->tab);
    memset((void*)&c->tab_prev, 0, sizeof c->tab_prev);
    memset((void*)&c->mb, 0, sizeof c->mb);
    c->device = device_id;
    c->own_stream = true;
    c->host_bo = nullptr;
    c->host_bo_dev = nullptr;
    c->bo_seq = 0;
    c->w_ema = 0.0;
    c->n_contigs_seen = 0;
    c->rank = 0;
    c->world = 1;
    c->N = c->M = 0;
    c->Z = 0;
    c->max_count = 0;
    c->st_block = nullptr;
    c->sub_tab = nullptr;
    c->rowptr = nullptr;
    c->cc = nullptr;
    c->crow = nullptr;
    c->tabrec = nullptr;
    c->tiled_cc = nullptr;
    c->tile_work = nullptr;
    c->n_tile_work = 0;
    c->tile_trace = c->diff_trace = nullptr;
    c->init_prev = c->init_next = c->orientable = nullptr;
    c->black = nullptr;
    c->batch_out = nullptr;
    c->own_tag = c->own_idx = nullptr;
    c->dirty_buf = nullptr;
    c->d_results = nullptr;
    c->results_cap = 0;
    c->d_frags = c->d_cands = nullptr;
    c->cands_cap = 0;
    c->prev_touched = nullptr;
    c->pz_tab = c->pz_tab1 = nullptr;
    c->score_const = nullptr;
    c->full_const = nullptr;
    c->screen_const = nullptr;
    c->screen_worst = nullptr;
    c->n_screen_cols = c->n_screen_cont = 0;
    c->pz_n = c->pz_n1 = 0;
    c->timing_mask = 0xffff;
    c->side_busy = true;
    c->timing_every = 1;
    c->timing = false;
    c->n_batches = c->n_batch_committed = c->n_batch_pending = c->n_batch_predicted = 0;
    c->up_moves = c->up_max_c = 0;
    c->own_begin = c->own_end = 0;
    c->own_screened = 0;
    c->exact_grid = 0;
    c->max_L = c->max_SL = 0;
    c->full_windows = false;
    c->host_max = nullptr;
    for (int i = 0; i < T_COUNT; i++) {
        c->timers[i].name = kTimerNames[i];
        c->timers[i].total_ms = 0;
        c->timers[i].n = 0;
    }
    c->have_contacts = c->have_sub = c->have_state = c->have_init = c->have_params = false;
    c->init_links_inverse = true;
    HIPCK(hipStreamCreate(&c->stream));
    HIPCK(hipStreamCreate(&c->stream2));
    HIPCK(hipStreamCreate(&c->stream3));
    HIPCK(hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    c->host_nuis = nullptr;
    c->host_nuis_dev = nullptr;
    c->nuis_in_flight = false;
    c->diff_const = nullptr;
    c->scratch_diff = nullptr;
    c->tile_partial0 = nullptr;
    c->diff_seq = 0;
    c->nuis_diff = c->nuis_exact_queued = c->nuis_screen_rejected = false;
    c->scratch_exact = nullptr;
    c->nuis_sums_pending = c->nuis_accept_certain = false;
    c->exact_seq = 0;
    HIPCK(hipEventCreateWithFlags(&c->ev_exact, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&c->ev_walk, hipEventDisableTiming));
    c->worker = nullptr;
    c->probe_scr = nullptr;
    c->probe_void = nullptr;
    c->last_moved = true;
    c->n_accepts = 0;
    c->nh = NuisHist{nullptr, nullptr, nullptr, 0};
    c->scratch_hist = nullptr;
    c->nh_valid = false;
    c->nh_pending_slot = -1;
    c->nh_tracking = false;
    c->nh_policy_on = true;
    c->nh_p_changed = 0.1;
    c->nuis_tier = 1;
    c->nuis_tiles_listed = false;
    for (double& v : c->nhs) v = 0.0;
    for (double& v : c->nscr) v = 0.0;
    HIPCK(hipEventCreateWithFlags(&c->ev_slice, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
    DALLOC(c->glob, 1);
    DALLOC(c->scratch8, 8);
    DALLOC(c->scratch_nuis, 8);
    HIPCK(hipMemset(c->glob, 0, sizeof(Glob)));
    DALLOC(c->lgf_tab, LGF_TAB);
    /* log10(ob!) table (KA:111-124, 251-270): the 15 float-factorial constants on the host, the rest on the device */
    double small[15];
    for (int k = 0; k < 15; k++) {
        float r = 1;
        if (k < 10) {
            for (int q = 1; q <= k; q++) r = r * q;
        } else {
            r = ig_powf((float)k, (float)k, ig_tab()) * ig_expf(-(float)k, ig_tab()) * __builtin_sqrtf((float)(2 * 3.14159265358979323846 * (float)k));
        }
        small[k] = ig_log10((double)r, ig_tab());
    }
    Glob hg;
    memset(&hg, 0, sizeof hg);
    for (int k = 0; k < 15; k++) hg.lgf[k] = small[k];
    const int lb[6] = {1, 3, 5, 10, 20, 50}; /* CL:417 */
    for (int k = 0; k < 6; k++) hg.list_bounds[k] = lb[k];
    hg.slice_nb = 50 * 4;
    HIPCK(hipMemcpy(c->glob, &hg, sizeof hg, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_lgf_table, dim3((LGF_TAB + 255) / 256), dim3(256), 0, c->stream, c->lgf_tab, c->glob->lgf);
    HIPCK(hipStreamSynchronize(c->stream));
    *out = c;
    return 0;
}

/* the per-window arrays of the move buffers (strides sN / sM) */
static void free_window_buffers(MoveBuf& m)
{
    hipFree(m.Lloc);
    hipFree(m.lbloc);
    hipFree(m.slloc);
    hipFree(m.subs);
    hipFree(m.rowcnt);
    hipFree(m.rowbe);
    hipFree(m.coords);
    hipFree(m.loc);
    m.Lloc = m.lbloc = m.slloc = m.subs = m.rowcnt = nullptr;
    m.rowbe = nullptr;
    m.coords = nullptr;
    m.loc = nullptr;
    m.sN = m.sM = 0;
}
static void free_slice_pool(MoveBuf& m)
{
    hipFree(m.sl_li);
    hipFree(m.sl_lj);
    hipFree(m.sl_ob);
    hipFree(m.sl_pk);
    m.sl_li = m.sl_lj = m.sl_ob = nullptr;
    m.sl_pk = nullptr;
    m.pool_cap = 0;
}
/* one set of batch buffers */
static void free_movebuf(MoveBuf& m)
{
    free_window_buffers(m);
    free_slice_pool(m);
    hipFree(m.meta);
    hipFree(m.cmeta);
    hipFree(m.part);
    hipFree(m.scores);
    hipFree(m.slbound);
    hipFree(m.sloff);
    hipFree(m.qpart);
    hipFree(m.ctl);
    hipFree(m.sinfo);
    hipFree(m.rec);
    hipFree(m.scr);
    hipFree(m.scr_void);
    hipFree(m.scr_ub);
    hipFree(m.cont);
    hipFree(m.ident);
    hipFree(m.work);
    hipFree(m.slot_items);
    hipFree(m.order);
    hipFree(m.tail_n);
    hipFree(m.tail_ent);
    memset((void*)&m, 0, sizeof m);
}
static void free_move_buffers(ig_ctx* c)
{
    free_movebuf(c->mb);
    hipFree(c->own_tag);
    hipFree(c->own_idx);
    c->own_tag = c->own_idx = nullptr;
    hipFree(c->batch_out);
    hipFree(c->dirty_buf);
    c->dirty_buf = nullptr;
    c->batch_out = nullptr;
}

extern "C" void ig_destroy(ig_ctx* c)
{
    if (!c) return;
    nuis_worker_stop(c);
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->stream2);
    hipStreamDestroy(c->stream2);
    hipStreamSynchronize(c->stream3);
    hipStreamDestroy(c->stream3);
    hipEventDestroy(c->ev_gathered);
    hipEventDestroy(c->ev_main);
    hipEventDestroy(c->ev_exact);
    hipEventDestroy(c->ev_walk);
    hipFree(c->probe_scr);
    hipFree(c->probe_void);
    hipFree(c->nh.bins);
    hipFree(c->nh.dh);
    hipFree(c->nh.misc);
    hipFree(c->scratch_hist);
    hipFree(c->scratch_exact);
    hipFree(c->chain_sets);
    hipFree(c->chain_in);
    hipFree(c->chain_tests);
    hipFree(c->chain_out16);
    hipFree(c->chain_zs);
    if (c->chain_in_host) hipHostFree(c->chain_in_host);
    if (c->host_nuis) hipHostFree(c->host_nuis);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    hipFree(c->scratch_accept);
    hipFree(c->scratch_nuis);
    hipEventDestroy(c->ev_slice);
    hipEventDestroy(c->ev_tail);
    if (c->host_bo) hipHostFree(c->host_bo);
    if (c->host_max) hipHostFree(c->host_max);
    drain_timers(c);
    free_move_buffers(c);
    hipFree(c->st_block);
    hipFree(c->tab.dist);
    hipFree(c->tab_prev.dist);
    hipFree(c->sub_tab);
    hipFree(c->rowptr);
    hipFree(c->cc);
    hipFree(c->crow);
    hipFree(c->tabrec);
    hipFree(c->tiled_cc);
    hipFree(c->tile_work);
    hipFree(c->tile_hist);
    hipFree(c->tile_sig);
    hipFree(c->tile_info);
    hipFree(c->tile_dyn);
    hipFree(c->tile_dyn_list);
    hipFree(c->tile_partial);
    hipFree(c->tile_partial0);
    hipFree(c->diff_const);
    hipFree(c->scratch_diff);
    hipFree(c->init_prev);
    hipFree(c->init_next);
    hipFree(c->orientable);
    hipFree(c->black);
    hipFree(c->lgf_tab);
    hipFree(c->score_const);
    hipFree(c->full_const);
    hipFree(c->screen_const);
    hipFree(c->screen_worst);
    hipFree(c->glob);
    hipFree(c->d_results);
    hipFree(c->d_frags);
    hipFree(c->d_cands);
    hipFree(c->touched_bits);
    hipFree(c->prev_touched);
    hipFree(c->pz_tab);
    hipFree(c->scratch8);
    hipFree(c->pz_tab1);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int ig_sync(ig_ctx* c)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    flush_pending_sums(c);
    HIPCK(hipStreamSynchronize(c->stream));
    drain_timers(c);
    return 0;
}

extern "C" int ig_set_stream(ig_ctx* c, void* s)
{
    IG_JOIN(c);
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->own_stream) hipStreamDestroy(c->stream);
    if (s) {
        c->stream = (hipStream_t)s;
        c->own_stream = false;
    } else {
        HIPCK(hipStreamCreate(&c->stream));
        c->own_stream = true;
    }
    return 0;
}

/* the from-scratch likelihood of the non-zero pixels under tables `t` and parameter set `which` -> out[0..1] */
__global__ void k_set_par(Glob* g, int which, ig_params p, float mean_kb)
{
    g->par[which] = p;
    g->mean_kb = mean_kb;
}
static int g_full_hist = -1;
/* the from-scratch pass over all contacts under parameter set `which` -> out[0..1]; zero_out: the zero-pixel pass too
 * (-> zero_out[0..5], k_full_zero).  p_host: the set's parameters, not yet on the device (a nuisance step's test set): the
 * launches that would set them, build their tables and clear `out` (8 values) are part of the pass.
 * Returns true when the zero-pixel pass was taken care of. */
static bool launch_full_nz(ig_ctx* c, const Tables& t, int which, long long* out, PzTab pz, hipStream_t stream = nullptr,
                           long long* zero_out = nullptr, const ig_params* p_host = nullptr, float mean_kb = 0.0f)
{
    const int s_wgs = 8 * 256;
    if (!stream) stream = c->stream;
    static const int s_tiled = getenv("IG_FULL_TILED") ? atoi(getenv("IG_FULL_TILED")) : 1;
    if (g_full_hist < 0) g_full_hist = 1; /* ig_debug_set_full_hist(0): read every tile's contacts */
    const int s_hist = g_full_hist;
    const bool tiled = s_tiled && c->tiled_cc && c->n_tile_work > 0;
    const int n_pack = (c->M + FULL_TB - 1) / FULL_TB;
    if (p_host && tiled) {
        const int n_const = (std::max(std::max(pz.n, LDS_PZ + 2), std::max((int)IG_TAB_SIZE, LDS_LGF)) + 255) / 256;
        hipLaunchKernelGGL(k_nuis_prepare, dim3(n_pack + n_const), dim3(256), 0, stream, c->glob, which, *p_host, mean_kb, (float*)pz.v, pz.n, c->lgf_tab,
                           c->full_const, out, t, c->M, c->tabrec, c->tile_sig, FULL_TB, c->tile_dyn, n_pack);
    } else {
        if (p_host) {
            hipLaunchKernelGGL(k_set_par, dim3(1), dim3(1), 0, stream, c->glob, which, *p_host, mean_kb);
            if (pz.n > 0) hipLaunchKernelGGL(k_build_pz, dim3((pz.n + 255) / 256), dim3(256), 0, stream, c->glob, (float*)pz.v, pz.n, which);
            hipMemsetAsync(out, 0, 8 * sizeof(long long), stream);
        }
        if (c->Z <= 0) return false;
        if (tiled)
            hipLaunchKernelGGL(k_pack_tab_sig, dim3(n_pack), dim3(256), 0, stream, t, c->M, c->tabrec, c->tile_sig, FULL_TB, c->tile_dyn);
        else
            hipLaunchKernelGGL(k_pack_tab, dim3((c->M + 255) / 256), dim3(256), 0, stream, t, c->M, c->tabrec);
        /* the tables of this parameter set (k_score_list's own block is parameter set 0's) */
        hipLaunchKernelGGL(k_build_score_const, dim3((LDS_PZ + 2 + 255) / 256), dim3(256), 0, stream, c->glob, pz, c->lgf_tab, c->full_const, which);
    }
    if (tiled) {
        /* the off-diagonal tiles: summed from their count histograms where the two blocks share no contig, else put on the
         * list; in the same launch (blocks behind those of the tiles) the zero-pixel pass */
        const int per = TILE_TRANS_THREADS / 64;
        const int n_trans = (c->n_tile_info + per - 1) / per, n_zero = zero_out ? std::min(256, std::max(32, c->M / 4096)) : 0; /* the zero-pixel pass: ~4 sub-fragments per thread */
        if (n_trans + n_zero > 0)
            hipLaunchKernelGGL(k_tile_trans, dim3(n_trans + n_zero), dim3(TILE_TRANS_THREADS), 0, stream, c->tile_info, c->n_tile_info, c->tile_sig,
                               c->tile_hist, c->full_const, (TileDyn*)c->tile_dyn, c->tile_dyn_list, s_hist, c->tile_partial, n_trans, t, c->glob,
                               which, c->M, zero_out);
        /* persistent workgroups over the static items and then the list */
        /* two per CU; next to a move (the nuisance step's pass, on its own stream) one per CU: persistent workgroups that took
         * every wave slot would keep the move's kernels waiting until the pass is over (k_decide_batch: 70 instead of 12 us),
         * and the pass is bound by its arithmetic, not by its occupancy */
        static const int s_grid = getenv("IG_FULL_GRID") ? atoi(getenv("IG_FULL_GRID")) : 512;
        const int s_grid_side = 256;
        const int grid = std::min(c->n_tile_work, (stream != c->stream && c->side_busy) ? s_grid_side : s_grid);
        if (grid > 0)
            hipLaunchKernelGGL(k_full_nz_tiled, dim3(grid), dim3(FULL_TILED_THREADS), sizeof(FullTiledLds), stream, c->tile_work, c->tiled_cc,
                               c->tabrec, t.len, c->full_const, c->lgf_tab, c->M, pz.n, out, c->n_tile_static, (TileDyn*)c->tile_dyn,
                               c->tile_dyn_list, c->tile_trace, (zero_out && c->pub_sums) ? c->pub_sums : nullptr, ++c->sums_seq, c->tile_partial,
                               n_trans);
        if (zero_out && c->pub_sums) c->nuis_pub_sums = true;
        return zero_out != nullptr;
    }
    hipLaunchKernelGGL(k_full_nz, dim3(s_wgs), dim3(256), 0, stream, c->crow, c->cc, c->tabrec, t.len, c->full_const, c->lgf_tab,
                       (long long)c->Z, pz.n, out);
    return false;
}

/* The per-window arrays are strided by the largest window the genome can produce NOW: a window is the contig of the focal
 * bin plus the contig of a candidate, so twice the longest contig (Glob.max_L / max_SL: exact after a recount, raised by
 * every committed move that changed the genome by its window's total -- a batch can at most double it) with headroom,
 * not the whole genome: 12 MB instead of 0.84 GB per slot at the headline shape.  Grown when the bound grows. */
static int ensure_window_buffers(ig_ctx* c, MoveBuf& m)
{
    if (!m.capC || !m.capW) return 0;
    const int need_n = std::min(c->N, std::max(2 * c->max_L, 1)), need_m = std::min(c->M, std::max(2 * c->max_SL, 1));
    const bool s_full = c->full_windows; /* strides = the whole genome */
    if (m.Lloc && (s_full ? (m.sN == c->N && m.sM == c->M) : (m.sN >= need_n && m.sM >= need_m))) return 0;
    const int sN = s_full ? c->N : std::min(c->N, std::max(256, need_n + need_n / 2));
    const int sM = s_full ? c->M : std::min(c->M, std::max(768, need_m + need_m / 2));
    HIPCK(hipStreamSynchronize(c->stream));
    free_window_buffers(m);
    const size_t C = (size_t)m.capC * m.capW;
    DALLOC(m.Lloc, C * sN);
    DALLOC(m.lbloc, C * sN);
    DALLOC(m.slloc, C * sN);
    DALLOC(m.subs, C * sM);
    DALLOC(m.rowcnt, C * sM);
    DALLOC(m.rowbe, C * sM);
    DALLOC(m.coords, C * sM * NSLOT);
    DALLOC(m.loc, C * NSLOT * NDYN * sN);
    m.sN = sN;
    m.sM = sM;
    return 0;
}
static int ensure_window_buffers(ig_ctx* c) { return ensure_window_buffers(c, c->mb); }

/* the slice pool: room for the lists of a batch.  A slot whose lists do not fit behind the earlier ones is re-run; when
 * the FIRST slot of a batch does not fit, the host grows the pool (grow_slice_pool) and repeats the batch.  The worst case
 * of one slot is capC x Z entries (windows = the whole genome); it starts at 2 Z. */
static int alloc_slice_pool(MoveBuf& m, size_t entries)
{
    free_slice_pool(m);
    if (m.packed) {
        DALLOC(m.sl_pk, entries + 8192); /* slack: k_screen's look-ahead loads run past the end of the last list */
    } else {
        DALLOC(m.sl_li, entries);
        DALLOC(m.sl_lj, entries);
        DALLOC(m.sl_ob, entries);
    }
    m.pool_cap = (long long)entries;
    return 0;
}
static int alloc_slice_pool(ig_ctx* c, size_t entries) { return alloc_slice_pool(c->mb, entries); }
static size_t slice_pool_max(const ig_ctx* c) { return (size_t)std::max<long long>(c->Z, 1) * (size_t)std::max(c->mb.capC, 1); }
static int grow_slice_pool(ig_ctx* c)
{
    const size_t mx = slice_pool_max(c);
    if ((size_t)c->mb.pool_cap >= mx) return fail("the slice pool already holds the worst case of a move (%zu entries) and the move does not fit", mx);
    HIPCK(hipStreamSynchronize(c->stream));
    return alloc_slice_pool(c, std::min(mx, (size_t)c->mb.pool_cap * 4));
}

static int alloc_movebuf(ig_ctx* c, MoveBuf& m, int capC, int capW, int want_packed, size_t pool_entries);
static int ensure_move_buffers(ig_ctx* c, int capC, int capW = 1)
{
    const int want_packed = (!(getenv("IG_WIDE_LISTS") && atoi(getenv("IG_WIDE_LISTS"))) && c->M < (1 << 20) && c->max_count < (1 << 24)) ? 1 : 0;
    if (c->mb.capC >= capC && c->mb.capW >= capW && c->mb.N == c->N && c->mb.M == c->M && c->mb.packed == want_packed)
        return ensure_window_buffers(c);
    if (c->N == 0 || c->M == 0) return 0;
    capC = std::max(capC, c->mb.capC);
    capW = std::max(capW, c->mb.capW);
    HIPCK(hipStreamSynchronize(c->stream));
    free_move_buffers(c);
    const size_t N = c->N;
    c->mb.capC = capC; /* (slice_pool_max reads it) */
    size_t Zc = std::min(slice_pool_max(c), std::max<size_t>((size_t)1 << 22, 2 * (size_t)std::max<long long>(c->Z, 1))); /* 32 MB, or 2 Z */
    if (const char* e = getenv("IG_POOL_ENTRIES")) /* tests: a small pool forces the overflow / re-run / growth paths */
        Zc = std::max<size_t>((size_t)atoll(e), 1024);
    if (alloc_movebuf(c, c->mb, capC, capW, want_packed, Zc)) return -1;
    DALLOC(c->own_tag, N);
    DALLOC(c->own_idx, N);
    HIPCK(hipMemset(c->own_tag, 0xff, N * sizeof(int)));
    DALLOC(c->batch_out, 12);
    if (!c->host_bo && !(getenv("IG_NO_HOST_FLAG") && atoi(getenv("IG_NO_HOST_FLAG")))) {
        /* the batch outcome is also written to mapped host memory (commit_loop polls it); without it: copy + synchronise */
        int* hp = nullptr;
        if (hipHostMalloc((void**)&hp, 12 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                memset(hp, 0, 12 * sizeof(int));
                c->host_bo = hp;
                c->host_bo_dev = (int*)dp;
            } else {
                hipHostFree(hp);
            }
        }
        (void)hipGetLastError();
    }
    DALLOC(c->dirty_buf, 2 * IG_MAX_BATCH + 4);
    return ensure_window_buffers(c);
}

/* the arrays of one set of batch buffers but the per-window ones (ensure_window_buffers) */
static int alloc_movebuf(ig_ctx* c, MoveBuf& m, int capC, int capW, int want_packed, size_t pool_entries)
{
    const size_t C = (size_t)capC * capW;
    m.N = c->N;
    m.M = c->M;
    m.capC = capC;
    m.capW = capW;
    m.packed = want_packed; /* IG_WIDE_LISTS=1 (tests) forces the 12-byte form */
    if (alloc_slice_pool(m, pool_entries)) return -1;
    DALLOC(m.slbound, C * SLICE_SEG);
    DALLOC(m.sloff, C * SLICE_SEG);
    DALLOC(m.meta, C);
    DALLOC(m.cmeta, C * NSLOT * NCODE);
    DALLOC(m.part, C * P_STRIDE);
    DALLOC(m.qpart, C * Q_STRIDE);
    DALLOC(m.scores, C * IG_N_TMP_STRUCT);
    DALLOC(m.ctl, (size_t)capW);
    DALLOC(m.sinfo, C * NSLOT);
    m.rec_stride = rec_bytes_per_slot(capC);
    DALLOC(m.rec, m.rec_stride * (size_t)capW);
    DALLOC(m.scr, C * NSLOT);
    DALLOC(m.scr_void, C);
    DALLOC(m.scr_ub, C);
    DALLOC(m.cont, C);
    DALLOC(m.ident, C);
    HIPCK(hipMemset(m.ident, 0, C * sizeof(unsigned)));
    m.work_cap = (int)std::min<size_t>((size_t)1 << 20, 4 * C * NSLOT * SLICE_SEG + 4096);
    DALLOC(m.work, (size_t)m.work_cap + 32);
    DALLOC(m.slot_items, (size_t)capW * 8);
    DALLOC(m.order, C + (size_t)capW);
    HIPCK(hipMemset(m.order, 0, (C + (size_t)capW) * sizeof(int)));
    DALLOC(m.tail_n, C);
    DALLOC(m.tail_ent, C * 3 * 64);
    HIPCK(hipMemset(m.tail_n, 0xff, C * sizeof(int)));
    HIPCK(hipMemset(m.cont, 0xff, C * sizeof(unsigned)));
    HIPCK(hipMemset(m.cmeta, 0, C * NSLOT * NCODE * sizeof(ColMeta)));
    HIPCK(hipMemset(m.slbound, 0, C * SLICE_SEG * sizeof(long long)));
    HIPCK(hipMemset(m.ctl, 0, (size_t)capW * sizeof(MoveCtl)));
    return 0;
}

extern "C" int ig_upload_contacts(ig_ctx* c, const int32_t* row, const int32_t* col, const int32_t* cnt, int64_t Z, int32_t M,
                                  int32_t rank, int32_t world)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (Z < 0 || M <= 0) return fail("ig_upload_contacts: bad sizes");
    if (world < 1 || rank < 0 || rank >= world) return fail("ig_upload_contacts: bad shard %d/%d", rank, world);
    if (c->M && c->M != M) return fail("ig_upload_contacts: M=%d does not match the sub-fragment table (%d)", M, c->M);
    c->nh_valid = false;
    c->nh_pending_slot = -1;
    std::vector<long long> rp((size_t)M + 1, 0);
    std::vector<int2> cc((size_t)Z);
    int max_count = 0;
    for (int64_t k = 0; k < Z; k++) {
        max_count = std::max(max_count, (int)cnt[k]);
        const int r = row[k], q = col[k];
        if (r < 0 || r >= M || q <= r || q >= M) return fail("ig_upload_contacts: entry %lld (%d,%d) is not strict upper triangle", (long long)k, r, q);
        if (k > 0 && (row[k - 1] > r || (row[k - 1] == r && col[k - 1] >= q)))
            return fail("ig_upload_contacts: entries must be row-major sorted and distinct (at %lld)", (long long)k);
        rp[(size_t)r + 1]++;
        cc[(size_t)k] = make_int2(q, cnt[k]);
    }
    for (int i = 0; i < M; i++) rp[(size_t)i + 1] += rp[(size_t)i];
    hipFree(c->rowptr);
    hipFree(c->cc);
    hipFree(c->crow);
    hipFree(c->tabrec);
    DALLOC(c->rowptr, (size_t)M + 1);
    DALLOC(c->cc, (size_t)Z);
    DALLOC(c->crow, (size_t)std::max<int64_t>(Z, 1));
    DALLOC(c->tabrec, (size_t)M);
    if (!c->full_const) DALLOC(c->full_const, 1);
    if (Z) HIPCK(hipMemcpy(c->crow, row, (size_t)Z * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->rowptr, rp.data(), ((size_t)M + 1) * sizeof(long long), hipMemcpyHostToDevice));
    if (Z) HIPCK(hipMemcpy(c->cc, cc.data(), (size_t)Z * sizeof(int2), hipMemcpyHostToDevice));
    /* the tiled copy for the from-scratch pass (k_full_nz_tiled): counting sort of the contacts by (row block, column block) */
    hipFree(c->tiled_cc);
    hipFree(c->tile_work);
    hipFree(c->tile_hist);
    hipFree(c->tile_sig);
    hipFree(c->tile_info);
    hipFree(c->tile_dyn);
    hipFree(c->tile_dyn_list);
    hipFree(c->tile_partial);
    hipFree(c->tile_partial0);
    c->tile_partial0 = nullptr;
    c->tile_partial = nullptr;
    c->tile_info = nullptr;
    c->tile_dyn = nullptr;
    c->tile_dyn_list = nullptr;
    c->tiled_cc = nullptr;
    c->tile_work = nullptr;
    c->tile_hist = nullptr;
    c->tile_sig = nullptr;
    c->n_tile_work = 0;
    {
        const int64_t nb = ((int64_t)M + FULL_TB - 1) / FULL_TB;
        if (Z > 0 && nb <= 2048) {
            static bool s_attr = false;
            if (!s_attr) {
                HIPCK(hipFuncSetAttribute((const void*)k_full_nz_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                s_attr = true;
            }
            std::vector<int64_t> tptr((size_t)(nb * nb) + 1, 0); /* contacts per tile -> first contact of a tile */
            for (int64_t k = 0; k < Z; k++) tptr[(size_t)((row[k] / FULL_TB) * nb + col[k] / FULL_TB) + 1]++;
            for (size_t i = 1; i < tptr.size(); i++) tptr[i] += tptr[i - 1];
            /* histogram of the counts of every off-diagonal tile (a tile that turns out to hold trans pairs only is summed
             * from it, k_tile_trans); tiles with a count outside 1 .. TILE_HB-1 have none: their items are static, like the
             * diagonal tiles' */
            std::vector<int> tile_hist((size_t)(nb * nb), -1);
            std::vector<unsigned> hist;
            std::vector<TileInfo> tiles;
            {
                std::vector<char> bad((size_t)(nb * nb), 0);
                for (int64_t k = 0; k < Z; k++)
                    if (cnt[k] < 1 || cnt[k] >= TILE_HB) bad[(size_t)((row[k] / FULL_TB) * nb + col[k] / FULL_TB)] = 1;
                for (int64_t bi = 0; bi < nb; bi++)
                    for (int64_t bj = bi + 1; bj < nb; bj++) {
                        const size_t t = (size_t)(bi * nb + bj);
                        if (!bad[t] && tptr[t + 1] > tptr[t] && tptr[t + 1] - tptr[t] < (1LL << 31)) {
                            tile_hist[t] = (int)tiles.size();
                            tiles.push_back(TileInfo{(int)bi, (int)bj, 0, 0});
                        }
                    }
                hist.assign(std::max<size_t>(tiles.size(), 1) * TILE_HB, 0u);
                for (int64_t k = 0; k < Z; k++) {
                    const int h = tile_hist[(size_t)((row[k] / FULL_TB) * nb + col[k] / FULL_TB)];
                    if (h >= 0) hist[(size_t)h * TILE_HB + cnt[k]]++;
                }
            }
            /* work items: the static ones first (k_full_nz_tiled's grid), then those of the tiles with a histogram */
            std::vector<TileWork> work;
            for (int pass = 0; pass < 2; pass++) {
                for (int64_t bi = 0; bi < nb; bi++)
                    for (int64_t bj = bi; bj < nb; bj++) {
                        const int64_t b = tptr[(size_t)(bi * nb + bj)], e = tptr[(size_t)(bi * nb + bj) + 1];
                        const int h = tile_hist[(size_t)(bi * nb + bj)];
                        if ((h >= 0) != (pass == 1)) continue;
                        if (h >= 0) tiles[(size_t)h].first_item = (int)work.size();
                        const size_t first = work.size();
                        for (int64_t o = b; o < e; o += FULL_CHUNK)
                            work.push_back(TileWork{(long long)o, (int)std::min<int64_t>(FULL_CHUNK, e - o), (int)bi, (int)bj, 0});
                        for (size_t q = first; q < work.size(); q++) work[q].pad = (int)(work.size() - 1 - q); /* items of this tile behind this one */
                        if (h >= 0) tiles[(size_t)h].n_items = (int)work.size() - tiles[(size_t)h].first_item;
                    }
                if (pass == 0) c->n_tile_static = (int)work.size();
            }
            std::vector<uint2> tc((size_t)Z);
            std::vector<int64_t> cur(tptr.begin(), tptr.end() - 1);
            for (int64_t k = 0; k < Z; k++) {
                const int64_t t = (row[k] / FULL_TB) * nb + col[k] / FULL_TB;
                tc[(size_t)cur[(size_t)t]++] = make_uint2((unsigned)(row[k] % FULL_TB) | ((unsigned)(col[k] % FULL_TB) << 11), (unsigned)cnt[k]);
            }
            DALLOC(c->tiled_cc, (size_t)Z);
            DALLOC(c->tile_work, work.size());
            DALLOC(c->tile_hist, hist.size());
            DALLOC(c->tile_sig, (size_t)nb * (SIG_WORDS + SIG_FOLD));
            DALLOC(c->tile_info, std::max<size_t>(tiles.size(), 1));
            DALLOC(c->tile_dyn, 4);
            DALLOC(c->tile_partial, 2 * ((tiles.size() + TILE_TRANS_THREADS / 64 - 1) / (TILE_TRANS_THREADS / 64) + 1));
            DALLOC(c->tile_partial0, 2 * ((tiles.size() + TILE_TRANS_THREADS / 64 - 1) / (TILE_TRANS_THREADS / 64) + 1));
            DALLOC(c->tile_dyn_list, std::max<size_t>(work.size() - (size_t)c->n_tile_static, 1));
            HIPCK(hipMemcpy(c->tiled_cc, tc.data(), (size_t)Z * sizeof(uint2), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(c->tile_work, work.data(), work.size() * sizeof(TileWork), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(c->tile_hist, hist.data(), hist.size() * sizeof(unsigned), hipMemcpyHostToDevice));
            if (!tiles.empty()) HIPCK(hipMemcpy(c->tile_info, tiles.data(), tiles.size() * sizeof(TileInfo), hipMemcpyHostToDevice));
            c->n_tile_info = (int)tiles.size();
            c->n_tile_work = (int)work.size();
        }
    }
    c->Z = Z;
    c->M = M;
    c->max_count = max_count;
    c->rank = rank;
    c->world = world;
    c->have_contacts = true;
    return 0;
}

extern "C" int ig_upload_subfrag_table(ig_ctx* c, const float* xyzw, int32_t M)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (M <= 0) return fail("ig_upload_subfrag_table: M <= 0");
    if (c->M && c->M != M) return fail("ig_upload_subfrag_table: M=%d does not match the contacts (%d)", M, c->M);
    std::vector<SubTab> t((size_t)M);
    for (int s = 0; s < M; s++) {
        t[s].parent = (int)xyzw[4 * (size_t)s];
        t[s].wat = xyzw[4 * (size_t)s + 1];
        t[s].cri = xyzw[4 * (size_t)s + 2];
        t[s].w = (int)xyzw[4 * (size_t)s + 3];
        if (s > 0 && (t[s].parent < t[s - 1].parent || (t[s].parent == t[s - 1].parent && t[s].w != t[s - 1].w + 1)))
            return fail("ig_upload_subfrag_table: sub-fragments of a bin must be contiguous and ordered (at %d)", s);
    }
    hipFree(c->sub_tab);
    DALLOC(c->sub_tab, (size_t)M);
    HIPCK(hipMemcpy(c->sub_tab, t.data(), (size_t)M * sizeof(SubTab), hipMemcpyHostToDevice));
    hipFree(c->tab.dist);
    hipFree(c->tab_prev.dist);
    for (Tables* tb : {&c->tab, &c->tab_prev}) {
        int* blk;
        DALLOC(blk, 6 * (size_t)M + 2);
        tb->dist = (float*)blk;
        tb->stot = (float*)(blk + (size_t)M);
        tb->len = blk + 2 * (size_t)M;
        tb->cp = (int2*)(blk + 4 * (size_t)M + ((4 * (size_t)M) & 1)); /* keep the int2 array 8-byte aligned */
    }
    hipFree(c->prev_touched);
    DALLOC(c->prev_touched, (size_t)M);
    hipFree(c->touched_bits);
    DALLOC(c->touched_bits, 2 * ((size_t)(M + 31) / 32 + 1));
    HIPCK(hipMemset(c->touched_bits, 0, 2 * ((size_t)(M + 31) / 32 + 1) * sizeof(unsigned)));
    c->M = M;
    c->have_sub = true;
    return 0;
}

static int launch_recompute(ig_ctx* c);

/* the incremental genome distance of k_commit_batch evaluates every credit a move can change exactly once; its rule for
 * telling who evaluates a fragment reached through several links needs the initial prev / next to be mutually inverse
 * (true of every genome made of contigs; an arbitrary pair of arrays takes the one-move path, which recounts all credits) */
static bool links_inverse(const int32_t* ip, const int32_t* in, size_t n)
{
    for (size_t f = 0; f < n; f++) {
        const int p = ip[f], q = in[f];
        if (p < -1 || q < -1 || p >= (int)n || q >= (int)n) return false;
        if (p >= 0 && (in[p] != (int)f || p == (int)f)) return false;
        if (q >= 0 && (ip[q] != (int)f || q == (int)f)) return false;
        if (p >= 0 && p == q) return false;
    }
    return true;
}

extern "C" int ig_upload_state(ig_ctx* c, const int32_t* soa, int32_t N)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (N <= 0) return fail("ig_upload_state: N <= 0");
    if (!c->have_sub) return fail("ig_upload_state: upload the sub-fragment table first");
    c->nuis_spec = c->spec_valid = false;
    const size_t n = N;
    std::vector<int> host(17 * n);
    /* soa member order (KA:40-58): 0 pos 1 sub_pos 2 id_c 3 start_bp 4 len_bp 5 sub_len 6 circ 7 id 8 prev 9 next
     * 10 l_cont 11 sub_l_cont 12 l_cont_bp 13 ori 14 rep 15 activ 16 id_d */
    static const int dyn_src[NDYN] = {0, 1, 2, 3, 6, 8, 9, 10, 11, 12, 13};
    for (int k = 0; k < NDYN; k++) memcpy(&host[k * n], soa + dyn_src[k] * n, n * sizeof(int));
    memcpy(&host[11 * n], soa + 4 * n, n * sizeof(int));  /* len_bp */
    memcpy(&host[12 * n], soa + 5 * n, n * sizeof(int));  /* sub_len */
    memcpy(&host[14 * n], soa + 14 * n, n * sizeof(int)); /* rep */
    memcpy(&host[15 * n], soa + 15 * n, n * sizeof(int)); /* activ */
    memcpy(&host[16 * n], soa + 16 * n, n * sizeof(int)); /* id_d */
    /* first sub-fragment of each bin + consistency with the table */
    std::vector<SubTab> t((size_t)c->M);
    HIPCK(hipMemcpy(t.data(), c->sub_tab, (size_t)c->M * sizeof(SubTab), hipMemcpyDeviceToHost));
    long long acc = 0;
    for (size_t f = 0; f < n; f++) {
        host[13 * n + f] = (int)acc;
        const int sl = soa[5 * n + f];
        if (sl < 1 || acc + sl > c->M) return fail("ig_upload_state: sub_len of bin %zu inconsistent with the sub-fragment table", f);
        for (int w = 0; w < sl; w++)
            if (t[(size_t)acc + w].parent != (int)f || t[(size_t)acc + w].w != w)
                return fail("ig_upload_state: sub-fragment %lld is not (bin %zu, index %d)", acc + w, f, w);
        acc += sl;
        if (soa[15 * n + f] != 1) return fail("ig_upload_state: inactive fragments are not supported (dead in the reference)");
    }
    if (acc != c->M) return fail("ig_upload_state: bins cover %lld sub-fragments, table has %d", acc, c->M);
    /* internal contig ids: any injective relabelling works; keep the caller's, they are >= 0 */
    int max_c = 0;
    for (size_t f = 0; f < n; f++) {
        if (soa[2 * n + f] < 0) return fail("ig_upload_state: negative contig id");
        max_c = std::max(max_c, soa[2 * n + f]);
    }
    hipFree(c->st_block);
    DALLOC(c->st_block, 17 * n);
    HIPCK(hipMemcpy(c->st_block, host.data(), 17 * n * sizeof(int), hipMemcpyHostToDevice));
    int** sp = (int**)&c->st;
    for (int k = 0; k < 17; k++) sp[k] = c->st_block + k * n;
    const bool size_changed = (c->N != N);
    c->N = N;
    if (size_changed || !c->have_init) { /* default initial genome for the distance = this state (CL:269-276) */
        hipFree(c->init_prev);
        hipFree(c->init_next);
        hipFree(c->orientable);
        hipFree(c->black);
        DALLOC(c->init_prev, n);
        DALLOC(c->init_next, n);
        DALLOC(c->orientable, n);
        DALLOC(c->black, n);
        std::vector<int> orient(n);
        for (size_t f = 0; f < n; f++) orient[f] = soa[5 * n + f] > 1;
        HIPCK(hipMemcpy(c->init_prev, soa + 8 * n, n * sizeof(int), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(c->init_next, soa + 9 * n, n * sizeof(int), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(c->orientable, orient.data(), n * sizeof(int), hipMemcpyHostToDevice));
        HIPCK(hipMemset(c->black, 0, n));
        c->have_init = true;
        c->init_links_inverse = links_inverse(soa + 8 * n, soa + 9 * n, n);
    }
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    hg.N = N;
    hg.M = c->M;
    hg.next_cid = max_c + 1;
    hg.n_tot_pxl = (double)c->M * ((double)c->M - 1.0) / 2.0; /* CL:366 */
    for (int i = 0; i < 12; i++) hg.valid_insert[i] = 0; /* CL:421 */
    hg.error = 0;
    HIPCK(hipMemcpy(c->glob, &hg, sizeof hg, hipMemcpyHostToDevice));
    c->have_state = true;
    if (ensure_move_buffers(c, 8)) return -1;
    return launch_recompute(c);
}

/* tables, head count, genome-distance credits and (when parameters are known) the exact
 * likelihood sums of the current state */
static int launch_recompute(ig_ctx* c)
{
    c->nh_valid = false; /* (the histogram of the screened nuisance pass: rebuilt by the next run) */
    c->nh_pending_slot = -1;
    if (!c->have_state || !c->have_sub) return 0;
    const int N = c->N, M = c->M;
    hipLaunchKernelGGL(k_fill_tables, dim3((M + 255) / 256), dim3(256), 0, c->stream, c->st, c->sub_tab, c->tab, M);
    HIPCK(hipMemcpyAsync(c->tab_prev.dist, c->tab.dist, (6 * (size_t)M + 2) * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    long long* scratch = c->scratch8; /* persistent: an allocation per call costs more than the small kernels */
    HIPCK(hipMemsetAsync(scratch, 0, 8 * sizeof(long long), c->stream));
    int* heads = (int*)(scratch + 6);
    HIPCK(hipMemsetAsync(&c->glob->max_L, 0, 2 * sizeof(int), c->stream)); /* max_L, max_SL: recounted */
    hipLaunchKernelGGL(k_count_heads, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->st, N, heads, c->glob);
    HIPCK(hipMemsetAsync(&c->glob->credit2_acc, 0, sizeof(long long), c->stream));
    hipLaunchKernelGGL(k_post, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->st, c->init_prev, c->init_next, c->orientable,
                       c->black, c->glob, N);
    if (c->have_params && c->have_contacts) {
        launch_full_nz(c, c->tab, 0, scratch, PzTab{c->pz_tab, c->pz_n});
        hipLaunchKernelGGL(k_full_zero, dim3(128), dim3(256), 0, c->stream, c->tab, c->glob, 0, M, scratch + 2);
    }
    long long h[8];
    HIPCK(hipMemcpyAsync(h, scratch, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    ig_acc_normalize((int64_t*)&h[0], (int64_t*)&h[1]);
    ig_acc_normalize((int64_t*)&h[2], (int64_t*)&h[3]);
    hg.nz_hi = h[0];
    hg.nz_lo = h[1];
    hg.z_hi = h[2];
    hg.z_lo = h[3];
    hg.n_intra = h[4];
    hg.n_contigs = ((int*)&h[6])[0];
    c->max_L = hg.max_L;
    c->max_SL = hg.max_SL;
    hg.credit2 = hg.credit2_acc;
    hg.credit2_acc = 0;
    hg.n_prev_touched = 0;
    HIPCK(hipMemcpy(c->glob, &hg, sizeof hg, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int ig_set_params(ig_ctx* c, const float p[8], float mean_subfrag_kb, int which)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (which != 0 && which != 1) return fail("ig_set_params: which must be 0 or 1");
    if (which == 0) c->nuis_spec = c->spec_valid = false; /* moves scored ahead (ig_nuis_step_begin) were scored under the old set */
    HIPCK(hipStreamSynchronize(c->stream));
    ig_params hp = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]};
    if (which == 0) c->par_model = hp;
    HIPCK(hipMemcpy(&c->glob->par[which], &hp, sizeof hp, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&c->glob->mean_kb, &mean_subfrag_kb, sizeof(float), hipMemcpyHostToDevice));
    {
        /* P_z table of this parameter set; length: first rank distance whose s_z reaches d_max (+1), capped */
        float*& tab = which == 0 ? c->pz_tab : c->pz_tab1;
        int& n = which == 0 ? c->pz_n : c->pz_n1;
        if (!tab) DALLOC(tab, PZ_MAX);
        double need = (mean_subfrag_kb > 0) ? (double)p[5] / (double)mean_subfrag_kb + 2.0 : 0.0;
        n = (need > 0 && need < (double)PZ_MAX) ? (int)need : ((need >= (double)PZ_MAX) ? PZ_MAX : 0);
        if (n > 0) hipLaunchKernelGGL(k_build_pz, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->glob, tab, n, which);
        if (which == 0) { /* the constants k_score_list stages */
            if (!c->score_const) DALLOC(c->score_const, 1);
            hipLaunchKernelGGL(k_build_score_const, dim3((LDS_PZ + 2 + 255) / 256), dim3(256), 0, c->stream, c->glob, PzTab{tab, n}, c->lgf_tab,
                               c->score_const, 0);
            if (!c->screen_const) DALLOC(c->screen_const, 1);
            hipLaunchKernelGGL(k_build_screen_const, dim3((LDS_PZ + 2 + 255) / 256), dim3(256), 0, c->stream, c->glob, PzTab{tab, n}, c->screen_const);
        }
    }
    if (which == 0) {
        c->have_params = true;
        return launch_recompute(c); /* the maintained exact sums depend on param_simu */
    }
    return 0;
}

extern "C" int ig_set_insert_config(ig_ctx* c, const int32_t list_bounds[IG_N_INSERT_BLOCKS], int32_t max_bounds_insert)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(c->glob->list_bounds, list_bounds, 6 * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&c->glob->slice_nb, &max_bounds_insert, sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int ig_set_initial_genome(ig_ctx* c, const int32_t* ip, const int32_t* in, const int32_t* orientable,
                                     const int32_t* blacklisted, int32_t nb)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (!c->have_state) return fail("ig_set_initial_genome: upload the state first");
    const size_t n = c->N;
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(c->init_prev, ip, n * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->init_next, in, n * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->orientable, orientable, n * sizeof(int), hipMemcpyHostToDevice));
    std::vector<unsigned char> b(n, 0);
    for (int i = 0; i < nb; i++) {
        if (blacklisted[i] < 0 || (size_t)blacklisted[i] >= n) return fail("ig_set_initial_genome: blacklisted id out of range");
        b[blacklisted[i]] = 1;
    }
    int cnt = 0;
    for (size_t i = 0; i < n; i++) cnt += b[i];
    HIPCK(hipMemcpy(c->black, b.data(), n, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&c->glob->n_black, &cnt, sizeof(int), hipMemcpyHostToDevice));
    c->init_links_inverse = links_inverse(ip, in, n);
    return launch_recompute(c);
}

/* canonical contig numbering of modify_gl_cuda_buffer (CL:2715-2881): contigs enumerated by
 * ascending index of their pos==0 fragment (the order an in-order select_uniq_id_c produces,
 * KA:357-406), stable sort by length descending (CL:69-77), id = (n-1) - rank (KA:4689-4692). */
static void canonical_ids(const int* pos, const int* cid, const int* L, size_t n, std::vector<int>& out, int* n_contigs)
{
    std::vector<int> heads;
    for (size_t f = 0; f < n; f++)
        if (pos[f] == 0) heads.push_back((int)f);
    std::vector<int> order(heads.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return L[heads[a]] > L[heads[b]]; });
    int max_c = 0;
    for (size_t f = 0; f < n; f++) max_c = std::max(max_c, cid[f]);
    std::vector<int> map((size_t)max_c + 1, -1);
    const int nc = (int)heads.size();
    for (int r = 0; r < nc; r++) map[cid[heads[order[r]]]] = (nc - 1) - r;
    out.resize(n);
    for (size_t f = 0; f < n; f++) out[f] = map[cid[f]];
    *n_contigs = nc;
}

extern "C" int ig_download_state(ig_ctx* c, int32_t* soa)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (!c->have_state) return fail("ig_download_state: no state");
    const size_t n = c->N;
    HIPCK(hipStreamSynchronize(c->stream));
    std::vector<int> host(17 * n);
    HIPCK(hipMemcpy(host.data(), c->st_block, 17 * n * sizeof(int), hipMemcpyDeviceToHost));
    static const int dyn_src[NDYN] = {0, 1, 2, 3, 6, 8, 9, 10, 11, 12, 13};
    for (int k = 0; k < NDYN; k++) memcpy(soa + dyn_src[k] * n, &host[k * n], n * sizeof(int));
    memcpy(soa + 4 * n, &host[11 * n], n * sizeof(int));
    memcpy(soa + 5 * n, &host[12 * n], n * sizeof(int));
    memcpy(soa + 14 * n, &host[14 * n], n * sizeof(int));
    memcpy(soa + 15 * n, &host[15 * n], n * sizeof(int));
    memcpy(soa + 16 * n, &host[16 * n], n * sizeof(int));
    for (size_t f = 0; f < n; f++) soa[7 * n + f] = (int)f;
    std::vector<int> ids;
    int nc;
    canonical_ids(&host[0], &host[2 * n], &host[7 * n], n, ids, &nc);
    memcpy(soa + 2 * n, ids.data(), n * sizeof(int));
    return 0;
}

extern "C" int ig_renumber_contigs(ig_ctx* c, int32_t* n_contigs, float* mean_len, int32_t* max_id)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    int nc;
    HIPCK(hipMemcpy(&nc, &c->glob->n_contigs, sizeof(int), hipMemcpyDeviceToHost));
    if (n_contigs) *n_contigs = nc;
    if (mean_len) *mean_len = (float)c->N / (float)nc;
    if (max_id) *max_id = nc - 1;
    return 0;
}

extern "C" int ig_bomb(ig_ctx* c, const int32_t* shuffle)
{
    IG_JOIN(c);
    (void)shuffle; /* explode_genome writes id_c = shuffle[i] (KA:419); the renumbering that follows (CL:1948) erases it */
    HIPCK(hipSetDevice(c->device));
    if (!c->have_state) return fail("ig_bomb: no state");
    c->nuis_spec = c->spec_valid = false;
    hipLaunchKernelGGL(k_explode, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->st, c->N);
    int next = c->N;
    HIPCK(hipMemcpyAsync(&c->glob->next_cid, &next, sizeof(int), hipMemcpyHostToDevice, c->stream));
    return launch_recompute(c);
}

extern "C" int ig_genome_distance(ig_ctx* c, double* d)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    const double norm = 3.0 * (double)(hg.N - hg.n_black);
    *d = (norm - 0.5 * (double)hg.credit2) / norm;
    return 0;
}

extern "C" int ig_get_valid_insert(ig_ctx* c, int32_t out12[12])
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(out12, c->glob->valid_insert, 12 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ig_full_likelihood(ig_ctx* c, int which, int use_prev, double* nz, double* z, int64_t* limbs5)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (!c->have_contacts || !c->have_state || !c->have_params) return fail("ig_full_likelihood: contacts, state and parameters are required");
    if (which != 0 && which != 1) return fail("ig_full_likelihood: which must be 0 or 1");
    long long* scratch = c->scratch8; /* persistent: an allocation per call costs more than the small kernels */
    HIPCK(hipMemsetAsync(scratch, 0, 8 * sizeof(long long), c->stream));
    Tables& t = use_prev ? c->tab_prev : c->tab;
    /* which == 1 before any ig_set_params(.., 1): no table yet, every P_z is evaluated directly */
    const PzTab pz = which == 0 ? PzTab{c->pz_tab, c->pz_n} : PzTab{c->pz_tab1, c->pz_tab1 ? c->pz_n1 : 0};
    if (!launch_full_nz(c, t, which, scratch, pz, nullptr, scratch + 2))
        hipLaunchKernelGGL(k_full_zero, dim3(128), dim3(256), 0, c->stream, t, c->glob, which, c->M, scratch + 2);
    long long h[8];
    HIPCK(hipMemcpyAsync(h, scratch, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    ig_acc_normalize((int64_t*)&h[0], (int64_t*)&h[1]);
    ig_acc_normalize((int64_t*)&h[2], (int64_t*)&h[3]);
    if (nz) *nz = ig_acc_to_double(h[0], h[1]);
    if (z) { /* CL:755-759 with the float log_e of the kernels replaced by the host's double constant */
        const double log_e = 0.43429448190325182;
        double n_tot_pxl;
        float v_inter;
        const int vi_bits = (int)h[7];
        memcpy(&n_tot_pxl, &h[5], sizeof n_tot_pxl); /* k_full_zero: n_tot_pxl and v_inter of the parameter set */
        memcpy(&v_inter, &vi_bits, sizeof v_inter);
        const double val_intra = ig_acc_to_double(h[2], h[3]) * log_e;
        const double val_inter = log_e * (n_tot_pxl - (double)h[4]) * -1.0 * (double)v_inter;
        *z = val_intra + val_inter;
    }
    if (limbs5)
        for (int i = 0; i < 5; i++) limbs5[i] = h[i];
    return 0;
}

/* ------------------------------------------------------------------ move driver */

static int ensure_io(ig_ctx* c, int n_moves, int max_c)
{
    /* pinned staging of the lists and the result records: copies from / to the caller's pageable arrays block the host for
     * 10-20 us each, which is a tenth of a call of one batch */
    const size_t need = (size_t)n_moves * (sizeof(int) + (size_t)IG_MAX_CANDIDATES * sizeof(int) + sizeof(ig_move_result));
    if (c->h_stage_bytes < need) {
        if (c->h_stage) hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_bytes = 0;
        const size_t want = std::max(need, (size_t)1 << 16);
        if (hipHostMalloc((void**)&c->h_stage, want, hipHostMallocDefault) == hipSuccess) c->h_stage_bytes = want;
        else (void)hipGetLastError(); /* no pinned memory: the copies go through the caller's arrays */
    }
    if (c->results_cap < n_moves) {
        hipFree(c->d_results);
        hipFree(c->d_frags);
        DALLOC(c->d_results, (size_t)n_moves);
        DALLOC(c->d_frags, (size_t)n_moves);
        c->results_cap = n_moves;
    }
    if (c->cands_cap < n_moves * max_c) {
        hipFree(c->d_cands);
        DALLOC(c->d_cands, (size_t)n_moves * max_c);
        c->cands_cap = n_moves * max_c;
    }
    return 0;
}

static int check_ready(ig_ctx* c)
{
    if (!c->have_contacts || !c->have_sub || !c->have_state || !c->have_params)
        return fail("contacts, sub-fragment table, state and parameters must be uploaded before a move");
    flush_pending_sums(c);
    c->nuis_caught_up = false;
    c->main_drained = false;
    c->nuis_spec = c->spec_valid = false; /* every entry point that runs moves passes here: a run of ig_nuis_step_begin ends with it */
    return 0;
}

static int g_tail_quirk = 1;

/* two-tier scoring: the smallest grid of the exact kernel a batch may be launched with.  k_contend cuts a slot's work into items
 * so that the slot alone needs at most half of the grid; whatever the item size there is up to one partly filled item per
 * (candidate, column, list segment): max_c x NSLOT x SLICE_SEG of them (2 000 at 5 candidates, 6 400 at 16) */
static int exact_grid_floor(const ig_ctx* c, int max_c)
{
    return std::min(c->mb.work_cap, std::max(4096, 2 * std::max(max_c, 1) * NSLOT * SLICE_SEG + 2048));
}

/* one-move calls: the bound on the longest contig comes back with the result (the window strides follow it) */
static int queue_max_readback(ig_ctx* c)
{
    if (!c->host_max) {
        HIPCK(hipHostMalloc((void**)&c->host_max, 2 * sizeof(int), hipHostMallocDefault));
        c->host_max[0] = c->host_max[1] = 0;
    }
    HIPCK(hipMemcpyAsync(c->host_max, &c->glob->max_L, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    return 0;
}
static void take_max_readback(ig_ctx* c)
{
    if (!c->host_max) return;
    c->max_L = std::max(c->max_L, c->host_max[0]);
    c->max_SL = std::max(c->max_SL, c->host_max[1]);
}

/* enqueue the scoring launches of W move slots (moves move0 .. move0+W-1 of the uploaded lists);
 * phase 0 = up to k_score_list (the sums that are all-reduced when sharded), 1 = the rest, 2 = both */
/* k_rescore_prepare: slots whose structural half (windows, candidate genomes, columns, slice lists) stands but whose
 * parameter-dependent half has to be (re)done -- the parameters changed (an accepted nuisance step), or the slots were only
 * gathered / mutated / sliced so far: accumulators back to zero, the zero-pixel sums of every column under the model's
 * current P_z table (what k_mutate left, KA:3919-4002, from the columns it wrote) */
__global__ void __launch_bounds__(256) k_rescore_prepare(Glob* g, MoveBuf mb, PzTab pz, int w_begin)
{
    const int slot = blockIdx.x, c = blockIdx.y, w = w_begin + blockIdx.z, t = threadIdx.x;
    MoveCtl& mc = mb.ctl[w];
    if (c >= mc.C) return;
    const int cw = CW(w, c);
    if (slot == 0) { /* once per candidate */
        for (int i = t; i < NSLOT * 2; i += blockDim.x) mb.part[(size_t)cw * P_STRIDE + P_NZ + i] = 0;
        for (int i = t; i < NSLOT * 2; i += blockDim.x) {
            mb.qpart[(size_t)cw * Q_STRIDE + Q_NZFULL + i] = 0;
            mb.qpart[(size_t)cw * Q_STRIDE + Q_TAIL + i] = 0;
            ((long long*)mb.scr)[(size_t)cw * NSLOT * 2 + i] = 0;
        }
        if (t == 0) {
            mb.scr_void[cw] = 0;
            mb.scr_ub[cw] = 0;
            mb.cont[cw] = 0xffffffffu;
            mb.ident[cw] = 0;
            if (c == 0) {
                mc.exact_chunk = 0;
                if (mc.overflow == 2) mc.overflow = 0; /* the exact kernel's grid is dealt out again (the slice pool's verdict, 1, stands) */
                mc.pred = -1;
                mc.pred_pad = -1;
                mc.pd_hi = mc.pd_lo = 0;
                if (blockIdx.z == 0 && mb.work)
                    for (int q = 0; q < 16; q++) mb.work[q] = 0;
            }
        }
    }
    const CandMeta& m = mb.meta[cw];
    const int k = m.kidx[slot];
    if (k < 0 || m.m_loc > mb.sM) return;
    const ig_params p = g->par[0];
    const float mean = g->mean_kb;
    const uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * mb.sM;
    const ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
    long long hi = 0, lo = 0;
    for (int ls = t; ls < m.m_loc; ls += blockDim.x) {
        const uint2 v = col[ls];
        const int npos = (int)(v.y & 0x0fffffffu), code = (int)(v.y >> 28);
        if (npos > 0) {
            const long long q2 = zero_q(p, npos, cm[code].len, cm[code].stot, mean, pz.v, pz.n);
            hi += q2 >> 32;
            lo += (long long)(unsigned int)q2;
        }
    }
    __shared__ long long red[2][4];
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    if ((t & 63) == 0) {
        red[0][t >> 6] = hi;
        red[1][t >> 6] = lo;
    }
    __syncthreads();
    if (t == 0) {
        long long* q = mb.qpart + (size_t)cw * Q_STRIDE;
        q[Q_Z + 2 * k] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        q[Q_Z + 2 * k + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

/* par_begin / par_end: the slots whose PARAMETER-DEPENDENT half (tail walk, screening, contenders, exact kernel, records) is
 * done by this call (default: all of [w_begin, w_end)); par_only: nothing but that half, on slots gathered, mutated and
 * sliced by an earlier call (the runs of (move, nuisance step) pairs: the structural half of a batch survives an accepted
 * step, and is scored in pieces that follow the run lengths) */
static void enqueue_score(ig_ctx* c, int move0, int W, int max_c, int force_slot, int phase, int w_begin = 0, int w_end = -1, int par_begin = -1,
                          int par_end = -1, bool par_only = false)
{
    /* slots [w_begin, w_end) are sliced and scored here (multi-GPU: the other ranks score the rest and the slot-major
     * records are all-gathered); the candidate genomes of EVERY slot are built on every rank, the commit step needs them */
    if (w_end < 0) w_end = W;
    const int nW = w_end - w_begin;
    const int pb = par_begin < 0 ? w_begin : par_begin, pe = par_end < 0 ? w_end : par_end;
    const int nWp = pe - pb;
    struct StaleGuard { /* the kernels launched below see the list of contigs modified since these slots were gathered */
        ig_ctx* c;
        StaleGuard(ig_ctx* ctx, bool on) : c(ctx) { c->mb.stale = on ? c->dirty_buf : nullptr; }
        ~StaleGuard() { c->mb.stale = nullptr; }
    } stale_guard(c, par_only && pb > 0);
    const int N = c->N;
    const int gN = std::max((N + 255) / 256, W);
    const PzTab pz{c->pz_tab, c->pz_n};
    if (phase == 0 || phase == 2) {
        const int n_tw = (c->M + 31) / 32 + 1;
        if (!par_only) {
            c->touched_flip ^= 1;
            c->mb.touched = c->touched_bits + (size_t)c->touched_flip * n_tw;
        }
        if (!par_only) {
            TimedLaunch t(c, T_GATHER);
            hipLaunchKernelGGL(k_gather, dim3(gN), dim3(256), 0, c->stream, c->st, c->glob, c->mb, c->d_cands, c->d_frags, move0, W, max_c,
                               c->tab, c->tab_prev, c->prev_touched, force_slot, c->touched_bits + (size_t)(c->touched_flip ^ 1) * n_tw, n_tw);
        }
        if (!par_only) {
            TimedLaunch t(c, T_MUTATE);
            /* slots split over GPUs: only the own slots' candidate genomes are built here; k_mutate_winners rebuilds what the
             * commit step applies from the other ranks' slots */
            c->own_begin = w_begin;
            c->own_end = w_end;
            if (nW > 0) {
                /* long contigs (late in an assembly: windows of thousands of sub-fragments): more threads per candidate genome */
                const int mean_len = c->n_contigs_seen > 0 ? c->N / c->n_contigs_seen : 0;
                const int mutate_threads = mean_len >= 600 ? 1024 : (mean_len >= 150 ? 512 : 256);
                hipLaunchKernelGGL(k_mutate, dim3(NSLOT, max_c, nW), dim3(mutate_threads), 0, c->stream, c->st, c->tab, c->sub_tab, c->rowptr,
                                   c->glob, c->mb, pz, w_begin);
            }
        }
        if (force_slot < 0 && nW > 0) {
            if (!par_only) {
                TimedLaunch t(c, T_SLICE);
                hipLaunchKernelGGL(k_offsets, dim3(1), dim3(OFFSETS_THREADS), 0, c->stream, c->mb, W, w_begin, w_end, max_c);
                /* a wave walks a row (a workgroup 4 rows at a time).  Measured at cfg3 (us per launch of 24 slots): 32 workgroups per
                 * candidate 193, 64: 156, 96: 148, 128: 136, 256 with the chunks of a row dealt to several waves: 146 */
                /* ... and with the lists in 8 segments (round 3), workgroups per plane -> moves/s: cfg3 24: 43.1 k, 32: 44.2, 40: 45.5, 48: 45.2 - 45.7,
                 * 64: 45.4, 80: 45.3, 96: 44.6, 128: 43.9; cfg2 32: 57.3 k, 48: 55.9, 96: 55.5; cfg5 32: 28.4 k, 48: 29.9, 96: 30.1; bigctg 32: 5.3 k,
                 * 48: 5.75, 96: 5.93 -- the count follows the contacts a plane holds (two contigs' rows: 2 Z / contigs of the last batch) */
                const int rb_auto = c->n_contigs_seen > 0
                                        ? std::min(SLICE_RB, std::max(32, (int)(2.0 * (double)c->Z / (double)c->n_contigs_seen / 2100.0)))
                                        : SLICE_RB;
                const int rb = rb_auto;
                const int s_share = 1, s_maxj = 1 << 20; /* A's rows once per move */
                if (c->mb.packed)
                    hipLaunchKernelGGL(k_slice<true>, dim3(rb, max_c + 1, nW), dim3(256), 0, c->stream, c->rowptr, c->cc, c->tab, c->glob, c->mb, c->rank,
                                       c->world, w_begin, s_share, s_maxj);
                else
                    hipLaunchKernelGGL(k_slice<false>, dim3(rb, max_c + 1, nW), dim3(256), 0, c->stream, c->rowptr, c->cc, c->tab, c->glob, c->mb, c->rank,
                                       c->world, w_begin, s_share, s_maxj);
            }
            if (par_only && nWp > 0)
                hipLaunchKernelGGL(k_rescore_prepare, dim3(NSLOT, max_c, nWp), dim3(256), 0, c->stream, c->glob, c->mb, pz, pb);
            if (nWp > 0) {
            /* two-tier scoring (batches on one handle, packed lists): every column through the float screening kernel, the exact
             * kernel only for the columns that can still win (ig_kernels_screen.cuh).  IG_SCREEN=0: everything exact;
             * IG_SCREEN_VERIFY=1: everything exact AND screened, the bound checked column by column. */
            const int s_screen = getenv("IG_SCREEN") ? atoi(getenv("IG_SCREEN")) : 1; /* read per launch, like the next one */
            const int verify = getenv("IG_SCREEN_VERIFY") ? atoi(getenv("IG_SCREEN_VERIFY")) : 0; /* read per launch: a test toggles it */
            /* exact_next: the decide step met a score of exactly 0.0 among the contenders (stop code 3) -- this scoring is exact in every column */
            const bool exact_once = c->exact_next && !par_only;
            if (exact_once) c->exact_next = false;
            const bool screen = (s_screen || verify) && phase == 2 && W > 1 && c->world == 1 && c->mb.packed && !exact_once;
            /* the Q5 tail walk only needs the slice: in the screening kernel's launch (k_screen_tail), else on a second stream
             * next to k_score_list */
            c->tail_fused = screen;
            if (phase == 2 && !c->tail_fused) {
                hipEventRecord(c->ev_slice, c->stream);
                hipStreamWaitEvent(c->stream2, c->ev_slice, 0);
                hipLaunchKernelGGL(k_tail, dim3(max_c, nWp), dim3(256), 0, c->stream2, c->rowptr, c->cc, c->tab, c->glob, c->mb, c->lgf_tab,
                                   g_tail_quirk, pz, pb);
                hipEventRecord(c->ev_tail, c->stream2);
            }
            int contenders_only = 0;
            if (screen) {
                const int ny = (NSLOT + 1) / 2;
                /* the long lists first (MoveBuf.order, k_offsets): when this launch screens the slots that one placed */
                const int use_order = (!par_only && pb == w_begin && pe == w_end && nWp * max_c <= (int)OFFSETS_THREADS) ? 1 : 0;
                /* narrow batches (late in an assembly: few long contigs, a conflict at nearly every move): few (segment, pair,
                 * candidate) triples, each with a long list -- Q workgroups split a segment so that the launch has ~7 000 of them */
                const int Q = std::max(1, std::min(8, (7000 + SLICE_SEG * ny * max_c * nWp - 1) / (SLICE_SEG * ny * max_c * nWp)));
                {
                    TimedLaunch t(c, T_SCREEN);
                    if (c->tail_fused)
                        hipLaunchKernelGGL(k_screen_tail, dim3(max_c * nWp + SLICE_SEG * Q * ny * max_c * nWp), dim3(SCORE_THREADS), 0, c->stream,
                                           c->screen_const, c->mb, c->mb.scr, c->mb.scr_void, c->mb.scr_ub, max_c, pb, max_c * nWp, c->rowptr, c->cc,
                                           c->tab, c->glob, c->lgf_tab, g_tail_quirk, pz, use_order, Q);
                    else
                        hipLaunchKernelGGL(k_screen<0>, dim3(SLICE_SEG * Q, ny, max_c * nWp), dim3(SCORE_THREADS), 0, c->stream, c->screen_const,
                                           c->mb, c->mb.scr, c->mb.scr_void, c->mb.scr_ub, max_c, pb, use_order, Q);
                }
                /* IG_SCREEN_PROBE=mask (tools/screen_probe.py): the screening kernel WITHOUT the parts in the mask, launched behind the real
                 * one on the same lists and columns, sums into scratch words: timed as "probe" (ig_kernel_time_ms), results untouched */
                static const int s_probe = getenv("IG_SCREEN_PROBE") ? atoi(getenv("IG_SCREEN_PROBE")) : -1;
                if (s_probe >= 0) {
                    const size_t C = (size_t)c->mb.capC * c->mb.capW;
                    if (!c->probe_scr) {
                        if (dalloc(&c->probe_scr, C * NSLOT) || dalloc(&c->probe_void, 2 * C)) return;
                    }
                    hipMemsetAsync(c->probe_scr, 0, C * NSLOT * sizeof(ScreenSum), c->stream);
                    hipMemsetAsync(c->probe_void, 0, 2 * C * sizeof(unsigned), c->stream);
                    TimedLaunch tp(c, T_PROBE);
                    const dim3 grid(SLICE_SEG * Q, ny, max_c * nWp);
#define IG_PROBE(A)                                                                                                                                   \
    hipLaunchKernelGGL(k_screen<A>, grid, dim3(SCORE_THREADS), 0, c->stream, c->screen_const, c->mb, c->probe_scr, c->probe_void, c->probe_void + C, \
                       max_c, pb, use_order, Q)
                    switch (s_probe) {
                    case 1: IG_PROBE(1); break;
                    case 2: IG_PROBE(2); break;
                    case 3: IG_PROBE(3); break;
                    case 4: IG_PROBE(4); break;
                    case 5: IG_PROBE(5); break;
                    case 6: IG_PROBE(6); break;
                    case 7: IG_PROBE(7); break;
                    case 8: IG_PROBE(8); break;
                    case 16: IG_PROBE(16); break;
                    case 32: IG_PROBE(32); break;
                    case 512: IG_PROBE(512); break;
                    case 1024: IG_PROBE(1024); break;
                    case 2048: IG_PROBE(2048); break;
                    case 1032: IG_PROBE(1032); break; /* 1024 + 8 */
                    case 1544: IG_PROBE(1544); break; /* 1024 + 512 + 8 */
                    case 8192: IG_PROBE(8192); break;
                    case 8199: IG_PROBE(8199); break; /* 8192 + 7 */
                    case 4096: IG_PROBE(4096); break;
                    case 1056: IG_PROBE(1056); break; /* 1024 + 32 */
                    case 96: IG_PROBE(96); break;   /* 32 + 64 */
                    case 160: IG_PROBE(160); break; /* 32 + 128 */
                    case 288: IG_PROBE(288); break; /* 32 + 256 */
                    case 480: IG_PROBE(480); break; /* 32 + 64 + 128 + 256 */
                    default: IG_PROBE(0); break;
                    }
#undef IG_PROBE
                }
                if (c->exact_grid <= 0) c->exact_grid = 32768;
                c->exact_grid = std::max(std::min(c->exact_grid, c->mb.work_cap), exact_grid_floor(c, max_c));
                if (!c->tail_fused) hipStreamWaitEvent(c->stream, c->ev_tail, 0); /* the contender test reads the exact tail sums (quirk Q5) */
                flush_pending_sums(c); /* ... and the maintained sum (its intervals are placed with it) */
                hipLaunchKernelGGL(k_contend, dim3(nWp), dim3(256), 0, c->stream, c->glob, c->mb, c->mb.scr, c->mb.scr_void, c->mb.scr_ub, c->mb.cont, pb,
                                   0, c->exact_grid, EXACT_CHUNK);
                hipLaunchKernelGGL(k_worklist, dim3(nWp), dim3(256), 0, c->stream, c->mb, c->mb.cont, pb, c->exact_grid);
                contenders_only = verify ? 0 : 1;
            }
            c->own_screened = screen ? (verify ? 2 : 1) : 0;
            TimedLaunch t(c, T_SCORE);
            const int s_eb = SLICE_SEG; /* one workgroup per (segment, column, candidate) */
            const int s_abl = getenv("IG_ABLATE") ? atoi(getenv("IG_ABLATE")) : 0; /* read per launch: a test toggles it */
            if (contenders_only) {
                /* the work list k_contend left; the grid follows what the previous batches needed (commit_loop), the workgroups
                 * past the end of the list leave on their first load, a slot whose items do not fit is re-run */
                hipLaunchKernelGGL(k_score_list<LDS_COL_SMALL>, dim3(c->exact_grid), dim3(SCORE_THREADS), 0, c->stream, c->score_const, c->mb,
                                   c->lgf_tab, pz, s_abl, max_c, pb, 1);
            } else {
                hipLaunchKernelGGL(k_score_list<LDS_COL_SMALL>, dim3(s_eb, NSLOT, max_c * nWp), dim3(SCORE_THREADS), 0, c->stream, c->score_const,
                                   c->mb, c->lgf_tab, pz, s_abl, max_c, pb, 0);
            }
            if (screen && verify) {
                if (!c->screen_worst) {
                    if (dalloc(&c->screen_worst, 2) == 0) hipMemsetAsync(c->screen_worst, 0, 2 * sizeof(double), c->stream);
                }
                hipLaunchKernelGGL(k_screen_verify, dim3(nWp), dim3(256), 0, c->stream, c->glob, c->mb, c->mb.scr, c->mb.scr_void, c->mb.scr_ub, pb,
                                   c->screen_worst);
            }
            }
        }
    }
    if (phase == 1 || phase == 2) {
        flush_pending_sums(c); /* (k_predict reads the maintained sum: with or without the screening tier in front of it) */
        if (force_slot < 0 && nWp > 0) {
            if (phase == 1) /* after the all-reduce of the list lengths (contact shards): no overlap */
                hipLaunchKernelGGL(k_tail, dim3(max_c, nWp), dim3(256), 0, c->stream, c->rowptr, c->cc, c->tab, c->glob, c->mb, c->lgf_tab,
                                   g_tail_quirk, pz, pb);
            else if (!c->tail_fused)
                hipStreamWaitEvent(c->stream, c->ev_tail, 0);
            TimedLaunch t(c, T_FINALIZE);
            hipLaunchKernelGGL(k_records, dim3(max_c, nWp), dim3(64), 0, c->stream, c->mb, pb, c->own_screened ? 1 : 0);
            /* batches: predicted windowed winners get their exact delta now (not those decided one move per call, ig_nuis_step_begin:
             * a pause costs them nothing they would not wait for anyway) */
            if (phase == 2 && W > 1 && c->world == 1 && !c->no_predict) {
                hipLaunchKernelGGL(k_predict, dim3(nWp), dim3(256), 0, c->stream, c->glob, c->mb, pb, 0);
                hipLaunchKernelGGL(k_predict, dim3(nWp), dim3(256), 0, c->stream, c->glob, c->mb, pb, 1);
                hipLaunchKernelGGL(k_delta, dim3(DELTA_RB, 2, nWp), dim3(SCORE_THREADS), 0, c->stream, c->rowptr, c->cc, c->tab, c->tab_prev,
                                   c->prev_touched, c->glob, c->mb, c->lgf_tab, pz, pb, 1, 0);
            }
        }
    }
}

/* scores + argmax of slot w (or the forced choice of ig_apply) */
static void enqueue_choose(ig_ctx* c, int w, int force_slot)
{
    TimedLaunch t(c, T_ARGMAX);
    if (force_slot < 0) hipLaunchKernelGGL(k_scores, dim3(1), dim3(256), 0, c->stream, c->glob, c->mb, w);
    else hipLaunchKernelGGL(k_force_choice, dim3(1), dim3(1), 0, c->stream, c->glob, c->mb, force_slot);
}

/* one-move tail: exact delta, apply, genome distance, result record */
/* a move is about to be applied outside a run of (move, nuisance step) pairs: the histogram of the screened pass's first tier
 * (NuisHist) does not follow it */
static inline void nh_untracked_move(ig_ctx* c)
{
    if (!c->nh_tracking) {
        c->nh_valid = false;
        c->nh_pending_slot = -1;
    }
}

static void enqueue_apply(ig_ctx* c, int move, int w, int forced, bool log_dirty = false)
{
    nh_untracked_move(c);
    const int N = c->N;
    const PzTab pz{c->pz_tab, c->pz_n};
    {
        TimedLaunch t(c, T_DELTA);
        hipLaunchKernelGGL(k_delta, dim3(DELTA_RB, 2, 1), dim3(SCORE_THREADS), 0, c->stream, c->rowptr, c->cc, c->tab, c->tab_prev,
                           c->prev_touched, c->glob, c->mb, c->lgf_tab, pz, w, 0, 0);
    }
    {
        TimedLaunch t(c, T_APPLY);
        hipLaunchKernelGGL(k_apply, dim3(64), dim3(256), 0, c->stream, c->st, c->tab, c->glob, c->mb, w, forced, c->prev_touched);
    }
    {
        TimedLaunch t(c, T_POST);
        hipLaunchKernelGGL(k_post, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->st, c->init_prev, c->init_next, c->orientable,
                           c->black, c->glob, N);
    }
    {
        TimedLaunch t(c, T_COMMIT);
        hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, c->stream, c->glob, c->mb, c->d_results, move, w, log_dirty ? c->dirty_buf : (int*)nullptr);
    }
}

static void enqueue_move(ig_ctx* c, int move, int max_c, int force_slot, int phase)
{
    enqueue_score(c, move, 1, max_c, force_slot, phase);
    if (phase == 1 || phase == 2) enqueue_choose(c, 0, force_slot);
}

static int validate_move(ig_ctx* c, int frag_a, const int32_t* cands, int C)
{
    if (C < 1 || C > IG_MAX_CANDIDATES) return fail("a move needs 1..%d candidates (got %d)", IG_MAX_CANDIDATES, C);
    if (frag_a < 0 || frag_a >= c->N) return fail("fragment %d out of range", frag_a);
    for (int i = 0; i < C; i++) {
        if (cands[i] < 0 || cands[i] >= c->N) return fail("candidate %d out of range", cands[i]);
        if (cands[i] == frag_a)
            return fail("candidate == focal fragment (%d): the reference reads stale buffers here (quirk Q13), not supported", frag_a);
    }
    return 0;
}

static int g_zero_inject = 0; /* ig_debug_set_zero_inject: the decide step treats every n-th move of a two-tier batch as one with a score of exactly 0.0 */

/* commit the scored batch [move0, move0 + w_now): k_commit_batch, the one-move tail for a windowed winner, resume */
/* the decide + apply launches of the slots [next, w_now) of the batch at move `done` */
static void launch_commit(ig_ctx* c, int done, int w_now, int next, int resumed_plain, bool publish = false)
{
    nh_untracked_move(c);
    flush_pending_sums(c);
    TimedLaunch t(c, T_COMMIT);
    static const int s_fused = getenv("IG_FUSED_COMMIT") ? atoi(getenv("IG_FUSED_COMMIT")) : 1;
    /* one launch where that pays (k_decide_commit: seven waves work behind the decide wave): plain batches of moves on windows that
     * 448 threads apply as fast as k_commit_batch's 1 024 -- not a run's one-move launches (their host waits for the record, which the
     * fused kernel writes behind a longer prologue: 10.2 k instead of 10.7 k), not the late shapes (bigctg: 5.7 k instead of 5.9 k),
     * not a rank's share of a batch (the winners of the other ranks are mutated in between).  IG_FUSED_COMMIT=0 / 2: never / always. */
    const bool fuse = s_fused >= 2 || (s_fused == 1 && !publish && c->max_SL <= 4096);
    /* contender-only records (two-tier scoring): the decide step checks for scores of exactly 0.0 (decide_body, stop code 3) */
    const int zcheck = (c->own_screened == 1 ? 1 : 0) | (std::max(g_zero_inject, 0) << 8);
    if (fuse && !(c->own_begin > 0 || c->own_end < w_now)) {
        const int seq = ++c->bo_seq;
        hipLaunchKernelGGL(k_decide_commit, dim3(1), dim3(64 + FUSED_CW * 64), 0, c->stream, c->glob, c->mb, c->d_results, done, w_now, next, c->dirty_buf,
                           c->batch_out, (volatile int*)c->host_bo_dev, seq, resumed_plain, c->st, c->tab, c->tab_prev, c->init_prev, c->init_next,
                           c->orientable, c->black, c->own_tag, c->own_idx, c->prev_touched, publish ? c->host_nuis_dev : nullptr,
                           publish ? ++c->res_seq : 0, zcheck);
        return;
    }
    hipLaunchKernelGGL(k_decide_batch, dim3(1), dim3(64), 0, c->stream, c->glob, c->mb, c->d_results, done, w_now, next, c->dirty_buf,
                       c->batch_out, (volatile int*)c->host_bo_dev, ++c->bo_seq, resumed_plain, zcheck);
    if (c->own_begin > 0 || c->own_end < w_now)
        hipLaunchKernelGGL(k_mutate_winners, dim3(2, w_now), dim3(256), 0, c->stream, c->st, c->tab, c->sub_tab, c->rowptr, c->glob,
                           c->mb, PzTab{c->pz_tab, c->pz_n}, next, c->own_begin, c->own_end, c->batch_out);
    hipLaunchKernelGGL(k_commit_batch, dim3(1), dim3(COMMIT_THREADS), 0, c->stream, c->st, c->tab, c->tab_prev, c->glob, c->mb,
                       c->init_prev, c->init_next, c->orientable, c->black, c->own_tag, c->own_idx, c->prev_touched, c->d_results, done,
                       w_now, next, c->batch_out, publish ? c->host_nuis_dev : nullptr, publish ? ++c->res_seq : 0);
}

/* what the last k_decide_batch launch reported (batch_out[0..12)), as soon as it is there: the kernel writes a copy to mapped,
 * coherent host memory; the host spins on it, every so often makes sure the stream is still alive (a fault must not hang
 * the host) and falls back to the device copy when the stream has drained without the flag */
static int wait_commit(ig_ctx* c, int bo[12], bool first_of_batch, int next)
{
    if (c->host_bo) {
        volatile int* hb = c->host_bo;
        bool got = false;
        for (unsigned spin = 0; !got; spin++) {
            if (hb[7] == c->bo_seq) {
                got = true;
            } else if ((spin & 0xfff) == 0xfff) {
                const hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return fail("batch commit failed: %s", hipGetErrorString(q));
            }
        }
        if (got) {
            std::atomic_thread_fence(std::memory_order_acquire);
            for (int i = 0; i < 12; i++) bo[i] = hb[i];
        } else {
            HIPCK(hipMemcpy(bo, c->batch_out, 12 * sizeof(int), hipMemcpyDeviceToHost));
        }
    } else {
        HIPCK(hipMemcpyAsync(bo, c->batch_out, 12 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
    }
    if (first_of_batch) c->n_batches++;
    c->n_batch_committed += bo[0] - next;
    c->n_batch_predicted += bo[4];
    c->n_contigs_seen = bo[5];
    c->max_L = std::max(c->max_L, bo[8]);
    c->max_SL = std::max(c->max_SL, bo[9]);
    c->last_stop = bo[10];
    if (bo[10] == 3) { /* a scored column came out as exactly 0.0 under the live scalars: the next scoring is exact in every column */
        c->exact_next = true;
        c->n_zero_fallbacks++;
    }
    return 0;
}

static int commit_loop(ig_ctx* c, int done, int w_now, int* next_out)
{
    int next = 0; /* slots [0, next) of this batch are committed */
    for (;;) {
        launch_commit(c, done, w_now, next, 0);
        int bo[12];
        if (wait_commit(c, bo, next == 0, next)) return -1;
        if (bo[2] && next == 0 && bo[0] == 0 && bo[1] < 0) { /* the batch's first slot did not fit the slice pool (1) or the exact
                                                               * kernel's grid (2): the caller enlarges it and repeats the batch; (3) a
                                                               * score of exactly 0.0: the caller repeats it, every column exact */
            *next_out = -bo[2];
            return 0;
        }
        if (c->own_screened == 1 && next == 0) /* first commit of this batch: size the exact kernel's next grid */
            c->exact_grid = std::min(c->mb.work_cap, std::max(exact_grid_floor(c, c->up_max_c), (int)(1.25 * bo[6]) + 2048));
        next = bo[0];
        if (bo[1] >= 0) { /* slot bo[1] chose a windowed winner: delta + apply with the one-move kernels, then go on */
            enqueue_apply(c, done + bo[1], bo[1], 0);
            c->n_batch_pending++;
            next = bo[1] + 1;
            if (next < w_now) continue;
        }
        break;
    }
    if (next == 0) {
        Glob hg;
        HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
        return fail("device-side consistency failure %d in a batch at move %d", hg.error, done);
    }
    *next_out = next;
    return 0;
}

static int g_batch_w = -1; /* moves scored per launch in ig_step_batch: env IG_BATCH_W, default 24; 1 = one move at a time */

/* the per-slot window buffers (strides: three times the longest contig, at most the genome): keep them under ~64 GB */
static int max_batch_width(ig_ctx* c, int max_c)
{
    const double sN = std::min<double>(c->N, std::max(256.0, 3.0 * c->max_L)), sM = std::min<double>(c->M, std::max(768.0, 3.0 * c->max_SL));
    const double per_slot = (double)std::max(8, max_c) * ((double)NSLOT * NDYN * sN * 4.0 + sM * NSLOT * 8.0 + 3.0 * sN * 4.0 + 2.0 * sM * 4.0);
    const int fit = (int)std::max(1.0, 64e9 / std::max(per_slot, 1.0));
    return std::min(fit, IG_MAX_BATCH);
}

static int batch_width(ig_ctx* c, int max_c)
{
    if (g_batch_w < 0) {
        const char* e = getenv("IG_BATCH_W");
        g_batch_w = e ? atoi(e) : 24;
        g_batch_w = std::min(std::max(g_batch_w, 1), IG_MAX_BATCH);
    }
    return std::min(g_batch_w, max_batch_width(c, max_c));
}

extern "C" int ig_batch_max_width(ig_ctx* c, int32_t max_c) { IG_JOIN(c); return max_batch_width(c, max_c); }

/* validate and upload the pre-drawn (fragment, candidates) lists of a run of moves */
static int upload_moves(ig_ctx* c, int n_moves, const int32_t* frags, const int32_t* cands, int max_c)
{
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("max_c out of range");
    for (int i = 0; i < n_moves; i++) {
        int C = 0;
        while (C < max_c && cands[(size_t)i * max_c + C] >= 0) C++;
        for (int k = C; k < max_c; k++)
            if (cands[(size_t)i * max_c + k] >= 0) return fail("candidates must be packed before the -1 padding");
        if (validate_move(c, frags[i], cands + (size_t)i * max_c, C)) return -1;
    }
    if (ensure_io(c, n_moves, max_c)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, frags, (size_t)n_moves * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)n_moves * max_c * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->up_moves = n_moves;
    c->up_max_c = max_c;
    return 0;
}

static int download_results(ig_ctx* c, int n_moves, ig_move_result* results)
{
    const size_t bytes = (size_t)n_moves * sizeof(ig_move_result);
    ig_move_result* stage = (c->h_stage && c->h_stage_bytes >= bytes) ? (ig_move_result*)c->h_stage : nullptr;
    HIPCK(hipMemcpyAsync(stage ? stage : results, c->d_results, bytes, hipMemcpyDeviceToHost, c->stream));
    if (queue_max_readback(c)) return -1;
    HIPCK(hipStreamSynchronize(c->stream));
    if (stage) memcpy(results, stage, bytes);
    HIPCK(hipGetLastError());
    {
        size_t pending = 0; /* timer events are read when the times are asked for (ig_kernel_time_ms), not inside every call */
        for (int i = 0; i < T_COUNT; i++) pending += c->timers[i].ev.size();
        if (pending > 4096) drain_timers(c);
    }
    take_max_readback(c);
    c->full_windows = false;
    for (int i = 0; i < n_moves; i++)
        if (results[i].error) return fail("device-side consistency failure %d at move %d", results[i].error, i);
    return 0;
}

/* the moves [0, n_moves) of the uploaded lists: speculative batches of up to Wmax moves (Wmax == 1: one move per launch
 * sequence).  `ready(first, count)` is called before moves [first, first + count) are enqueued: the fused draw + step entry
 * point waits there for its drawing thread and uploads the candidate lists drawn so far. */
template <class Ready>
static int run_moves(ig_ctx* c, int n_moves, int max_c, int Wmax, Ready ready)
{
    if (Wmax == 1) {
        for (int i = 0; i < n_moves; i++) {
            if (ready(i, 1)) return -1;
            enqueue_move(c, i, max_c, -1, 2);
            enqueue_apply(c, i, 0, 0);
        }
        return 0;
    }
    /* speculative batches: score W moves against the same state, commit the conflict-free prefix on the device,
     * finish a winner that needs the exact delta pass with the one-move tail, continue after it */
    /* The width follows the conflict rate: where few contigs are left (late in an assembly) nearly every move touches a
     * contig an earlier move of the batch modified, and slots scored behind the first conflict are wasted work.  Moving
     * average of the moves a batch got through (a batch that got through all of them counts double: the run was at
     * least that long); the next batch is 1.5 x that, at most Wmax.  Results do not depend on the widths. */
    if (c->w_ema <= 0.0 || c->w_ema > Wmax) c->w_ema = Wmax;
    int done = 0;
    while (done < n_moves) {
        const int w_want = std::max(2, std::min(Wmax, (int)(1.5 * c->w_ema + 1.5)));
        const int w_now = std::min(w_want, n_moves - done);
        if (ready(done, w_now)) return -1;
        if (ensure_window_buffers(c)) return -1; /* the longest contig may have grown */
        enqueue_score(c, done, w_now, max_c, -1, 2);
        int next = 0;
        if (commit_loop(c, done, w_now, &next)) return -1;
        if (next < 0) { /* the first slot did not fit: more room, the same batch again (nothing was committed) */
            if (next == -1) {
                if (grow_slice_pool(c)) return -1;
            } else if (next == -3) { /* (wait_commit has set exact_next: the same batch again without the screening tier) */
            } else {
                if (c->exact_grid >= c->mb.work_cap) return fail("the exact kernel's work list cannot hold the first move of a batch");
                c->exact_grid = std::min(c->mb.work_cap, c->exact_grid * 4);
            }
            continue;
        }
        if (next < w_now && c->last_stop == 1 && (size_t)c->mb.pool_cap < slice_pool_max(c)) {
            /* cut short by the slice pool, not by a conflict: twice the room for the batches to come (the pool starts at 2 Z
             * entries; small problems with long contigs need more than that for 24 slots) */
            HIPCK(hipStreamSynchronize(c->stream));
            if (alloc_slice_pool(c, std::min(slice_pool_max(c), (size_t)c->mb.pool_cap * 2))) return -1;
        } else if (w_now == w_want) { /* a batch cut short by the end of the run says nothing */
            c->w_ema = 0.6 * c->w_ema + 0.4 * (next >= w_now ? std::min(2.0 * w_now, (double)Wmax) : (double)next);
        }
        done += next;
    }
    return 0;
}

extern "C" int ig_step_batch(ig_ctx* c, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c,
                             ig_move_result* results)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (n_moves <= 0) return 0;
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("ig_step_batch: max_c out of range");
    if (c->world > 1) return fail("ig_step_batch: this handle scores a contact shard (ig_set_shard %d/%d): use ig_step_begin / all-reduce / ig_step_finish", c->rank, c->world);
    const int Wmax = c->init_links_inverse ? batch_width(c, max_c) : 1;
    c->full_windows = (Wmax == 1 && n_moves > 1); /* moves enqueued one behind the other: no host round trip to follow the contig lengths */
    if (ensure_move_buffers(c, std::max(8, (int)max_c), Wmax)) return -1;
    if (upload_moves(c, n_moves, frags, cands, max_c)) return -1;
    if (run_moves(c, n_moves, max_c, Wmax, [](int, int) { return 0; })) return -1;
    return download_results(c, n_moves, results);
}

/* n_moves consecutive step_sampler calls INCLUDING their first step, the candidate draw (CL:1403-1408 -> return_neighbours
 * CL:3103-3141): a host thread draws the lists of the moves ahead on the caller's copy of numpy's MT19937 state
 * (ig_draw.cpp) while the launches of the moves in flight run; the lists are uploaded as they appear.  The draws do not
 * depend on the genome, so the result is the same as drawing before each move.  cands_out [n_moves x n_neighbours]
 * receives the lists (sorted, -1 padded). */
extern "C" int ig_step_batch_draw(ig_ctx* c, ig_neighbours* nb, uint32_t* mt_key624, int32_t* mt_pos, int32_t n_moves,
                                  const int32_t* frags, int32_t n_neighbours, int32_t* cands_out, ig_move_result* results)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (n_moves <= 0) return 0;
    if (!nb || !mt_key624 || !mt_pos || !cands_out) return fail("ig_step_batch_draw: NULL argument");
    const int max_c = n_neighbours;
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("ig_step_batch_draw: n_neighbours out of range");
    if (c->world > 1) return fail("ig_step_batch_draw: this handle scores a contact shard (ig_set_shard %d/%d)", c->rank, c->world);
    for (int i = 0; i < n_moves; i++)
        if (frags[i] < 0 || frags[i] >= c->N) return fail("fragment %d out of range", frags[i]);
    const int Wmax = c->init_links_inverse ? batch_width(c, max_c) : 1;
    c->full_windows = (Wmax == 1 && n_moves > 1);
    if (ensure_move_buffers(c, std::max(8, (int)max_c), Wmax)) return -1;
    if (ensure_io(c, n_moves, max_c)) return -1;
    /* the lists go to the device from pinned staging (layout of h_stage: result records, fragments, candidates) */
    int* st_frags = nullptr;
    int* st_cands = nullptr;
    if (c->h_stage) {
        st_frags = (int*)((char*)c->h_stage + (size_t)n_moves * sizeof(ig_move_result));
        st_cands = st_frags + n_moves;
        memcpy(st_frags, frags, (size_t)n_moves * sizeof(int));
    }
    HIPCK(hipMemcpyAsync(c->d_frags, st_frags ? st_frags : frags, (size_t)n_moves * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->up_moves = n_moves;
    c->up_max_c = max_c;
    std::atomic<int> drawn(0), draw_rc(0);
    /* the lists of the first moves are drawn here (a call of one batch -- the usual benchmark call is 20 moves -- then needs
     * no thread at all: creating and joining one costs as much as drawing 30 lists), the rest on a thread, ahead of the launches */
    const int chunk = 32;
    const int inline_n = std::min(n_moves, chunk);
    if (ig_neighbours_draw(nb, mt_key624, mt_pos, frags, inline_n, max_c, cands_out)) return fail("candidate draw failed (fragment out of the distributions' range)");
    drawn.store(inline_n, std::memory_order_release);
    std::thread drawer;
    if (n_moves > inline_n)
        drawer = std::thread([&]() {
            for (int i = inline_n; i < n_moves; i += chunk) {
                const int n = std::min(chunk, n_moves - i);
                if (ig_neighbours_draw(nb, mt_key624, mt_pos, frags + i, n, max_c, cands_out + (size_t)i * max_c)) {
                    draw_rc.store(-1, std::memory_order_release);
                    return;
                }
                drawn.store(i + n, std::memory_order_release);
            }
        });
    int uploaded = 0;
    int rc = run_moves(c, n_moves, max_c, Wmax, [&](int first, int count) -> int {
        int have;
        while ((have = drawn.load(std::memory_order_acquire)) < first + count)
            if (draw_rc.load(std::memory_order_acquire)) return fail("candidate draw failed (fragment out of the distributions' range)");
        if (have > uploaded) { /* everything drawn so far: the later batches find their lists on the device already */
            for (int i = uploaded; i < have; i++) {
                int C = 0;
                while (C < max_c && cands_out[(size_t)i * max_c + C] >= 0) C++;
                if (validate_move(c, frags[i], cands_out + (size_t)i * max_c, C)) return -1;
            }
            const int* src = cands_out + (size_t)uploaded * max_c;
            if (st_cands) {
                memcpy(st_cands + (size_t)uploaded * max_c, src, (size_t)(have - uploaded) * max_c * sizeof(int));
                src = st_cands + (size_t)uploaded * max_c;
            }
            HIPCK(hipMemcpyAsync(c->d_cands + (size_t)uploaded * max_c, src, (size_t)(have - uploaded) * max_c * sizeof(int),
                                 hipMemcpyHostToDevice, c->stream));
            uploaded = have;
        }
        return 0;
    });
    if (drawer.joinable())
    drawer.join(); /* the generator state the caller puts back is the one after ALL draws, also on an error */
    if (rc) return -1;
    if (draw_rc.load()) return fail("candidate draw failed");
    return download_results(c, n_moves, results);
}

/* ---- the same, one step at a time, for callers that split the slots of a batch over several GPUs ---------- */

extern "C" int ig_batch_upload(ig_ctx* c, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c, int32_t max_w)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (n_moves <= 0) return fail("ig_batch_upload: no moves");
    if (max_w < 1 || max_w > max_batch_width(c, max_c))
        return fail("ig_batch_upload: batch width %d out of 1..%d (ig_batch_max_width)", max_w, max_batch_width(c, max_c));
    if (c->world > 1) return fail("ig_batch_upload: contact shards (ig_set_shard) and slot splitting are exclusive");
    /* k_commit_batch counts every genome-distance credit exactly once by a rule on mutually inverse initial links -- also for a
     * batch of one move */
    if (!c->init_links_inverse) return fail("ig_batch_upload: the initial prev / next arrays are not mutually inverse: ig_step / ig_step_batch only");
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("ig_batch_upload: max_c out of range");
    if (ensure_move_buffers(c, std::max(8, (int)max_c), max_w)) return -1;
    return upload_moves(c, n_moves, frags, cands, max_c);
}

extern "C" int ig_batch_score(ig_ctx* c, int32_t move0, int32_t W, int32_t slot_begin, int32_t slot_end)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (W < 1 || W > c->mb.capW || move0 < 0 || move0 + W > c->up_moves) return fail("ig_batch_score: batch out of range");
    if (slot_begin < 0 || slot_end > W || slot_begin > slot_end) return fail("ig_batch_score: slot range out of range");
    if (ensure_window_buffers(c)) return -1; /* the longest contig may have grown */
    enqueue_score(c, move0, W, c->up_max_c, -1, 2, slot_begin, slot_end);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int ig_batch_records(ig_ctx* c, void** records, int64_t* bytes_per_slot)
{
    IG_JOIN(c);
    if (!c->mb.rec) return fail("ig_batch_records: no batch buffers yet (ig_batch_upload first)");
    *records = c->mb.rec;
    *bytes_per_slot = (int64_t)c->mb.rec_stride;
    return 0;
}

extern "C" int ig_batch_commit(ig_ctx* c, int32_t move0, int32_t W, int32_t* n_committed)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (W < 1 || W > c->mb.capW || move0 < 0 || move0 + W > c->up_moves) return fail("ig_batch_commit: batch out of range");
    int next = 0;
    if (commit_loop(c, move0, W, &next)) return -1;
    if (next < 0) { /* the first slot did not fit the slice pool / the exact kernel's grid: more room, the caller scores the batch again */
        if (next == -1) {
            if (grow_slice_pool(c)) return -1;
        } else if (next == -3) { /* (the next ig_batch_score is exact in every column: exact_next) */
        } else {
            if (c->exact_grid >= c->mb.work_cap) return fail("the exact kernel's work list cannot hold the first move of a batch");
            c->exact_grid = std::min(c->mb.work_cap, c->exact_grid * 4);
        }
        next = 0;
    } else if (next < W && c->last_stop == 1 && (size_t)c->mb.pool_cap < slice_pool_max(c)) { /* cut short by the slice pool: more room for the batches to come */
        HIPCK(hipStreamSynchronize(c->stream));
        if (alloc_slice_pool(c, std::min(slice_pool_max(c), (size_t)c->mb.pool_cap * 2))) return -1;
    }
    *n_committed = next;
    return 0;
}

extern "C" int ig_batch_results(ig_ctx* c, int32_t n_moves, ig_move_result* results)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (n_moves < 0 || n_moves > c->up_moves) return fail("ig_batch_results: out of range");
    return download_results(c, n_moves, results);
}

extern "C" int ig_set_batch_width(int w)
{
    g_batch_w = std::min(std::max(w, 1), IG_MAX_BATCH);
    return 0;
}

/* bytes of the move buffers: {per-window arrays (strides sN, sM), slice pool, everything else sized by slots and candidates} */
extern "C" int ig_scratch_bytes(ig_ctx* c, int64_t out3[3])
{
    IG_JOIN(c);
    const MoveBuf& m = c->mb;
    const int64_t C = (int64_t)m.capC * m.capW;
    out3[0] = C * ((int64_t)m.sN * 4 * 3 + (int64_t)m.sM * 4 * 2 + (int64_t)m.sM * NSLOT * 8 + (int64_t)NSLOT * NDYN * m.sN * 4);
    out3[1] = m.pool_cap * (m.packed ? 8 : 12);
    out3[2] = C * (int64_t)(SLICE_SEG * 16 + sizeof(CandMeta) + NSLOT * NCODE * sizeof(ColMeta) + (P_STRIDE + Q_STRIDE) * 8 + IG_N_TMP_STRUCT * 8 +
                            NSLOT * 8 + NSLOT * 16 + 16) +
              (int64_t)m.rec_stride * m.capW + (int64_t)m.work_cap * 8 + (int64_t)c->N * 8;
    return 0;
}

extern "C" int ig_batch_stats(ig_ctx* c, int64_t out3[4])
{
    IG_JOIN(c);
    out3[0] = c->n_batches;
    out3[1] = c->n_batch_committed;
    out3[2] = c->n_batch_pending;
    out3[3] = c->n_batch_predicted;
    return 0;
}

extern "C" int ig_step(ig_ctx* c, int32_t frag_a, const int32_t* cands, int32_t C, ig_move_result* out, double* scores)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (c->world > 1) return fail("ig_step: this handle scores a contact shard (ig_set_shard %d/%d): partial sums only -- use ig_step_begin / all-reduce / ig_step_finish", c->rank, c->world);
    if (validate_move(c, frag_a, cands, C)) return -1;
    if (ensure_move_buffers(c, std::max(8, (int)C))) return -1;
    if (ensure_io(c, 1, C)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)C * sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, C, -1, 2);
    if (scores) HIPCK(hipMemcpyAsync(scores, c->mb.scores, (size_t)C * IG_N_TMP_STRUCT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    enqueue_apply(c, 0, 0, 0);
    HIPCK(hipMemcpyAsync(out, c->d_results, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
    if (queue_max_readback(c)) return -1;
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    take_max_readback(c);
    if (out->error) return fail("device-side consistency failure %d", out->error);
    return 0;
}

extern "C" int ig_score_move(ig_ctx* c, int32_t frag_a, const int32_t* cands, int32_t C, double* scores)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (c->world > 1) return fail("ig_score_move: this handle scores a contact shard (ig_set_shard %d/%d): partial sums only -- use ig_step_begin / all-reduce / ig_step_finish", c->rank, c->world);
    if (validate_move(c, frag_a, cands, C)) return -1;
    if (ensure_move_buffers(c, std::max(8, (int)C))) return -1;
    if (ensure_io(c, 1, C)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)C * sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, C, -1, 2);
    HIPCK(hipMemcpyAsync(scores, c->mb.scores, (size_t)C * IG_N_TMP_STRUCT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    return 0;
}

extern "C" int ig_apply(ig_ctx* c, int32_t frag_a, int32_t frag_b, int32_t op)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (c->world > 1) return fail("ig_apply: this handle scores a contact shard (ig_set_shard %d/%d): partial sums only -- use ig_step_begin / all-reduce / ig_step_finish", c->rank, c->world);
    if (op < 0 || op >= IG_N_TMP_STRUCT) return fail("ig_apply: op out of range");
    if (validate_move(c, frag_a, &frag_b, 1)) return -1;
    if (ensure_move_buffers(c, 8)) return -1;
    if (ensure_io(c, 1, 1)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, &frag_b, sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, 1, op, 2);
    enqueue_apply(c, 0, 0, 1);
    ig_move_result r;
    HIPCK(hipMemcpyAsync(&r, c->d_results, sizeof r, hipMemcpyDeviceToHost, c->stream));
    if (queue_max_readback(c)) return -1;
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    take_max_readback(c);
    if (r.error) return fail("device-side consistency failure %d", r.error);
    return 0;
}

extern "C" int ig_set_shard(ig_ctx* c, int32_t rank, int32_t world)
{
    IG_JOIN(c);
    if (world < 1 || rank < 0 || rank >= world) return fail("ig_set_shard: bad shard %d/%d", rank, world);
    HIPCK(hipStreamSynchronize(c->stream));
    c->rank = rank;
    c->world = world;
    return 0;
}

extern "C" int64_t ig_partials_count(ig_ctx* c) { (void)nuis_join(c); return (int64_t)c->mb.capC * P_STRIDE; }
extern "C" void* ig_partials_device_ptr(ig_ctx* c) { (void)nuis_join(c); return c->mb.part; }

extern "C" int ig_step_begin(ig_ctx* c, int32_t frag_a, const int32_t* cands, int32_t C)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (validate_move(c, frag_a, cands, C)) return -1;
    if (ensure_move_buffers(c, std::max(8, (int)C))) return -1;
    if (ensure_io(c, 1, C)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)C * sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, C, -1, 0);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int ig_step_finish(ig_ctx* c, ig_move_result* out, double* scores)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    int C = 0;
    enqueue_move(c, 0, IG_MAX_CANDIDATES, -1, 1);
    (void)C;
    if (scores) {
        MoveCtl mc;
        HIPCK(hipMemcpyAsync(&mc, c->mb.ctl, sizeof mc, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        HIPCK(hipMemcpyAsync(scores, c->mb.scores, (size_t)mc.C * IG_N_TMP_STRUCT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    enqueue_apply(c, 0, 0, 0);
    HIPCK(hipMemcpyAsync(out, c->d_results, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
    if (queue_max_readback(c)) return -1;
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    take_max_readback(c);
    if (out->error) return fail("device-side consistency failure %d", out->error);
    return 0;
}

/* ---- a move and the nuisance step that follows it, in flight together (instagraal.py:217-262 for cycles > 4) ----------
 * step_nuisance_parameters (CL:2961-3051) evaluates the full likelihood under its test parameters on the coordinates of the
 * state BEFORE the move that was just applied (quirk Q12) and needs nothing else from that move but its score: the pass
 * over all contacts does not have to wait for the move.  ig_nuis_begin enqueues the move (score + apply, library stream)
 * and, on a second stream behind the move's k_gather (after which tab_prev is that earlier state), the full pass under
 * p_test; ig_nuis_end waits for both; ig_nuis_accept makes the test parameters the model's. */
/* tab_prev := the state before the move about to be scored (what k_gather does first thing; here ahead of it, so that the
 * nuisance pass can start next to the move instead of behind its launches) */
__global__ void k_catch_up(Tables tab, Tables tab_prev, const int* __restrict__ prev_touched, const Glob* g)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g->n_prev_touched; i += gridDim.x * blockDim.x) {
        const int s = prev_touched[i];
        tab_prev.dist[s] = tab.dist[s];
        tab_prev.stot[s] = tab.stot[s];
        tab_prev.cp[s] = tab.cp[s];
        tab_prev.len[s] = tab.len[s];
    }
}

/* pinned and mapped: the kernels write a step's results there themselves and raise a flag (no copy, no stream synchronisation) */
static int ensure_host_nuis(ig_ctx* c)
{
    if (c->host_nuis) return 0;
    NuisHost* hp = nullptr;
    if (hipHostMalloc((void**)&hp, sizeof(NuisHost), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        void* dp = nullptr;
        memset(hp, 0, sizeof(NuisHost));
        c->host_nuis = hp;
        if (!(getenv("IG_NO_HOST_FLAG") && atoi(getenv("IG_NO_HOST_FLAG"))) && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess)
            c->host_nuis_dev = (NuisHost*)dp;
        (void)hipGetLastError();
        return 0;
    }
    (void)hipGetLastError();
    HIPCK(hipHostMalloc((void**)&c->host_nuis, sizeof(NuisHost), hipHostMallocDefault));
    memset(c->host_nuis, 0, sizeof(NuisHost));
    return 0;
}

/* spin until *flag == seq (written by a kernel into mapped host memory) or `stream` has drained; true: the flag is there */
static bool wait_host_flag(volatile int* flag, int seq, hipStream_t stream)
{
    for (unsigned spin = 0;; spin++) {
        if (*flag == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return true;
        }
        if ((spin & 0xfff) == 0xfff && hipStreamQuery(stream) != hipErrorNotReady) return *flag == seq;
    }
}

/* ---- the Metropolis test from a screened pass (ig_kernels_nuis.cuh) ------------------------------------------------------
 * IG_NUIS_SCREEN=0: every step through the exact pass, as before; IG_NUIS_SCREEN_VERIFY=1: both, the bound checked on the host */
static int g_nuis_screen = -1, g_nuis_screen_verify = 0, g_nuis_hist_trace = 0;
static int g_nuis_hist = -1; /* tier 0 of the screened pass: ig_set_nuis_hist / IG_NUIS_HIST (0: off, 1: where its cost model says, 2: always) */
static int g_nuis_chain = -1; /* chains of pairs decided on the device: ig_set_nuis_chain / IG_NUIS_CHAIN (default 1) */
static int g_nuis_w = -1;    /* moves scored ahead per launch in a run of ig_nuis_step_begin: env IG_NUIS_W, ig_set_nuis_width; 0: follow the run lengths */
static void nuis_latch_env()
{
    if (g_nuis_hist < 0) g_nuis_hist = getenv("IG_NUIS_HIST") ? atoi(getenv("IG_NUIS_HIST")) : 1;
    if (g_nuis_w < 0) g_nuis_w = getenv("IG_NUIS_W") ? std::max(0, atoi(getenv("IG_NUIS_W"))) : 0;
    if (g_nuis_chain < 0) g_nuis_chain = getenv("IG_NUIS_CHAIN") ? atoi(getenv("IG_NUIS_CHAIN")) : 1;
    if (g_nuis_screen < 0) g_nuis_screen = getenv("IG_NUIS_SCREEN") ? atoi(getenv("IG_NUIS_SCREEN")) : 1;
    g_nuis_screen_verify = getenv("IG_NUIS_SCREEN_VERIFY") ? atoi(getenv("IG_NUIS_SCREEN_VERIFY")) : 0; /* per run: a test toggles it */
    g_nuis_hist_trace = getenv("IG_NUIS_HIST_TRACE") ? atoi(getenv("IG_NUIS_HIST_TRACE")) : 0;
}
static bool nuis_screen_usable(ig_ctx* c)
{
    /* (the environment is read on the caller's thread, where a run begins -- nuis_latch_env: this function also runs on the helper
     * thread, and getenv next to a setenv of the interpreter's thread is undefined) */
    static const int s_tiled = getenv("IG_FULL_TILED") ? atoi(getenv("IG_FULL_TILED")) : 1;
    return (g_nuis_screen || g_nuis_screen_verify) && c->nuis_spec && c->host_nuis_dev && s_tiled && c->tiled_cc && c->n_tile_work > 0 &&
           c->score_const && c->screen_const && c->pz_tab && g_full_hist != 0;
}
extern "C" int ig_set_nuis_screen(int on)
{
    g_nuis_screen = on ? 1 : 0;
    return 0;
}

/* the exact tiles kernel over the list the step's k_tile_trans left (tables, signatures, constants of the test set are in place) */
static void launch_nuis_exact_tiles(ig_ctx* c, hipStream_t s3, long long* out = nullptr, bool publish = true)
{
    if (!out) out = c->scratch_nuis;
    static const int s_grid = getenv("IG_FULL_GRID") ? atoi(getenv("IG_FULL_GRID")) : 512;
    const int per = TILE_TRANS_THREADS / 64;
    const int n_trans = (c->n_tile_info + per - 1) / per;
    const int grid = std::min(c->n_tile_work, s_grid);
    hipLaunchKernelGGL(k_full_nz_tiled, dim3(grid), dim3(FULL_TILED_THREADS), sizeof(FullTiledLds), s3, c->tile_work, c->tiled_cc, c->tabrec,
                       c->tab_prev.len, c->full_const, c->lgf_tab, c->M, c->pz_n1, out, c->n_tile_static, (TileDyn*)c->tile_dyn,
                       c->tile_dyn_list, (long long*)nullptr, publish ? c->host_nuis_dev : (NuisHost*)nullptr, publish ? ++c->sums_seq : 0, c->tile_partial,
                       n_trans);
    if (publish) c->nuis_pub_sums = true;
}

/* tier 0 of the screened pass (ig_kernels_nuis.cuh): the histogram of the cis contacts' distances.  IG_NUIS_HIST=0: off */
static bool nuis_hist_usable(ig_ctx* c)
{
    if (g_nuis_hist < 0) g_nuis_hist = 1; /* (IG_NUIS_HIST: nuis_latch_env) */
    if (!g_nuis_hist || !nuis_screen_usable(c)) return false;
    if (g_nuis_hist >= 2) return true; /* (2: whatever the cost model says -- tests) */
    /* The histogram pays where the pass over the contacts is long and the moves are local.  Per step it costs its evaluation
     * (~16 us) plus, for the share p of the moves that change the genome, the walk over the contacts inside the move's two contigs
     * -- about 2 Z / n_contigs of them, 10 atomics each at ~24 G/s (profiles/r03_microbench.txt); the pass over the contacts costs
     * ~20 us of launches + the cis tiles' 8 bytes per contact at ~5 TB/s.  Few long contigs (the late stage of an assembly: 4
     * contigs of 1 000 bins, 58 % of the moves change the genome) turn the balance: 3.0 k (move + step)/s with the histogram
     * there, 3.9 k without.  With hysteresis; a histogram that is switched off is dropped (the walks stop) and rebuilt when it
     * comes back. */
    const double Z = (double)c->Z, nc = (double)std::max(c->n_contigs_seen, 1), p = c->nh_p_changed;
    /* (the pass over the contacts is void where a table is longer than its staged part and a contig longer than that -- it reads P_z
     * by rank distance from LDS only -- and the exact pass, ~2.5 x the float one, runs instead; the histogram keeps every rank distance) */
    const bool t1_void = std::max(c->pz_n, c->pz_n1) > LDS_PZ && c->max_SL > LDS_PZ;
    const double t0 = 16.0 + p * (2.0 * Z / nc) * 10.0 / 24e3, t1 = 20.0 + Z * 8.0 * 0.4 / 5e6 * (t1_void ? 2.5 : 1.0);
    if (c->n_contigs_seen <= 0) return c->nh_policy_on; /* (no batch decided yet) */
    if (c->nh_policy_on ? t0 > 1.25 * t1 : t0 < 0.8 * t1) {
        c->nh_policy_on = !c->nh_policy_on;
        if (!c->nh_policy_on) {
            c->nh_valid = false;
            c->nh_pending_slot = -1;
        }
    }
    return c->nh_policy_on;
}
extern "C" int ig_set_nuis_hist(int on)
{
    g_nuis_hist = on < 0 ? 0 : std::min(on, 2);
    return 0;
}
static int ensure_nuis_hist(ig_ctx* c)
{
    const int dh_n = std::max(c->M, LDS_PZ); /* a rank distance is below the number of sub-fragments */
    if (c->nh.bins && c->nh.dh_n == dh_n) return 0;
    if (c->nh.bins) {
        hipFree(c->nh.bins);
        hipFree(c->nh.dh);
        hipFree(c->nh.misc);
        hipFree(c->scratch_hist);
        c->nh = NuisHist{nullptr, nullptr, nullptr, 0};
        c->scratch_hist = nullptr;
    }
    DALLOC(c->nh.bins, (size_t)NH_NB * 4);
    DALLOC(c->nh.dh, (size_t)dh_n + 1);
    DALLOC(c->nh.misc, NH_MISC);
    c->nh.dh_n = dh_n;
    DALLOC(c->scratch_hist, 16);
    HIPCK(hipMemset(c->scratch_hist, 0, 16 * sizeof(long long)));
    c->nh_valid = false;
    return 0;
}
/* the last move of the run is not in the histogram yet (a step is evaluated on the state BEFORE its move): walk it in now -- before
 * anything replaces what the walk reads: the move's slot in the batch buffers, and tab_prev as of before the move */
static int nh_flush_pending(ig_ctx* c)
{
    const int w = c->nh_pending_slot;
    c->nh_pending_slot = -1;
    if (w < 0 || !c->nh_valid || !c->nh.bins) return 0;
    const int s_blocks = 128;
    hipLaunchKernelGGL(k_hist_walk, dim3(s_blocks), dim3(256), 0, c->stream3, c->rowptr, c->cc, c->tab_prev, c->glob, c->mb, w, c->nh);
    HIPCK(hipEventRecord(c->ev_walk, c->stream3));
    HIPCK(hipStreamWaitEvent(c->stream, c->ev_walk, 0));
    c->nhs[6] += 1.0;
    return 0;
}

/* what every tier of a step's screened pass needs first: the test set's tables and constants, tab_prev caught up (fuse_catch) and
 * packed, the scratch words cleared (k_nuis_prepare) */
static int launch_nuis_prepare(ig_ctx* c, const ig_params& hp, float mean_kb, hipStream_t s3, bool fuse_catch)
{
    if (!c->diff_const) {
        DALLOC(c->diff_const, 1);
        DALLOC(c->scratch_diff, 8);
        HIPCK(hipFuncSetAttribute((const void*)k_full_diff_tiled<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DiffLds)));
        HIPCK(hipFuncSetAttribute((const void*)k_full_diff_tiled<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DiffLds)));
    }
    const Tables& t = c->tab_prev;
    const int n_pack = (c->M + FULL_TB - 1) / FULL_TB;
    const int n_const = (std::max(std::max(c->pz_n1, LDS_PZ + 2), std::max((int)IG_TAB_SIZE, LDS_LGF)) + 255) / 256;
    hipLaunchKernelGGL(k_nuis_prepare, dim3(n_pack + n_const), dim3(256), 0, s3, c->glob, 1, hp, mean_kb, c->pz_tab1, c->pz_n1, c->lgf_tab, c->full_const,
                       c->scratch_nuis, t, c->M, c->tabrec, c->tile_sig, FULL_TB, c->tile_dyn, n_pack, c->score_const, c->pz_n, c->diff_const,
                       c->scratch_diff, c->screen_const, fuse_catch ? c->tab : Tables{nullptr, nullptr, nullptr, nullptr}, c->prev_touched);
    /* the library stream's next kernels replace what the catch-up reads: they wait for it (not for the pass behind it) */
    if (fuse_catch) {
        HIPCK(hipEventRecord(c->ev_main, s3));
        HIPCK(hipStreamWaitEvent(c->stream, c->ev_main, 0));
    }
    c->nuis_tiles_listed = false;
    return 0;
}

/* the list of the tiles whose contacts a pass has to read, the all-trans tiles' histogram sums under both sets (k_tile_trans);
 * with_zero: the zero-pixel sum of the test set in the same launch (else: someone else's job -- the histogram tier's launch) */
static void launch_nuis_tile_list(ig_ctx* c, hipStream_t s3, bool with_zero)
{
    const int per = TILE_TRANS_THREADS / 64;
    const int n_trans = (c->n_tile_info + per - 1) / per, n_zero = with_zero ? std::min(256, std::max(32, c->M / 4096)) : 0;
    if (n_trans + n_zero > 0) /* (a matrix of a single tile has no off-diagonal ones to list) */
        hipLaunchKernelGGL(k_tile_trans, dim3(n_trans + n_zero), dim3(TILE_TRANS_THREADS), 0, s3, c->tile_info, c->n_tile_info, c->tile_sig, c->tile_hist,
                           c->full_const, (TileDyn*)c->tile_dyn, c->tile_dyn_list, 1, c->tile_partial, n_trans, c->tab_prev, c->glob, 1, c->M,
                           c->scratch_nuis + 2, c->score_const, c->tile_partial0);
    c->nuis_tiles_listed = true;
}

/* tier 1: the float pass over the contacts (k_full_diff_tiled) behind the tile list */
static int launch_nuis_diff(ig_ctx* c, const ig_params& hp, hipStream_t s3, bool with_zero)
{
    launch_nuis_tile_list(c, s3, with_zero);
    const int per = TILE_TRANS_THREADS / 64;
    const int n_trans = (c->n_tile_info + per - 1) / per;
    /* two workgroups per CU less two: the one-wave kernels of the move next to the pass (decide, commit) find a SIMD with registers
     * to spare at once (the stream workgroups take all 512 VGPRs of a SIMD between them).  (Measured and dropped: k_tile_trans's
     * blocks as the head of this kernel's grid, the stream workgroups waiting for them in front of the list: 95 - 112 us instead of
     * 11 + 68 - 84 -- 400 head blocks with this kernel's footprint hold the machine before the first contact is read.) */
    static const int s_grid = getenv("IG_FULL_GRID") ? atoi(getenv("IG_FULL_GRID")) : 510;
    const int s_grid_side = 256;
    const int grid = std::min(c->n_tile_work, c->side_busy ? s_grid_side : s_grid);
    {
        TimedLaunch tl(c, T_DIFF, s3);
        /* a proposal that leaves slope and amplitude alone (d_max, trans level): the kernel without transcendental functions; the
         * device has the last word (DiffConst.zdy: the other kernel's pass comes out void) */
        const bool zdy = hp.slope == c->par_model.slope && hp.c1 == c->par_model.c1 && hp.fact == c->par_model.fact;
        ++c->diff_seq;
        if (zdy)
            hipLaunchKernelGGL(k_full_diff_tiled<true>, dim3(grid), dim3(DIFF_THREADS), sizeof(DiffLds), s3, c->tile_work, c->tiled_cc, c->tabrec,
                               c->diff_const, c->M, c->scratch_diff, c->n_tile_static, (TileDyn*)c->tile_dyn, c->tile_dyn_list, c->host_nuis_dev,
                               c->diff_seq, c->tile_partial, c->tile_partial0, n_trans, c->scratch_nuis, c->diff_trace);
        else
            hipLaunchKernelGGL(k_full_diff_tiled<false>, dim3(grid), dim3(DIFF_THREADS), sizeof(DiffLds), s3, c->tile_work, c->tiled_cc, c->tabrec,
                               c->diff_const, c->M, c->scratch_diff, c->n_tile_static, (TileDyn*)c->tile_dyn, c->tile_dyn_list, c->host_nuis_dev,
                               c->diff_seq, c->tile_partial, c->tile_partial0, n_trans, c->scratch_nuis, c->diff_trace);
    }
    c->nuis_tier = 1;
    if (g_nuis_screen_verify) { /* the exact pass behind it, unconditionally */
        launch_nuis_exact_tiles(c, s3);
        c->nuis_exact_queued = true;
    }
    return 0;
}

/* tier 0: the histogram (built here if it is not valid: the first step of a run, or after moves outside one), its evaluation and
 * the zero-pixel sum of the test set in one launch */
static int launch_nuis_hist(ig_ctx* c, hipStream_t s3)
{
    if (ensure_nuis_hist(c)) return -1;
    if (!c->nh_valid) {
        HIPCK(hipMemsetAsync(c->nh.bins, 0, (size_t)NH_NB * 4 * sizeof(long long), s3));
        HIPCK(hipMemsetAsync(c->nh.dh, 0, ((size_t)c->nh.dh_n + 1) * sizeof(long long), s3));
        HIPCK(hipMemsetAsync(c->nh.misc, 0, NH_MISC * sizeof(long long), s3));
        hipLaunchKernelGGL(k_hist_build, dim3(2048), dim3(256), 0, s3, c->rowptr, c->cc, c->tab_prev, c->M, c->nh);
        c->nh_valid = true;
        c->nhs[7] += 1.0;
    }
    const int n_zero = std::min(256, std::max(32, c->M / 1024));
    ++c->diff_seq;
    hipLaunchKernelGGL(k_hist_eval, dim3(n_zero + NH_NB / 256 + NH_DH_BLOCKS), dim3(256), 0, s3, c->nh, c->glob, c->full_const, c->score_const,
                       c->diff_const, c->scratch_hist, c->host_nuis_dev, c->diff_seq, c->tab_prev, c->M, c->scratch_nuis + 2, n_zero, c->scratch_nuis,
                       PzTab{c->pz_tab1, c->pz_n1}, PzTab{c->pz_tab, c->pz_n});
    c->nuis_tier = 0;
    return 0;
}

/* a step's screened pass: the tier it starts with */
static int launch_nuis_screened(ig_ctx* c, const ig_params& hp, float mean_kb, hipStream_t s3, bool fuse_catch)
{
    if (launch_nuis_prepare(c, hp, mean_kb, s3, fuse_catch)) return -1;
    c->nuis_diff = true;
    c->nuis_exact_queued = false;
    c->nuis_pub_sums = false;
    if (nuis_hist_usable(c)) return launch_nuis_hist(c, s3);
    return launch_nuis_diff(c, hp, s3, true);
}

/* tab_prev := the state before the move about to be decided, then (second stream) the full pass under p_test on it */
static int enqueue_nuis_pass(ig_ctx* c, const float p_test[8], float mean_subfrag_kb)
{
    if (ensure_host_nuis(c)) return -1;
    if (!c->pz_tab1) DALLOC(c->pz_tab1, PZ_MAX);
    if (nh_flush_pending(c)) return -1; /* (before tab_prev catches up with the last move) */
    /* the nuisance pass first (second stream), the move behind it (library stream): the pass is the longer of the two and
     * would otherwise start only when the host is through with the move's dozen launches */
    hipStream_t s3 = c->stream3;
    bool on_side = false, fuse_catch = false;
    const bool use_diff = nuis_screen_usable(c);
    if (!c->nuis_caught_up) {
        if (c->main_drained && use_diff) {
            fuse_catch = on_side = true; /* the screened pass's first launch does it (k_nuis_prepare), see launch_nuis_diff */
        } else if (c->main_drained) {
            /* nothing is queued on the library stream (the previous step's results are on the host, nothing was promoted since):
             * the catch-up runs at the head of the pass's own stream -- an event from the library stream to this one costs ~15 us
             * of idle queue at the start of every step -- and the library stream's next kernels, which replace what it reads,
             * wait for IT (that wait is not on the step's critical path) */
            hipLaunchKernelGGL(k_catch_up, dim3(16), dim3(256), 0, s3, c->tab, c->tab_prev, c->prev_touched, c->glob);
            HIPCK(hipEventRecord(c->ev_main, s3));
            HIPCK(hipStreamWaitEvent(c->stream, c->ev_main, 0));
            on_side = true;
        } else {
            hipLaunchKernelGGL(k_catch_up, dim3(16), dim3(256), 0, c->stream, c->tab, c->tab_prev, c->prev_touched, c->glob);
            HIPCK(hipEventRecord(c->ev_gathered, c->stream)); /* also: behind an accepted step's kernels (ig_nuis_accept), which read what the pass overwrites */
        }
    }
    c->nuis_caught_up = false;
    c->main_drained = false;
    const ig_params hp = {p_test[0], p_test[1], p_test[2], p_test[3], p_test[4], p_test[5], p_test[6], p_test[7]};
    c->nuis_test = hp;
    c->nuis_mean_kb = mean_subfrag_kb;
    const double need = (mean_subfrag_kb > 0) ? (double)p_test[5] / (double)mean_subfrag_kb + 2.0 : 0.0;
    c->pz_n1 = (need > 0 && need < (double)PZ_MAX) ? (int)need : ((need >= (double)PZ_MAX) ? PZ_MAX : 0);
    if (!on_side) HIPCK(hipStreamWaitEvent(s3, c->ev_gathered, 0));
    c->nuis_pub_sums = false;
    c->nuis_diff = false;
    c->nuis_screen_rejected = false;
    if (use_diff) return launch_nuis_screened(c, hp, mean_subfrag_kb, s3, fuse_catch);
    c->pub_sums = (c->nuis_spec && c->host_nuis_dev) ? c->host_nuis_dev : nullptr; /* launch_full_nz: the tiled kernel's last workgroup publishes */
    const bool zero_done = launch_full_nz(c, c->tab_prev, 1, c->scratch_nuis, PzTab{c->pz_tab1, c->pz_n1}, s3, c->scratch_nuis + 2, &hp, mean_subfrag_kb);
    c->pub_sums = nullptr;
    if (!zero_done) hipLaunchKernelGGL(k_full_zero, dim3(128), dim3(256), 0, s3, c->tab_prev, c->glob, 1, c->M, c->scratch_nuis + 2);
    if (!c->nuis_pub_sums) HIPCK(hipMemcpyAsync(c->host_nuis->sums, c->scratch_nuis, 8 * sizeof(long long), hipMemcpyDeviceToHost, s3));
    return 0;
}

extern "C" int ig_nuis_begin(ig_ctx* c, int32_t frag_a, const int32_t* cands, int32_t C, const float p_test[8], float mean_subfrag_kb)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (c->world > 1) return fail("ig_nuis_begin: this handle scores a contact shard");
    if (c->nuis_in_flight) return fail("ig_nuis_begin: the previous step was not ended (ig_nuis_end)");
    nuis_latch_env();
    if (validate_move(c, frag_a, cands, C)) return -1;
    if (ensure_move_buffers(c, std::max(8, (int)C))) return -1;
    if (ensure_io(c, 1, C)) return -1;
    if (ensure_host_nuis(c)) return -1;
    c->host_nuis->frag = frag_a;
    for (int i = 0; i < C; i++) c->host_nuis->cands[i] = cands[i];
    HIPCK(hipMemcpyAsync(c->d_frags, &c->host_nuis->frag, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, c->host_nuis->cands, (size_t)C * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->nuis_in_flight = true;
    c->nuis_spec = false;
    c->spec_valid = false;
    c->spec_slot = 0;
    if (enqueue_nuis_pass(c, p_test, mean_subfrag_kb)) return -1;
    /* the move */
    enqueue_move(c, 0, C, -1, 2);
    enqueue_apply(c, 0, 0, 0);
    HIPCK(hipMemcpyAsync(&c->host_nuis->res, c->d_results, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
    if (queue_max_readback(c)) return -1;
    HIPCK(hipGetLastError());
    return 0;
}

/* ---- the same for a RUN of (move, nuisance step) pairs, the moves scored ahead in speculative batches ----------------------
 * A rejected nuisance step changes nothing a move reads, so the moves behind it can be scored before its outcome is
 * known: ig_nuis_run_begin uploads the lists of the run; ig_nuis_step_begin(i) enqueues step i's pass, scores a batch of
 * moves starting at i if move i has no valid scores yet (first step, after an accepted step, after a conflict, batch used
 * up), and decides + applies move i ALONE from its records (k_decide_batch over one slot); ig_nuis_end / ig_nuis_accept as
 * above.  An accepted step invalidates the slots scored ahead.  Same results as one move and one step at a time. */
static int nuis_spec_width(ig_ctx* c)
{
    if (g_nuis_w < 0) g_nuis_w = 0; /* (IG_NUIS_W: nuis_latch_env) */
    const int s_w = g_nuis_w;
    const int cap = std::min(c->mb.capW, IG_MAX_BATCH);
    if (s_w > 0) return std::min(s_w, cap);
    if (c->spec_ema <= 0.0) c->spec_ema = 3.0;
    return std::max(1, std::min(cap, (int)(1.5 * c->spec_ema + 1.5)));
}

/* 1: the initial prev / next arrays (ig_upload_state / ig_set_initial_genome) are mutually inverse -- what the batch commit's
 * genome-distance bookkeeping relies on; 0: moves are applied one at a time */
extern "C" int ig_links_inverse(ig_ctx* c) { IG_JOIN(c); return c->init_links_inverse ? 1 : 0; }

extern "C" int ig_set_nuis_width(int w)
{
    g_nuis_w = std::max(0, w);
    return 0;
}

extern "C" int ig_nuis_run_begin(ig_ctx* c, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (c->world > 1) return fail("ig_nuis_run_begin: this handle scores a contact shard");
    if (c->nuis_in_flight) return fail("ig_nuis_run_begin: a step is in flight (ig_nuis_end)");
    nuis_latch_env();
    if (n_moves <= 0) return fail("ig_nuis_run_begin: no moves");
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("ig_nuis_run_begin: max_c out of range");
    if (!c->init_links_inverse) return fail("ig_nuis_run_begin: the initial prev / next arrays are not mutually inverse (ig_links_inverse): ig_nuis_begin, one pair at a time");
    if (nh_flush_pending(c)) return -1; /* the last move of the run before: into the histogram while its slot is still there */
    const int s_cap = 24; /* (wider structural batches: 40 / 64 slots 8.7 / 8.0 k instead of 9.1 k -- slots behind the first conflict are wasted) */
    if (g_nuis_w < 0) g_nuis_w = 0; /* (IG_NUIS_W: nuis_latch_env) */
    const int Wmax = std::max(1, std::min(std::max(s_cap, g_nuis_w), max_batch_width(c, max_c)));
    if (ensure_move_buffers(c, std::max(8, (int)max_c), Wmax)) return -1;
    if (upload_moves(c, n_moves, frags, cands, max_c)) return -1;
    c->nuis_spec = true;
    c->spec_valid = false;
    c->spec_prev_pending = false;
    c->spec_base = c->spec_W = c->spec_next = c->spec_move = c->spec_slot = 0;
    c->spec_par_begin = c->spec_par_end = 0;
    c->full_windows = false;
    return 0;
}

/* Widths.  The PARAMETER width (slots screened / scored per launch) follows the number of moves decided between two accepted
 * steps; the STRUCTURAL width (slots gathered, mutated and sliced per launch) the number of moves a batch gets through before
 * a conflict -- an accepted step voids only the former.  IG_NUIS_W / ig_set_nuis_width fixes both (tests). */
static int nuis_struct_width(ig_ctx* c)
{
    const int cap = std::min(c->mb.capW, IG_MAX_BATCH);
    if (g_nuis_w > 0) return std::min(g_nuis_w, cap);
    if (c->spec_struct_ema <= 0.0) c->spec_struct_ema = cap;
    return std::max(nuis_spec_width(c), std::min(cap, (int)(1.5 * c->spec_struct_ema + 1.5)));
}

/* the parameter-dependent scores ahead are of no use any more (accepted step) */
static void nuis_par_invalidate(ig_ctx* c)
{
    if (c->spec_valid && c->spec_par_end > c->spec_par_begin) {
        const int used = c->spec_next - c->spec_par_begin, had = c->spec_par_end - c->spec_par_begin;
        if (used >= 0) { /* a piece used up counts double: the run was at least that long */
            const double len = (used >= had) ? 2.0 * had : (double)used;
            c->spec_ema = 0.7 * (c->spec_ema > 0 ? c->spec_ema : 3.0) + 0.3 * len;
        }
    }
    c->spec_par_begin = c->spec_par_end = c->spec_next;
}

/* the batch in the buffers is of no use any more (conflict, used up, a run's end) */
static void nuis_spec_invalidate(ig_ctx* c)
{
    if (c->spec_valid && c->spec_W > 0) {
        const double len = (c->spec_next >= c->spec_W) ? 2.0 * c->spec_W : (double)c->spec_next;
        c->spec_struct_ema = 0.7 * (c->spec_struct_ema > 0 ? c->spec_struct_ema : (double)c->spec_W) + 0.3 * len;
        nuis_par_invalidate(c);
    }
    c->spec_valid = false;
}

/* score a batch of moves starting at `move`: the structural half of all its slots, the parameter half of the first ones */
/* chains of pairs decided on the device (ig_nuis_chain_begin): 1 (default; env IG_NUIS_CHAIN, ig_set_nuis_chain): the runs score the
 * parameter half of EVERY slot of a batch (a chain takes the slots as far as the batch stands, an accepted step is rare where chains
 * pay) and predict the windowed winners' deltas (a pending move would end a chain) */
static bool nuis_chain_on() { return g_nuis_chain != 0; }
extern "C" int ig_set_nuis_chain(int on)
{
    g_nuis_chain = on ? 1 : 0;
    return 0;
}

static int nuis_spec_score(ig_ctx* c, int move)
{
    if (nh_flush_pending(c)) return -1; /* (the buffers of the last move's slot are about to be overwritten) */
    c->spec_changed = false;
    nuis_spec_invalidate(c);
    const int W = std::min(nuis_struct_width(c), c->up_moves - move);
    const int r = nuis_chain_on() ? W : std::min(nuis_spec_width(c), W);
    if (ensure_window_buffers(c)) return -1; /* the longest contig may have grown */
    c->no_predict = !nuis_chain_on(); /* (decided one move per step: a pause for a windowed winner costs nothing the step would not wait for anyway) */
    enqueue_score(c, move, W, c->up_max_c, -1, 2, 0, W, 0, r, false);
    c->no_predict = false;
    c->spec_base = move;
    c->spec_W = W;
    c->spec_next = 0;
    c->spec_par_begin = 0;
    c->spec_par_end = r;
    c->spec_valid = true;
    c->spec_prev_pending = false;
    return 0;
}

/* the structural half of the batch stands (no conflict so far, slots left): the parameter half of the next slots -- after an
 * accepted step (new parameters), or because the piece scored before is used up */
static bool nuis_spec_can_rescore(const ig_ctx* c, int move)
{
    return c->spec_valid && c->spec_next < c->spec_W && c->spec_base + c->spec_next == move;
}
static int nuis_spec_rescore(ig_ctx* c)
{
    nuis_par_invalidate(c);
    const int pb = c->spec_next, pe = nuis_chain_on() ? c->spec_W : std::min(c->spec_W, pb + nuis_spec_width(c));
    c->no_predict = !nuis_chain_on();
    enqueue_score(c, c->spec_base, c->spec_W, c->up_max_c, -1, 2, 0, c->spec_W, pb, pe, true);
    c->no_predict = false;
    c->spec_par_begin = pb;
    c->spec_par_end = pe;
    return 0;
}

extern "C" int ig_nuis_step_begin(ig_ctx* c, int32_t move, const float p_test[8], float mean_subfrag_kb)
{
    IG_JOIN(c);
    return nuis_step_begin_impl(c, move, p_test, mean_subfrag_kb);
}
static int nuis_step_begin_impl(ig_ctx* c, int32_t move, const float p_test[8], float mean_subfrag_kb)
{
    HIPCK(hipSetDevice(c->device));
    if (!c->nuis_spec) return fail("ig_nuis_step_begin: no run (ig_nuis_run_begin)");
    if (c->nuis_in_flight) return fail("ig_nuis_step_begin: the previous step was not ended (ig_nuis_end)");
    if (move != c->spec_move || move >= c->up_moves) return fail("ig_nuis_step_begin: move %d, expected %d of %d", move, c->spec_move, c->up_moves);
    c->nuis_in_flight = true;
    c->nh_tracking = true; /* until the step is ended: the one move it applies is walked into the histogram at the head of the next step */
    /* a batch is scored in this step (or still being scored: right after an accepted step): the pass leaves half of every CU
     * to it; else it takes the machine (all that runs next to it is one decision and one apply) */
    const bool restruct = !nuis_spec_can_rescore(c, move);
    const bool repar = !restruct && c->spec_next >= c->spec_par_end; /* the piece scored under the current parameters is used up */
    c->side_busy = restruct || repar || c->spec_next == c->spec_par_begin;
    if (enqueue_nuis_pass(c, p_test, mean_subfrag_kb)) return -1;
    c->side_busy = true;
    if (restruct) {
        if (nuis_spec_score(c, move)) return -1;
    } else if (repar) {
        if (nuis_spec_rescore(c)) return -1;
    }
    /* the result record reaches the host as soon as the move is applied: written by k_commit_batch itself where the host
     * memory is mapped, else copied (and copied in the rare cases ig_nuis_end has to redo the move) */
    c->nuis_pub_res = c->host_nuis_dev != nullptr;
    launch_commit(c, c->spec_base, c->spec_next + 1, c->spec_next, c->spec_prev_pending ? 0 : 1, c->nuis_pub_res);
    if (!c->nuis_pub_res) {
        HIPCK(hipMemcpyAsync(&c->host_nuis->res, c->d_results + move, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
        if (queue_max_readback(c)) return -1;
    }
    HIPCK(hipGetLastError());
    return 0;
}

/* the host half of a step of a run: the decision of move spec_move is in (or the slot has to be scored again) */
static int nuis_spec_finish(ig_ctx* c)
{
    bool redone = false;
    for (int attempt = 0;; attempt++) {
        int bo[12];
        const int w = c->spec_next;
        if (wait_commit(c, bo, w == c->spec_par_begin, w)) return -1;
        if (c->own_screened == 1 && w == c->spec_par_begin && !(bo[2] && bo[0] == 0 && bo[1] < 0))
            c->exact_grid = std::min(c->mb.work_cap, std::max(exact_grid_floor(c, c->up_max_c), (int)(1.25 * bo[6]) + 2048));
        if (bo[0] == w + 1) {
            c->spec_prev_pending = false;
            break;
        }
        if (bo[1] == w) { /* a windowed winner without a predicted delta: the one-move tail */
            enqueue_apply(c, c->spec_base + w, w, 0, true); /* (its contigs onto the batch's list of modified ones) */
            c->n_batch_pending++;
            c->spec_prev_pending = false;
            redone = true;
            break;
        }
        /* not decided: a contig of the move was modified by an earlier move of the batch (w > 0), or the first slot did not
         * fit the slice pool / the exact kernel's grid: score a batch from this move */
        if (w == 0) {
            if (bo[2] == 1) {
                if (grow_slice_pool(c)) return -1;
            } else if (bo[2] == 2) {
                if (c->exact_grid >= c->mb.work_cap) return fail("the exact kernel's work list cannot hold the first move of a batch");
                c->exact_grid = std::min(c->mb.work_cap, c->exact_grid * 4);
            } else if (bo[2] == 3) { /* (a score of exactly 0.0: exact_next is set, the scoring below leaves the screening tier out) */
            } else {
                Glob hg;
                HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
                return fail("device-side consistency failure %d at move %d of a run", hg.error, c->spec_move);
            }
            c->spec_valid = false; /* says nothing about run lengths */
        }
        if (attempt > 8) return fail("move %d of a run could not be decided", c->spec_move);
        if (nuis_spec_score(c, c->spec_move)) return -1;
        launch_commit(c, c->spec_base, 1, 0, 0);
        redone = true;
    }
    c->spec_slot = c->spec_next;
    c->nuis_nzb_copied = false;
    if (redone) {
        c->nuis_pub_res = false;
        HIPCK(hipMemcpyAsync(&c->host_nuis->res, c->d_results + c->spec_move, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
        /* ... and the maintained sum of the state the move was decided against (the decide step left it in the slot's control
         * block): the step's screened pass starts from it */
        HIPCK(hipMemcpyAsync((void*)c->host_nuis->nzb, &c->mb.ctl[c->spec_slot].nzb_hi, 2 * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        c->nuis_nzb_copied = true;
        if (queue_max_readback(c)) return -1;
    }
    c->spec_next++;
    c->spec_move++;
    return 0;
}

/* z of a pass from its eight sums (as ig_full_likelihood) */
static double nuis_z_from_sums(const long long h_in[8])
{
    long long h[8];
    memcpy(h, h_in, sizeof h);
    ig_acc_normalize((int64_t*)&h[2], (int64_t*)&h[3]);
    const double log_e = 0.43429448190325182;
    double n_tot_pxl;
    float v_inter;
    const int vi_bits = (int)h[7];
    memcpy(&n_tot_pxl, &h[5], sizeof n_tot_pxl);
    memcpy(&v_inter, &vi_bits, sizeof v_inter);
    return ig_acc_to_double(h[2], h[3]) * log_e + log_e * (n_tot_pxl - (double)h[4]) * -1.0 * (double)v_inter;
}

/* Tu = {temperature, u} of the Metropolis test when the caller is ig_nuis_step_next: a step whose screened interval lies
 * below T ln u is rejected without the exact pass (nuis_screen_rejected; *nz_test is then the interval's midpoint) */
static int nuis_end_body(ig_ctx* c, ig_move_result* out, double* nz_test, double* z_test, int64_t* limbs5, const double* Tu);
static int nuis_end_impl(ig_ctx* c, ig_move_result* out, double* nz_test, double* z_test, int64_t* limbs5, const double* Tu)
{
    const int rc = nuis_end_body(c, out, nz_test, z_test, limbs5, Tu);
    c->nh_tracking = false;
    if (rc) { /* (whatever went wrong: the histogram is rebuilt by the next step that wants it) */
        c->nh_valid = false;
        c->nh_pending_slot = -1;
    }
    return rc;
}
static int nuis_end_body(ig_ctx* c, ig_move_result* out, double* nz_test, double* z_test, int64_t* limbs5, const double* Tu)
{
    HIPCK(hipSetDevice(c->device));
    if (!c->nuis_in_flight) return fail("ig_nuis_end: no step in flight");
    c->nuis_in_flight = false;
    c->nuis_screen_rejected = false;
    c->nuis_accept_certain = false;
    const auto w0 = std::chrono::steady_clock::now();
    if (c->nuis_spec && nuis_spec_finish(c)) return -1;
    bool have_nzb = false;
    int moved = 1; /* the move changed the genome (0 only when k_commit_batch said so) */
    c->last_moved = true;
    if (c->nuis_spec && c->nuis_pub_res && wait_host_flag(&c->host_nuis->res_seq, c->res_seq, c->stream)) {
        c->max_L = std::max(c->max_L, c->host_nuis->max_L);
        c->max_SL = std::max(c->max_SL, c->host_nuis->max_SL);
        have_nzb = true; /* the record came from k_commit_batch, with the maintained sum of the state before the move */
        moved = c->host_nuis->changed;
        c->last_moved = moved != 0;
    } else {
        if (c->nuis_spec && c->nuis_pub_res) { /* no flag although the stream has drained: fetch the record the plain way */
            HIPCK(hipMemcpyAsync(&c->host_nuis->res, c->d_results + c->spec_move - 1, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
            if (queue_max_readback(c)) return -1;
        }
        HIPCK(hipStreamSynchronize(c->stream));
        take_max_readback(c);
        have_nzb = c->nuis_spec && c->nuis_nzb_copied;
    }
    /* the screened pass: decide from its interval where that is possible -- the histogram's first (tier 0), then the float pass
     * over the contacts (tier 1), then the exact pass */
    bool scr_valid = false, scr0_valid = false;
    double scr_mid = 0.0, scr_B = 0.0, scr0_mid = 0.0, scr0_B = 0.0;
    if (c->nuis_diff) {
        NuisHost* hn = c->host_nuis;
        bool reject = false, accept = false;
        c->nscr[0] += 1.0;
        for (;;) {
            const int tier = c->nuis_tier;
            if (!wait_host_flag(&hn->diff_seq, c->diff_seq, c->stream3)) {
                HIPCK(hipStreamSynchronize(c->stream3));
                if (tier == 0) { /* (its words are cleared by the launch itself: the copies it leaves) */
                    long long t3[3];
                    HIPCK(hipMemcpy(t3, c->scratch_hist + 9, sizeof t3, hipMemcpyDeviceToHost));
                    for (int q = 0; q < 8; q++) hn->diff[q] = 0;
                    hn->diff[2] = t3[0];
                    hn->diff[3] = t3[1];
                    hn->diff[4] = t3[2];
                } else {
                    HIPCK(hipMemcpy((void*)hn->diff, c->scratch_diff, 8 * sizeof(long long), hipMemcpyDeviceToHost));
                }
                HIPCK(hipMemcpy((void*)hn->sums, c->scratch_nuis, 8 * sizeof(long long), hipMemcpyDeviceToHost));
            }
            long long d[8];
            memcpy(d, (const void*)hn->diff, sizeof d);
            scr_valid = false;
            if (tier == 0) c->nhs[0] += 1.0;
            if (d[4] != 0 || d[3] <= 0 || !have_nzb) {
                if (tier == 0) {
                    c->nhs[3] += 1.0;
                    if (g_nuis_hist_trace)
                        fprintf(stderr, "[nuis] move %d: histogram tier void: flags %lld sum %lld bound %lld contacts %lld record %d\n", c->spec_move - 1,
                                d[4], d[2], d[3], d[5], (int)have_nzb);
                    if (d[4] & 1) c->nhs[8] += 1.0;
                    if (d[4] & 2) c->nhs[9] += 1.0;
                    if (d[4] & 4) c->nhs[10] += 1.0;
                    if (!have_nzb) c->nhs[11] += 1.0;
                } else {
                    c->nscr[3] += 1.0; /* void: outside the screening term's domain, or the move did not come out of the batch commit */
                    if (d[4] & 1) c->nscr[8] += 1.0;  /* ... the parameter pair (one-log domain, size of the proposal) */
                    if (d[4] & 2) c->nscr[9] += 1.0;  /* ... a contact (ring, count, rank distance beyond the tables) */
                    if (d[4] & 4) c->nscr[10] += 1.0; /* ... a workgroup's sums (|log2 s|, |y|, not a number) */
                    if (!have_nzb) c->nscr[11] += 1.0;
                }
            } else {
                long long hi = hn->nzb[0] + d[0], lo = hn->nzb[1] + d[1];
                ig_acc_normalize((int64_t*)&hi, (int64_t*)&lo);
                const double base = ig_acc_to_double(hi, lo);
                scr_mid = base + (double)d[2] * (1.0 / DIFF_FIX);
                scr_B = (double)d[3] * (1.0 / DIFF_FIX) + 1e-6 + 1e-14 * (__builtin_fabs(base) + __builtin_fabs(hn->res.o)); /* + the double roundings here */
                scr_valid = true;
                if (tier == 0) {
                    c->nhs[4] += scr_B;
                } else {
                    c->nscr[4] = std::max(c->nscr[4], scr_B);
                    c->nscr[6] += scr_B;
                }
                if (Tu && Tu[0] > 0.0 && Tu[1] > 0.0 && !g_nuis_screen_verify) {
                    const double z = nuis_z_from_sums((const long long*)hn->sums);
                    const double x_hi = (((scr_mid + scr_B) + z) - hn->res.o) / Tu[0], x_lo = (((scr_mid - scr_B) + z) - hn->res.o) / Tu[0];
                    reject = exp(x_hi) <= Tu[1] * (1.0 - 1e-9); /* exp is monotone: every L_test in the interval gives a ratio below u */
                    accept = !reject && exp(x_lo) >= Tu[1] * (1.0 + 1e-9) && c->host_nuis_dev; /* ... above it: accepted whatever the exact sum */
                }
            }
            if (tier == 0) {
                if (reject) c->nhs[1] += 1.0;
                if (accept) c->nhs[2] += 1.0;
                if (!reject && !accept) { /* the histogram does not decide (or the check wants every tier): the pass over the contacts */
                    scr0_valid = scr_valid;
                    scr0_mid = scr_mid;
                    scr0_B = scr_B;
                    if (launch_nuis_diff(c, c->nuis_test, c->stream3, false)) return -1;
                    continue;
                }
            }
            break;
        }
        if (scr_valid && !reject && !accept) c->nscr[7] += 1.0;
        if (accept) {
            /* the exact pass is needed -- for the maintained sum under the new parameters and for the likelihood the step
             * returns -- but not for the decision: it runs on the side stream while the library stream promotes the parameters
             * and re-scores the moves ahead; its sums are promoted in front of the next kernel that reads the maintained sum
             * (flush_pending_sums), the caller fetches the exact value later (ig_nuis_exact_result) */
            if (!c->scratch_exact) {
                DALLOC(c->scratch_exact, 8);
                HIPCK(hipMemset(c->scratch_exact, 0, 8 * sizeof(long long)));
            }
            c->nscr[2] += 1.0;
            if (!c->nuis_tiles_listed) launch_nuis_tile_list(c, c->stream3, false);
            launch_nuis_exact_tiles(c, c->stream3, c->scratch_exact, false);
            HIPCK(hipEventRecord(c->ev_exact, c->stream3));
            c->nuis_sums_pending = true;
            c->nuis_accept_certain = true;
        }
        if (reject) {
            c->nscr[1] += 1.0;
            c->nuis_screen_rejected = true;
        }
        if (reject || accept) {
            c->nuis_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            HIPCK(hipGetLastError());
            drain_timers(c);
            c->main_drained = c->nuis_spec;
            *out = hn->res;
            if (out->error) return fail("device-side consistency failure %d", out->error);
            if (c->nuis_spec) c->nh_pending_slot = (c->nh_valid && moved) ? c->spec_slot : -1;
            if (c->nuis_spec && moved) c->spec_changed = true;
            if (c->nuis_spec) c->nh_p_changed = 0.98 * c->nh_p_changed + 0.02 * (moved ? 1.0 : 0.0);
            if (nz_test) *nz_test = scr_mid;
            if (z_test) *z_test = nuis_z_from_sums((const long long*)hn->sums);
            if (limbs5)
                for (int i = 0; i < 5; i++) limbs5[i] = 0;
            return 0;
        }
        c->nscr[2] += 1.0;
        if (!c->nuis_exact_queued) { /* the exact pass over the same list of tiles */
            if (!c->nuis_tiles_listed) launch_nuis_tile_list(c, c->stream3, false);
            launch_nuis_exact_tiles(c, c->stream3);
        }
    }
    if (!(c->nuis_spec && c->nuis_pub_sums && wait_host_flag(&c->host_nuis->sums_seq, c->sums_seq, c->stream3))) {
        if (c->nuis_spec && c->nuis_pub_sums)
            HIPCK(hipMemcpyAsync(c->host_nuis->sums, c->scratch_nuis, 8 * sizeof(long long), hipMemcpyDeviceToHost, c->stream3));
        HIPCK(hipStreamSynchronize(c->stream3));
    }
    c->nuis_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(); /* ig_debug_nuis_wait */
    HIPCK(hipGetLastError());
    drain_timers(c);
    c->main_drained = c->nuis_spec; /* a run's step: its last kernel on the library stream has delivered (or the stream was synchronised) */
    *out = c->host_nuis->res;
    if (out->error) return fail("device-side consistency failure %d", out->error);
    if (c->nuis_spec) c->nh_pending_slot = (c->nh_valid && moved) ? c->spec_slot : -1; /* (the histogram follows the move at the head of the next step) */
    if (c->nuis_spec && moved) c->spec_changed = true; /* (the batch in the buffers goes stale from here: time to score the next one) */
    if (c->nuis_spec) c->nh_p_changed = 0.98 * c->nh_p_changed + 0.02 * (moved ? 1.0 : 0.0); /* (the histogram tier's cost model, nuis_hist_usable) */
    if (scr_valid || scr0_valid) { /* the exact pass ran as well: how much of the bounds did the screened sums use?  (verify mode: the check) */
        long long e[2] = {c->host_nuis->sums[0], c->host_nuis->sums[1]};
        ig_acc_normalize((int64_t*)&e[0], (int64_t*)&e[1]);
        const double exact = ig_acc_to_double(e[0], e[1]);
        static const int s_nocheck = getenv("IG_NUIS_SCREEN_NOCHECK") ? atoi(getenv("IG_NUIS_SCREEN_NOCHECK")) : 0; /* tuning builds that compute garbage */
        if (scr_valid) {
            const double err = __builtin_fabs(scr_mid - exact);
            if (scr_B > 0.0) c->nscr[5] = std::max(c->nscr[5], err / scr_B);
            if (!(err <= scr_B) && !s_nocheck)
                return fail("screened nuisance pass: |screened - exact| = %.6g exceeds its bound %.6g (move %d)", err, scr_B, c->spec_move - 1);
        }
        if (g_nuis_hist_trace)
            fprintf(stderr, "[nuis] move %d exact %.9f  hist %s mid-exact %.3e B %.3e  pass %s mid-exact %.3e B %.3e\n", c->spec_move - 1, exact,
                    scr0_valid ? "ok" : "--", scr0_mid - exact, scr0_B, scr_valid ? "ok" : "--", scr_mid - exact, scr_B);
        if (scr0_valid) {
            const double err = __builtin_fabs(scr0_mid - exact);
            if (scr0_B > 0.0) c->nhs[5] = std::max(c->nhs[5], err / scr0_B);
            if (!(err <= scr0_B) && !s_nocheck)
                return fail("screened nuisance pass, histogram tier: |screened - exact| = %.6g exceeds its bound %.6g (move %d)", err, scr0_B, c->spec_move - 1);
        }
    }
    long long h[8];
    memcpy(h, (const void*)c->host_nuis->sums, sizeof h);
    ig_acc_normalize((int64_t*)&h[0], (int64_t*)&h[1]);
    ig_acc_normalize((int64_t*)&h[2], (int64_t*)&h[3]);
    if (nz_test) *nz_test = ig_acc_to_double(h[0], h[1]);
    if (z_test) { /* as ig_full_likelihood */
        const double log_e = 0.43429448190325182;
        double n_tot_pxl;
        float v_inter;
        const int vi_bits = (int)h[7];
        memcpy(&n_tot_pxl, &h[5], sizeof n_tot_pxl);
        memcpy(&v_inter, &vi_bits, sizeof v_inter);
        *z_test = ig_acc_to_double(h[2], h[3]) * log_e + log_e * (n_tot_pxl - (double)h[4]) * -1.0 * (double)v_inter;
    }
    if (limbs5)
        for (int i = 0; i < 5; i++) limbs5[i] = h[i];
    return 0;
}

extern "C" int ig_nuis_end(ig_ctx* c, ig_move_result* out, double* nz_test, double* z_test, int64_t* limbs5)
{
    IG_JOIN(c);
    return nuis_end_impl(c, out, nz_test, z_test, limbs5, nullptr);
}

/* {steps screened, rejected from the interval alone, exact passes behind a screened one, void, largest bound, largest used
 * fraction of a bound (where the exact pass ran), sum of the bounds, steps whose interval did not decide} since the handle was made */
extern "C" int ig_debug_nuis_screen_stats(ig_ctx* c, double out12[12])
{
    IG_JOIN(c);
    for (int i = 0; i < 12; i++) out12[i] = c->nscr[i];
    return 0;
}

/* the histogram tier: {evaluations, steps rejected there, accepted there, void, sum of its bounds, largest used fraction of a bound
 * (where the exact pass ran), moves walked into the histogram, builds from scratch} since the handle was made */
extern "C" int ig_debug_nuis_hist_stats(ig_ctx* c, double out12[12])
{
    IG_JOIN(c);
    for (int i = 0; i < 12; i++) out12[i] = c->nhs[i];
    return 0;
}

/* the maintained histogram against one built from scratch from the tables of the state it stands for (the state before the last
 * move of the run; the current state once that move has been walked in, which this call does): *mismatches = words that differ
 * (-1: no histogram) */
extern "C" int ig_debug_nuis_hist_check(ig_ctx* c, int64_t* mismatches)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    *mismatches = -1;
    if (c->nuis_in_flight) return fail("ig_debug_nuis_hist_check: a step is in flight");
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipStreamSynchronize(c->stream3));
    if (!c->nh.bins || !c->nh_valid) return 0;
    if (nh_flush_pending(c)) return -1;
    HIPCK(hipStreamSynchronize(c->stream3));
    const size_t nb = (size_t)NH_NB * 4, nd = (size_t)c->nh.dh_n + 1;
    NuisHist t{nullptr, nullptr, nullptr, c->nh.dh_n};
    DALLOC(t.bins, nb);
    DALLOC(t.dh, nd);
    DALLOC(t.misc, NH_MISC);
    HIPCK(hipMemset(t.bins, 0, nb * sizeof(long long)));
    HIPCK(hipMemset(t.dh, 0, nd * sizeof(long long)));
    HIPCK(hipMemset(t.misc, 0, NH_MISC * sizeof(long long)));
    hipLaunchKernelGGL(k_hist_build, dim3(2048), dim3(256), 0, c->stream3, c->rowptr, c->cc, c->tab, c->M, t);
    HIPCK(hipStreamSynchronize(c->stream3));
    std::vector<long long> a(nb + nd + NH_MISC), b(nb + nd + NH_MISC);
    HIPCK(hipMemcpy(a.data(), c->nh.bins, nb * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(a.data() + nb, c->nh.dh, nd * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(a.data() + nb + nd, c->nh.misc, NH_MISC * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(b.data(), t.bins, nb * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(b.data() + nb, t.dh, nd * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(b.data() + nb + nd, t.misc, NH_MISC * sizeof(long long), hipMemcpyDeviceToHost));
    hipFree(t.bins);
    hipFree(t.dh);
    hipFree(t.misc);
    int64_t bad = 0;
    for (size_t i = 0; i < a.size(); i++)
        if (i != nb + nd + 7) bad += a[i] != b[i];
    bad += a[nb + nd + 7] < b[nb + nd + 7]; /* (the largest rank distance ever entered: a loop bound that never comes down) */
    *mismatches = bad;
    return 0;
}

/* the exact likelihood of the non-zero pixels under the test parameters of the last step that ig_nuis_step_next reported with
 * *accepted = 3 (decided from the screened interval, the exact pass behind the decision) */
extern "C" int ig_nuis_exact_result(ig_ctx* c, double* nz_test)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (c->exact_seq == 0 && !c->nuis_sums_pending) return fail("ig_nuis_exact_result: no step was accepted ahead of its exact pass");
    flush_pending_sums(c);
    NuisHost* hn = c->host_nuis;
    if (!hn) return fail("ig_nuis_exact_result: no run");
    if (!wait_host_flag(&hn->exact_seq, c->exact_seq, c->stream)) {
        HIPCK(hipStreamSynchronize(c->stream));
        if (hn->exact_seq != c->exact_seq) return fail("ig_nuis_exact_result: the promotion of the sums did not report");
    }
    *nz_test = ig_acc_to_double(hn->exact[0], hn->exact[1]);
    return 0;
}

/* the accepted step's parameters become the model's (CL:3032-3036) without another pass over all contacts: the maintained
 * exact sum under the new parameters on the CURRENT state = their full pass on the state before the last move (what the
 * step just evaluated, quirk Q12) + that move's exact delta under them (k_delta over the touched contigs); the zero-pixel sum
 * is recounted (O(M)) */
/* acc8: {zero-pixel hi, lo, pair count (k_full_zero), .., .., .., the move's delta hi, lo (k_delta)} -- all zero between two
 * accepted steps: the promotion clears what it has read.  One launch: thread 0 of block 0 promotes (parameters, maintained
 * sums), every block builds its share of the model's score / screening constants from the SAME parameters (set 1: what set 0
 * is being overwritten with), i.e. k_build_score_const and k_build_screen_const for the promoted set. */
/* mode 0: everything; 1: parameters, constants and the zero-pixel sum only (the exact pass over all contacts is still running:
 * ig_nuis_accept behind a decisively accepted step); 2: the maintained sum, once that pass is through (flush_pending_sums: one
 * block), its exact limbs to the host */
__global__ void __launch_bounds__(256) k_nuis_promote(Glob* g, long long* full_sums, long long* acc8, PzTab pz, const double* __restrict__ lgf_tab,
                                                      ScoreConst* score_const, ScreenConst* screen_const, int mode, NuisHost* hn, int hn_seq)
{
    if (mode != 2) {
        build_score_const_block(g, pz, lgf_tab, score_const, 1);
        build_screen_const_block(g, pz, screen_const, 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (mode != 2) {
            g->par[0] = g->par[1];
            long long h = acc8[0], l = acc8[1];
            ig_acc_normalize((int64_t*)&h, (int64_t*)&l);
            g->z_hi = h;
            g->z_lo = l;
            if (acc8[2] != g->n_intra) g->error = 8; /* the pair count does not depend on the parameters */
            for (int q = 0; q < 6; q++) acc8[q] = 0;
        }
        if (mode != 1) {
            long long h = full_sums[0] + acc8[6], l = full_sums[1] + acc8[7];
            ig_acc_normalize((int64_t*)&h, (int64_t*)&l);
            g->nz_hi = h;
            g->nz_lo = l;
            acc8[6] = acc8[7] = 0;
            if (mode == 2) {
                long long eh = full_sums[0], el = full_sums[1];
                ig_acc_normalize((int64_t*)&eh, (int64_t*)&el);
                for (int q = 0; q < 8; q++) full_sums[q] = 0; /* the side buffer of the deferred pass: zero between two uses */
                if (hn) {
                    hn->exact[0] = eh;
                    hn->exact[1] = el;
                    __threadfence_system();
                    hn->exact_seq = hn_seq;
                }
            }
        }
    }
}

/* the maintained sum of a decisively accepted step, as soon as something is about to read it: behind the exact pass (event) */
static void flush_pending_sums(ig_ctx* c)
{
    if (!c->nuis_sums_pending) return;
    c->nuis_sums_pending = false;
    hipStreamWaitEvent(c->stream, c->ev_exact, 0);
    hipLaunchKernelGGL(k_nuis_promote, dim3(1), dim3(64), 0, c->stream, c->glob, c->scratch_exact, c->scratch_accept, PzTab{c->pz_tab, c->pz_n}, c->lgf_tab,
                       c->score_const, c->screen_const, 2, c->host_nuis_dev, ++c->exact_seq);
}

extern "C" int ig_nuis_accept(ig_ctx* c)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (c->nuis_in_flight) return fail("ig_nuis_accept: end the step first (ig_nuis_end)");
    if (!c->scratch_accept) {
        DALLOC(c->scratch_accept, 8);
        HIPCK(hipMemset(c->scratch_accept, 0, 8 * sizeof(long long)));
    }
    long long* acc8 = c->scratch_accept;
    c->n_accepts++;
    c->main_drained = false;
    const int w = c->spec_slot; /* the slot of the move just applied (0 unless it came out of a batch: ig_nuis_step_begin) */
    /* whatever was scored ahead was scored under the old parameters -- but only its parameter-dependent half: windows, candidate
     * genomes, columns and slice lists of the batch's remaining slots stand (IG_NUIS_KEEP=0: redone as well) */
    if (c->nuis_spec) nuis_par_invalidate(c);
    else nuis_spec_invalidate(c);
    const PzTab pz1{c->pz_tab1, c->pz_n1};
    /* the move's delta under the new parameters; contig membership of the partners as of BEFORE the move: tab_prev.  A move that
     * left the genome as it was (91 % of them: k_commit_batch says so with the record) has none: 46 us of every accepted step */
    if (c->last_moved || !c->nuis_spec)
        hipLaunchKernelGGL(k_delta, dim3(DELTA_RB, 2, 1), dim3(SCORE_THREADS), 0, c->stream, c->rowptr, c->cc, c->tab_prev, c->tab_prev,
                           c->prev_touched, c->glob, c->mb, c->lgf_tab, pz1, w, 2, 1, acc8 + 6);
    hipLaunchKernelGGL(k_full_zero, dim3(128), dim3(256), 0, c->stream, c->tab, c->glob, 1, c->M, acc8);
    /* the promotion, and the tables of the model's parameter set: the test set's P_z table becomes the model's */
    c->par_model = c->nuis_test;
    std::swap(c->pz_tab, c->pz_tab1);
    std::swap(c->pz_n, c->pz_n1);
    const PzTab pz0{c->pz_tab, c->pz_n};
    hipLaunchKernelGGL(k_nuis_promote, dim3((LDS_PZ + 2 + 255) / 256), dim3(256), 0, c->stream, c->glob, c->scratch_nuis, acc8, pz0, c->lgf_tab,
                       c->score_const, c->screen_const, c->nuis_sums_pending ? 1 : 0, (NuisHost*)nullptr, 0);
    HIPCK(hipGetLastError());
    return 0;
}

/* ig_nuis_end, the Metropolis decision of step_nuisance_parameters (CL:3026-3036: ratio = exp((L_test - L_move) / T) >= u),
 * ig_nuis_accept and the next move's ig_nuis_step_begin in ONE call: between the end of a step's kernels and the first
 * launch of the next step there is no host code but this.  The caller supplies the next step's test parameters for both
 * outcomes (they are prepared while this step's kernels run; one of them may be NULL: then, on that outcome, the next step is
 * left to the caller's ig_nuis_step_begin -- the promotion's kernels run meanwhile).  *accepted: 0 / 1, or 2 when exp() lands within 1e-9
 * relative of u -- then nothing was decided or enqueued and the caller goes on with its own arithmetic (ig_nuis_accept,
 * ig_nuis_step_begin). */
extern "C" int ig_nuis_step_next(ig_ctx* c, double temperature, double u, const float p_next_rejected[8], const float p_next_accepted[8],
                                 float mean_subfrag_kb, int32_t has_next, ig_move_result* out, double* nz_test, double* z_test, int32_t* accepted)
{
    IG_JOIN(c);
    if (!c->nuis_spec) return fail("ig_nuis_step_next: no run (ig_nuis_run_begin)");
    double nz = 0.0, z = 0.0;
    const double Tu[2] = {temperature, u};
    if (nuis_end_impl(c, out, &nz, &z, nullptr, Tu)) return -1;
    if (nz_test) *nz_test = nz;
    if (z_test) *z_test = z;
    const double ratio = exp(((nz + z) - out->o) / temperature);
    int acc;
    if (c->nuis_screen_rejected) acc = 0; /* every L_test of the screened interval gives a ratio below u */
    else if (c->nuis_accept_certain) acc = 1; /* ... above u */
    else if (ratio != ratio) acc = 0; /* NaN >= u is false */
    else if (ratio >= u * (1.0 + 1e-9)) acc = 1;
    else if (ratio <= u * (1.0 - 1e-9)) acc = 0;
    else acc = 2;
    *accepted = (acc == 1 && c->nuis_accept_certain) ? 3 : acc; /* 3: accepted, *nz_test is the screened midpoint, the exact value through ig_nuis_exact_result */
    if (acc == 2) return 0;
    if (acc == 1 && ig_nuis_accept(c)) return -1;
    const float* p_next = acc ? p_next_accepted : p_next_rejected;
    if (has_next && p_next) return nuis_defer_step_begin(c, c->spec_move, p_next, mean_subfrag_kb); /* (on the helper thread: see NuisWorker) */
    /* the caller has yet to work out the next test parameters; what is certain is that the moves ahead have to be scored
     * under the parameters just promoted: that needs nothing from the caller, and runs while it computes */
    if (has_next && acc == 1 && c->spec_move < c->up_moves) {
        /* (first what the pass of the next step waits for, or it would queue behind the scoring launches) */
        if (nh_flush_pending(c)) return -1; /* (the histogram's walk of the last move reads tab_prev as of before it) */
        /* INVARIANT: behind a decisively accepted step the exact pass may still be running on the side stream (nuis_sums_pending) while
         * this catch-up rewrites tab_prev.len of the last move's contigs.  The pass reads tab_prev.len on its ring path only, and a ring
         * among the contacts voids both screened tiers (no decisive accept then: the exact pass ran in front of the decision).  Whoever
         * lets a ring through the screened tiers must order this launch behind ev_exact. */
        hipLaunchKernelGGL(k_catch_up, dim3(16), dim3(256), 0, c->stream, c->tab, c->tab_prev, c->prev_touched, c->glob);
        HIPCK(hipEventRecord(c->ev_gathered, c->stream)); /* (the next pass next to the scoring launches, not behind them: 4.8 k against 4.5 k) */
        c->nuis_caught_up = true;
        if (nuis_spec_can_rescore(c, c->spec_move)) {
            if (nuis_spec_rescore(c)) return -1;
        } else if (nuis_spec_score(c, c->spec_move)) {
            return -1;
        }
    }
    return 0;
}


/* ---- chains of (move, nuisance step) pairs decided on the device (ig_common.cuh: ChainIn) ------------------------
 * ig_nuis_chain_begin(move, n_sets, ...): the pairs move .. move + n_sets - 1 of the run, as far as the device can take them alone:
 * segment after segment (k_catch_up, k_chain_prepare, k_hist_eval_chain, k_decide_chain, k_commit_batch on the library stream; the
 * histogram's walk of a move that changed the genome between two segments) until a pair needs the host.  Asynchronous (the
 * helper thread drives the segments; ig_nuis_chain_done polls, ig_nuis_chain_end waits): the caller prepares the proposals of the
 * steps behind meanwhile.  Every pair a chain completes is a move decided exactly as k_decide_batch decides it and a step
 * REJECTED from the histogram tier's interval with the margins of nuis_end_body; the pair a chain stops in front of is untouched. */
enum { CHAIN_R_SETS = 0, /* every set was used: the steps went through */
       CHAIN_R_TEST = 1,     /* the next pair's step is not a certain rejection (accepted, undecided, or its interval void) */
       CHAIN_R_CONFLICT = 2, /* the next move touches a contig an earlier move of its batch modified */
       CHAIN_R_PENDING = 3,  /* ... is a windowed winner without a predicted delta */
       CHAIN_R_OVERFLOW = 4, /* ... did not fit a pool / the exact kernel's grid / holds a score of exactly 0.0 */
       CHAIN_R_NO_SLOTS = 5, /* no slot scored under the model's parameters is left (batch used up, parameters promoted, first step) */
       CHAIN_R_UNSUPPORTED = 6 /* the histogram tier is not in use (its cost model, verify mode, no mapped host memory) */ };
static int ensure_chain(ig_ctx* c)
{
    if (c->chain_sets) return 0;
    DALLOC(c->chain_sets, CHAIN_SEG);
    DALLOC(c->chain_in, CHAIN_MAX);
    DALLOC(c->chain_tests, CHAIN_SEG);
    DALLOC(c->chain_out16, (size_t)CHAIN_SEG * 16);
    DALLOC(c->chain_zs, (size_t)CHAIN_SEG * 8);
    HIPCK(hipMemset(c->chain_out16, 0, (size_t)CHAIN_SEG * 16 * sizeof(long long)));
    return 0;
}
static int nuis_chain_impl(ig_ctx* c, int32_t move, int32_t n_sets, float mean_kb)
{
    c->chain_done = 0;
    c->chain_reason = CHAIN_R_UNSUPPORTED;
    HIPCK(hipSetDevice(c->device));
    if (!c->nuis_spec) return fail("ig_nuis_chain_begin: no run (ig_nuis_run_begin)");
    if (c->nuis_in_flight) return fail("ig_nuis_chain_begin: a step is in flight (ig_nuis_end)");
    if (move != c->spec_move || move >= c->up_moves) return fail("ig_nuis_chain_begin: move %d, expected %d of %d", move, c->spec_move, c->up_moves);
    c->n_chain_calls++;
    auto leave = [&](int reason) {
        c->chain_reason = reason;
        c->n_chain_stops[reason]++;
        return 0;
    };
    if (ensure_host_nuis(c)) return -1;
    if (!c->host_bo || g_nuis_screen_verify || !nuis_hist_usable(c)) return leave(CHAIN_R_UNSUPPORTED);
    if (!c->spec_valid || c->spec_next >= c->spec_par_end || c->spec_base + c->spec_next != move) return leave(CHAIN_R_NO_SLOTS);
    if (ensure_chain(c) || ensure_nuis_hist(c)) return -1;
    hipStream_t st = c->stream;
    HIPCK(hipMemcpyAsync(c->chain_in, c->chain_in_host, (size_t)n_sets * sizeof(ChainIn), hipMemcpyHostToDevice, st));
    c->nh_tracking = true; /* the moves applied below are followed by the histogram (walked in behind their segment) */
    struct Untrack {
        ig_ctx* c;
        ~Untrack() { c->nh_tracking = false; }
    } untrack{c};
    if (nh_flush_pending(c)) return -1; /* the last move of the step before (walk on the side stream, the library stream waits for it) */
    c->main_drained = false;
    int done = 0, reason = CHAIN_R_SETS;
    const PzTab pz0{c->pz_tab, c->pz_n};
    for (;;) {
        const int w0 = c->spec_next;
        const int avail = std::min(std::min(CHAIN_SEG, n_sets - done), std::min(c->spec_par_end - w0, c->up_moves - c->spec_move));
        if (avail <= 0) {
            reason = (n_sets - done <= 0) ? CHAIN_R_SETS : CHAIN_R_NO_SLOTS;
            break;
        }
        /* tab_prev := the current state (the zero-pixel sums of the test sets are taken on it; every step of the segment up to
         * its first changing move sees this state as "before its move", quirk Q12) */
        if (!c->nuis_caught_up) hipLaunchKernelGGL(k_catch_up, dim3(16), dim3(256), 0, st, c->tab, c->tab_prev, c->prev_touched, c->glob);
        c->nuis_caught_up = false;
        const int n_const = (std::max(std::max(PZ_MAX, LDS_PZ + 2), std::max((int)IG_TAB_SIZE, LDS_LGF)) + 255) / 256;
        hipLaunchKernelGGL(k_chain_prepare, dim3(n_const, avail), dim3(256), 0, st, c->glob, c->chain_in, done, mean_kb, c->chain_sets, c->lgf_tab,
                           c->score_const, c->pz_n, c->screen_const, c->chain_out16, c->chain_zs);
        if (!c->nh_valid) { /* (the first step of a run, or after moves outside one: from the tables just caught up) */
            HIPCK(hipMemsetAsync(c->nh.bins, 0, (size_t)NH_NB * 4 * sizeof(long long), st));
            HIPCK(hipMemsetAsync(c->nh.dh, 0, ((size_t)c->nh.dh_n + 1) * sizeof(long long), st));
            HIPCK(hipMemsetAsync(c->nh.misc, 0, NH_MISC * sizeof(long long), st));
            hipLaunchKernelGGL(k_hist_build, dim3(2048), dim3(256), 0, st, c->rowptr, c->cc, c->tab_prev, c->M, c->nh);
            c->nh_valid = true;
            c->nhs[7] += 1.0;
        }
        const int n_zero = std::min(256, std::max(32, c->M / 1024));
        hipLaunchKernelGGL(k_hist_eval_chain, dim3(n_zero + NH_NB / 256 + NH_DH_BLOCKS, avail), dim3(256), 0, st, c->nh, c->glob, c->chain_sets,
                           c->score_const, c->chain_out16, c->tab_prev, c->M, c->chain_zs, n_zero, pz0, c->chain_tests);
        flush_pending_sums(c);
        const int zcheck = (c->own_screened == 1 ? 1 : 0) | (std::max(g_zero_inject, 0) << 8);
        const ChainArgs ca{c->chain_tests, c->chain_in, done, 1.0};
        hipLaunchKernelGGL(k_decide_chain, dim3(1), dim3(64), 0, st, c->glob, c->mb, c->d_results, c->spec_base, w0 + avail, w0, c->dirty_buf, c->batch_out,
                           (volatile int*)c->host_bo_dev, ++c->bo_seq, c->spec_prev_pending ? 0 : 1, zcheck, ca);
        hipLaunchKernelGGL(k_commit_batch, dim3(1), dim3(COMMIT_THREADS), 0, st, c->st, c->tab, c->tab_prev, c->glob, c->mb, c->init_prev, c->init_next,
                           c->orientable, c->black, c->own_tag, c->own_idx, c->prev_touched, c->d_results, c->spec_base, w0 + avail, w0, c->batch_out,
                           (NuisHost*)nullptr, 0);
        int bo[12];
        if (wait_commit(c, bo, w0 == c->spec_par_begin, w0)) return -1;
        if (c->own_screened == 1 && w0 == c->spec_par_begin && !(bo[2] && bo[0] == w0 && bo[1] < 0))
            c->exact_grid = std::min(c->mb.work_cap, std::max(exact_grid_floor(c, c->up_max_c), (int)(1.25 * bo[6]) + 2048));
        const int j = bo[0] - w0;
        if (j < 0 || j > avail) return fail("ig_nuis_chain: the decide step reported %d of %d pairs", j, avail);
        c->n_chain_segments++;
        c->n_chain_pairs += j;
        c->spec_next += j;
        c->spec_move += j;
        done += j;
        if (j > 0) {
            c->spec_prev_pending = false;
            c->spec_slot = c->spec_next - 1;
            c->nhs[0] += j; /* evaluations of the histogram tier that decided a step, all of them rejections */
            c->nhs[1] += j;
            c->nscr[0] += j;
            c->nscr[1] += j;
            const bool moved = bo[11] != 0;
            c->last_moved = moved;
            for (int q = 0; q < j; q++) c->nh_p_changed = 0.98 * c->nh_p_changed + 0.02 * ((moved && q == j - 1) ? 1.0 : 0.0);
            if (moved) { /* the histogram follows the move before tab_prev does (here, on the library stream: the commit is still running) */
                hipLaunchKernelGGL(k_hist_walk, dim3(128), dim3(256), 0, st, c->rowptr, c->cc, c->tab_prev, c->glob, c->mb, c->spec_slot, c->nh);
                c->nhs[6] += 1.0;
            }
        }
        if (bo[10] == 5 || (bo[10] == 0 && bo[1] < 0 && j == avail)) continue; /* behind a changing move / the segment went through */
        if (bo[1] >= 0) reason = CHAIN_R_PENDING;
        else if (bo[10] == 4) reason = CHAIN_R_TEST;
        else if (bo[10] == 0) reason = CHAIN_R_CONFLICT;
        else reason = CHAIN_R_OVERFLOW;
        break;
    }
    HIPCK(hipGetLastError());
    c->nh_pending_slot = -1;
    c->chain_done = done;
    return leave(reason);
}

extern "C" int ig_nuis_chain_begin(ig_ctx* c, int32_t move, int32_t n_sets, const float* p_tests, const double* u, const double* temperature,
                                   float mean_subfrag_kb)
{
    IG_JOIN(c);
    if (n_sets < 1 || n_sets > CHAIN_MAX) return fail("ig_nuis_chain_begin: 1..%d sets (got %d)", CHAIN_MAX, n_sets);
    if (!c->chain_in_host) HIPCK(hipHostMalloc((void**)&c->chain_in_host, CHAIN_MAX * sizeof(ChainIn), hipHostMallocDefault));
    for (int k = 0; k < n_sets; k++) {
        ChainIn& ci = c->chain_in_host[k];
        memcpy(ci.p, p_tests + 8 * (size_t)k, sizeof ci.p);
        /* the step is rejected iff exp((L_test - L_move) / T) < u; on the device: (upper end of L_test) - L_move <= T (ln u - 2e-9), the
         * host's 1e-9 relative margin on u doubled and taken in the exponent (nuis_end_body); anything else is not decided there */
        const double T = temperature[k], uu = u[k];
        ci.ln_u = (T > 0.0 && uu > 0.0) ? T * (log(uu) - 2e-9) - 1e-12 * __builtin_fabs(T * log(uu)) : -IG_INF;
    }
    c->chain_busy = true;
    return nuis_defer(c, 1, move, nullptr, mean_subfrag_kb, n_sets);
}
/* 1: the chain has ended (ig_nuis_chain_end returns at once), 0: its segments are still being driven */
extern "C" int ig_nuis_chain_done(ig_ctx* c)
{
    NuisWorker* w = c->worker;
    return (!w || w->state.load(std::memory_order_acquire) != 1) ? 1 : 0;
}
extern "C" int ig_nuis_chain_end(ig_ctx* c, int32_t* n_done, int32_t* reason)
{
    IG_JOIN(c);
    if (!c->chain_busy) return fail("ig_nuis_chain_end: no chain was begun");
    c->chain_busy = false;
    *n_done = c->chain_done;
    *reason = c->chain_reason;
    return 0;
}
/* {calls, segments, pairs completed, ends by reason (CHAIN_R_*: 7 values)} since the handle was made */
extern "C" int ig_debug_nuis_chain_stats(ig_ctx* c, int64_t out10[10])
{
    IG_JOIN(c);
    out10[0] = c->n_chain_calls;
    out10[1] = c->n_chain_segments;
    out10[2] = c->n_chain_pairs;
    for (int q = 0; q < 7; q++) out10[3 + q] = c->n_chain_stops[q];
    return 0;
}

extern "C" int ig_kernel_time_ms(ig_ctx* c, const char* name, double* avg_ms, int64_t* n)
{
    IG_JOIN(c);
    drain_timers(c);
    for (int i = 0; i < T_COUNT; i++)
        if (!strcmp(name, c->timers[i].name)) {
            if (avg_ms) *avg_ms = c->timers[i].n ? c->timers[i].total_ms / (double)c->timers[i].n : 0.0;
            if (n) *n = c->timers[i].n;
            return 0;
        }
    return fail("ig_kernel_time_ms: unknown kernel '%s'", name);
}

extern "C" int ig_reset_timers(ig_ctx* c, int enable)
{
    IG_JOIN(c);
    HIPCK(hipStreamSynchronize(c->stream));
    drain_timers(c);
    for (int i = 0; i < T_COUNT; i++) {
        c->timers[i].total_ms = 0;
        c->timers[i].n = 0;
    }
    c->timing = enable != 0;
    c->timing_mask = enable > 1 ? (unsigned)(enable >> 1) : 0xffffu; /* enable = 1 | (mask << 1) */
    while (enable && c->ev_pool.size() < 16) { /* the first timed launches find their events ready */
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        c->ev_pool.push_back(e);
    }
    for (int i = 0; i < T_COUNT; i++) c->timers[i].seen = 0;
    return 0;
}

/* time every n-th launch of the selected kernels only (default 1: every launch) */
extern "C" int ig_set_timer_sampling(ig_ctx* c, int every)
{
    IG_JOIN(c);
    c->timing_every = std::max(1, every);
    return 0;
}

/* ------------------------------------------------------------------ debug ABI */

extern "C" int ig_debug_eval_terms(ig_ctx* c, const float* s, const float* s_tot, const int32_t* ob, int64_t n, float* ex, float* exc,
                                   double* term, int64_t* q)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (!c->have_params) return fail("ig_debug_eval_terms: set parameters first");
    float *ds, *dst, *dex, *dexc;
    int* dob;
    double* dterm;
    long long* dq;
    DALLOC(ds, (size_t)n);
    DALLOC(dst, (size_t)n);
    DALLOC(dob, (size_t)n);
    DALLOC(dex, (size_t)n);
    DALLOC(dexc, (size_t)n);
    DALLOC(dterm, (size_t)n);
    DALLOC(dq, (size_t)n);
    HIPCK(hipMemcpy(ds, s, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(dst, s_tot, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(dob, ob, n * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_terms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ds, dst, dob, (long long)n, c->glob,
                       c->lgf_tab, dex, dexc, dterm, dq);
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(ex, dex, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(exc, dexc, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(term, dterm, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(q, dq, n * sizeof(long long), hipMemcpyDeviceToHost));
    hipFree(ds);
    hipFree(dst);
    hipFree(dob);
    hipFree(dex);
    hipFree(dexc);
    hipFree(dterm);
    hipFree(dq);
    return 0;
}

extern "C" int ig_debug_candidate_state(ig_ctx* c, int32_t cand, int32_t slot, int32_t* soa)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    if (cand < 0 || cand >= c->mb.capC || slot < 0 || slot > IG_N_TMP_STRUCT) return fail("ig_debug_candidate_state: bad index");
    const size_t n = c->N;
    CandMeta m;
    HIPCK(hipMemcpy(&m, c->mb.meta + cand, sizeof m, hipMemcpyDeviceToHost));
    if (m.kidx[slot] < 0) return fail("ig_debug_candidate_state: slot %d was not materialised for candidate %d", slot, cand);
    /* start from the live genome (internal ids), overlay the local window */
    std::vector<int> host(17 * n);
    HIPCK(hipMemcpy(host.data(), c->st_block, 17 * n * sizeof(int), hipMemcpyDeviceToHost));
    const size_t sn = (size_t)c->mb.sN; /* stride of the window arrays */
    std::vector<int> loc((size_t)NDYN * sn), gid(sn);
    HIPCK(hipMemcpy(loc.data(), c->mb.loc + ((size_t)(cand * NSLOT + slot) * NDYN) * sn, (size_t)NDYN * sn * sizeof(int), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(gid.data(), c->mb.Lloc + (size_t)cand * sn, sn * sizeof(int), hipMemcpyDeviceToHost));
    for (int x = 0; x < m.n_loc; x++)
        for (int k = 0; k < NDYN; k++) host[k * n + gid[x]] = loc[k * sn + x];
    static const int dyn_src[NDYN] = {0, 1, 2, 3, 6, 8, 9, 10, 11, 12, 13};
    for (int k = 0; k < NDYN; k++) memcpy(soa + dyn_src[k] * n, &host[k * n], n * sizeof(int));
    memcpy(soa + 4 * n, &host[11 * n], n * sizeof(int));
    memcpy(soa + 5 * n, &host[12 * n], n * sizeof(int));
    memcpy(soa + 14 * n, &host[14 * n], n * sizeof(int));
    memcpy(soa + 15 * n, &host[15 * n], n * sizeof(int));
    memcpy(soa + 16 * n, &host[16 * n], n * sizeof(int));
    for (size_t f = 0; f < n; f++) soa[7 * n + f] = (int)f;
    return 0;
}

extern "C" int ig_debug_last_sums(ig_ctx* c, int64_t* nz_hi, int64_t* nz_lo, int64_t* z_hi, int64_t* z_lo, int64_t* n_intra,
                                  int64_t* ext_hi, int64_t* ext_lo, int64_t* n_slice, int32_t* n_uniq, int32_t* uniq)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    MoveCtl mc;
    HIPCK(hipMemcpy(&mc, c->mb.ctl, sizeof mc, hipMemcpyDeviceToHost));
    const int C = mc.C;
    std::vector<long long> part((size_t)C * P_STRIDE), qp((size_t)C * Q_STRIDE);
    std::vector<CandMeta> meta(C);
    HIPCK(hipMemcpy(part.data(), c->mb.part, part.size() * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(qp.data(), c->mb.qpart, qp.size() * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(meta.data(), c->mb.meta, C * sizeof(CandMeta), hipMemcpyDeviceToHost));
    for (int cc_ = 0; cc_ < C; cc_++) {
        const long long* p = &part[(size_t)cc_ * P_STRIDE];
        const long long* q = &qp[(size_t)cc_ * Q_STRIDE];
        n_uniq[cc_] = meta[cc_].n_uniq;
        n_slice[cc_] = slice_total(p);
        int64_t eh = p[P_NZ], el = p[P_NZ + 1];
        ig_acc_normalize(&eh, &el);
        ext_hi[cc_] = eh;
        ext_lo[cc_] = el;
        for (int s = 0; s < IG_N_TMP_STRUCT; s++) {
            const int o = cc_ * IG_N_TMP_STRUCT + s;
            uniq[o] = -1;
            nz_hi[o] = nz_lo[o] = z_hi[o] = z_lo[o] = n_intra[o] = 0;
        }
        for (int k = 1; k <= meta[cc_].n_uniq; k++) {
            const int s = meta[cc_].uniq[k - 1];
            const int o = cc_ * IG_N_TMP_STRUCT + s;
            uniq[cc_ * IG_N_TMP_STRUCT + (k - 1)] = s;
            int64_t h = q[Q_NZFULL + 2 * k], l = q[Q_NZFULL + 2 * k + 1];
            const int r = (int)(slice_total(p) % 64);
            if (r > 0 && (k - 1) >= r) { /* quirk Q5 */
                h -= q[Q_TAIL + 2 * k];
                l -= q[Q_TAIL + 2 * k + 1];
            }
            ig_acc_normalize(&h, &l);
            nz_hi[o] = h;
            nz_lo[o] = l;
            h = hg.z_hi + q[Q_Z + 2 * k] - q[Q_Z];
            l = hg.z_lo + q[Q_Z + 2 * k + 1] - q[Q_Z + 1];
            ig_acc_normalize(&h, &l);
            z_hi[o] = h;
            z_lo[o] = l;
            n_intra[o] = hg.n_intra + q[Q_NI + k] - q[Q_NI];
        }
    }
    return 0;
}

extern "C" int ig_debug_tables(ig_ctx* c, float* dist, int32_t* id_c, float* s_tot, int32_t* pos, int32_t* len)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    const size_t m = c->M;
    HIPCK(hipMemcpy(dist, c->tab.dist, m * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(s_tot, c->tab.stot, m * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(len, c->tab.len, m * 4, hipMemcpyDeviceToHost));
    std::vector<int2> cp(m);
    HIPCK(hipMemcpy(cp.data(), c->tab.cp, m * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; i++) {
        id_c[i] = cp[i].x;
        pos[i] = cp[i].y;
    }
    return 0;
}

/* v_log_f32 / v_exp_f32 over their whole domain against the contract's double functions (the screening bound assumes
 * both below SCR_KL = SCR_KE = 4 in these units): out[0], out[1] as k_transcendental_error defines them */
extern "C" int ig_debug_transcendental_error(ig_ctx* c, double out2[2])
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    double* d;
    DALLOC(d, 2);
    HIPCK(hipMemsetAsync(d, 0, 2 * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_transcendental_error, dim3(256 * 16), dim3(256), 0, c->stream, d);
    HIPCK(hipMemcpyAsync(out2, d, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    hipFree(d);
    return 0;
}

/* {largest |screened - exact| / bound seen, largest bound} of the runs under IG_SCREEN_VERIFY=1; {columns screened, columns
 * scored exactly} since the state was uploaded */
extern "C" int ig_debug_screen_stats(ig_ctx* c, double out4[6])
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    out4[0] = out4[1] = 0.0;
    if (c->screen_worst) HIPCK(hipMemcpy(out4, c->screen_worst, 2 * sizeof(double), hipMemcpyDeviceToHost));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    out4[2] = (double)hg.scr_cols;
    out4[3] = (double)hg.scr_cont;
    out4[4] = (double)hg.scr_terms;
    out4[5] = (double)hg.scr_terms_exact;
    return 0;
}

/* one from-scratch pass with every workgroup of k_full_nz_tiled leaving {start, end (100 MHz clock), XCC_ID << 32 | HW_ID, contacts}:
 * out [4 x n]; n_items receives the number of workgroups */
extern "C" int ig_debug_tile_trace(ig_ctx* c, int64_t* out, int64_t cap, int64_t* n_items)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    *n_items = c->n_tile_work;
    if (!out || cap < c->n_tile_work) return 0;
    long long* d = nullptr;
    DALLOC(d, (size_t)4 * c->n_tile_work);
    HIPCK(hipMemset(d, 0, (size_t)32 * c->n_tile_work));
    *n_items = std::min(c->n_tile_work, 4096);
    c->tile_trace = d;
    double nz;
    const int rc = ig_full_likelihood(c, 0, 0, &nz, nullptr, nullptr);
    c->tile_trace = nullptr;
    if (!rc) HIPCK(hipMemcpy(out, d, (size_t)32 * c->n_tile_work, hipMemcpyDeviceToHost));
    hipFree(d);
    return rc;
}

/* one screened nuisance pass (csrc/ig_kernels_nuis.cuh) under test parameters p_test on the tables of the state before the last
 * move, every workgroup leaving {start, end (100 MHz clock), XCC_ID << 32 | HW_ID, items << 32 | contacts, ticks until its blocks
 * were staged, ticks in its contact loops, 0, 0}: out [8 x n]; *n receives the number of workgroups; sums8 the pass's eight output
 * words */
extern "C" int ig_debug_diff_trace(ig_ctx* c, const float p_test[8], float mean_subfrag_kb, int64_t* out, int64_t cap, int64_t* n, int64_t* sums8)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    if (!c->tiled_cc || c->n_tile_work <= 0 || !c->score_const || !c->screen_const || !c->pz_tab) return fail("ig_debug_diff_trace: no tiled contacts / parameters");
    const int grid = std::min(c->n_tile_work, 512);
    *n = grid;
    if (!out || cap < grid) return 0;
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipStreamSynchronize(c->stream3));
    if (ensure_host_nuis(c)) return -1;
    if (!c->pz_tab1) DALLOC(c->pz_tab1, PZ_MAX);
    const double need = (mean_subfrag_kb > 0) ? (double)p_test[5] / (double)mean_subfrag_kb + 2.0 : 0.0;
    c->pz_n1 = (need > 0 && need < (double)PZ_MAX) ? (int)need : ((need >= (double)PZ_MAX) ? PZ_MAX : 0);
    const ig_params hp = {p_test[0], p_test[1], p_test[2], p_test[3], p_test[4], p_test[5], p_test[6], p_test[7]};
    long long* d = nullptr;
    DALLOC(d, (size_t)8 * grid);
    HIPCK(hipMemset(d, 0, (size_t)64 * grid));
    c->diff_trace = d;
    const bool sb = c->side_busy;
    c->side_busy = false;
    const bool ver = g_nuis_screen_verify;
    g_nuis_screen_verify = 0;
    const int rc = launch_nuis_prepare(c, hp, mean_subfrag_kb, c->stream3, false) || launch_nuis_diff(c, hp, c->stream3, true);
    g_nuis_screen_verify = ver;
    c->side_busy = sb;
    c->diff_trace = nullptr;
    c->nuis_diff = false;
    HIPCK(hipStreamSynchronize(c->stream3));
    if (!rc) HIPCK(hipMemcpy(out, d, (size_t)64 * grid, hipMemcpyDeviceToHost));
    if (!rc && sums8) HIPCK(hipMemcpy(sums8, c->scratch_diff, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    hipFree(d);
    return rc;
}

/* seconds ig_nuis_end (also inside ig_nuis_step_next) has spent waiting for the device since the handle was created */
extern "C" int ig_debug_nuis_wait(ig_ctx* c, double* seconds)
{
    IG_JOIN(c);
    *seconds = c->nuis_wait_s;
    return 0;
}

/* the from-scratch pass with (1, default) / without (0) the count histograms of the all-trans tiles: same sums */
extern "C" int ig_debug_set_full_hist(int on)
{
    g_full_hist = on ? 1 : 0;
    return 0;
}

/* fault injection for the decide step's zero-score rule: every n-th move of a two-tier batch is treated as one whose contenders hold
 * a score of exactly 0.0 (stop code 3: the move is scored again with every column exact); 0 = off.  *fallbacks (may be NULL): how
 * often a handle has taken that path. */
extern "C" int ig_debug_set_zero_inject(int every)
{
    g_zero_inject = every > 0 ? every : 0;
    return 0;
}
extern "C" int ig_debug_zero_fallbacks(ig_ctx* c, int64_t* fallbacks)
{
    IG_JOIN(c);
    *fallbacks = c->n_zero_fallbacks;
    return 0;
}

extern "C" int ig_debug_set_tail_quirk(int on)
{
    g_tail_quirk = on;
    return 0;
}

/* maintained exact sums {nz_hi, nz_lo, z_hi, z_lo, n_intra} and {n_contigs, next_cid, ch_c, ch_k, ch_slot, ch_windowed} */
extern "C" int ig_debug_globals(ig_ctx* c, int64_t* sums5, int32_t* ints6)
{
    IG_JOIN(c);
    HIPCK(hipSetDevice(c->device));
    flush_pending_sums(c);
    HIPCK(hipStreamSynchronize(c->stream));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    sums5[0] = hg.nz_hi;
    sums5[1] = hg.nz_lo;
    sums5[2] = hg.z_hi;
    sums5[3] = hg.z_lo;
    sums5[4] = hg.n_intra;
    ints6[0] = hg.n_contigs;
    ints6[1] = hg.next_cid;
    MoveCtl mc;
    memset(&mc, 0, sizeof mc);
    if (c->mb.ctl) HIPCK(hipMemcpy(&mc, c->mb.ctl, sizeof mc, hipMemcpyDeviceToHost));
    ints6[2] = mc.ch_c;
    ints6[3] = mc.ch_k;
    ints6[4] = mc.ch_slot;
    ints6[5] = mc.ch_windowed;

    return 0;
}
