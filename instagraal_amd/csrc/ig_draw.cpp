/*
 * ig_draw.cpp -- step 1 of step_sampler (CL:1401-1408): the candidate draw, return_neighbours (CL:3103-3141), off the
 * Python interpreter.  Host code only (no device work): part of libinstagraal_hip.so so that a run of moves -- draw,
 * scoring launches, results -- is ONE call, and the draw of the moves ahead overlaps the kernels of the moves in flight.
 *
 * The reference draws with numpy's GLOBAL legacy generator:
 *     np.random.choice(xk, min(n, nnz(pk)), p=pk(float32), replace=False)          CL:3113-3121
 *     np.random.choice(n_frags, n, replace=False)      (a bin without hetero contacts)  CL:3124
 * and every later stochastic decision of the run (the next cycle's shuffle, the nuisance proposals) continues the same
 * stream, so "bit-identical fragment orders under a fixed seed" requires the SAME numbers AND the same generator state
 * afterwards.  What numpy's RandomState.choice does on that path is restated here on a copy of the MT19937 state the
 * caller takes from np.random.get_state() and puts back with set_state():
 *   - doubles: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53 from two 32-bit outputs (legacy random_sample);
 *   - replace=False with p: loop { x = rand(size - n_uniq); p[found] = 0; cdf = cumsum(p) (sequential, double);
 *     cdf /= cdf[-1]; new = searchsorted(cdf, x, side='right'); keep the first occurrence of every value, in draw order };
 *   - replace=False without p: permutation(n)[:size] = Fisher-Yates from the top with masked-rejection integers.
 * tests/test_cpu_abi_and_host.py::test_c_draw_equals_numpy_choice compares lists and generator state with numpy itself.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/instagraal_hip.h"

int ig_fail_msg(const char* msg); /* ig_hip.hip: sets ig_last_error(), returns -1 */

namespace {

struct MT {
    uint32_t* key;
    int pos;
    inline void gen()
    {
        const uint32_t N = 624, M = 397, MATRIX_A = 0x9908b0dfU, UPPER = 0x80000000U, LOWER = 0x7fffffffU;
        uint32_t y;
        uint32_t kk;
        for (kk = 0; kk < N - M; kk++) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        for (; kk < N - 1; kk++) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        y = (key[N - 1] & UPPER) | (key[0] & LOWER);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        pos = 0;
    }
    inline uint32_t next32()
    {
        if (pos == 624) gen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680U;
        y ^= (y << 15) & 0xefc60000U;
        y ^= (y >> 18);
        return y;
    }
    inline double next_double()
    {
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    /* legacy random_interval(max): uniform on [0, max], masked rejection */
    inline uint64_t interval(uint64_t max)
    {
        if (max == 0) return 0;
        uint64_t mask = max, value;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        mask |= mask >> 32;
        if (max <= 0xffffffffULL) {
            while ((value = (next32() & mask)) > max) {}
        } else {
            while ((value = ((((uint64_t)next32()) << 32 | next32()) & mask)) > max) {}
        }
        return value;
    }
};

} // namespace

struct ig_neighbours {
    int32_t n_frags;
    std::vector<int64_t> indptr; /* [n_frags + 1] */
    std::vector<int32_t> xk;     /* partner bins */
    std::vector<double> pk;      /* float32 probabilities widened (what PyArray_FROM_OTF(p, NPY_DOUBLE) holds) */
    std::vector<int32_t> nnz;    /* per bin: entries with pk != 0 */
    std::vector<uint8_t> black;  /* [n_frags] */
    /* scratch */
    std::vector<double> p, cdf, x;
    std::vector<int64_t> found, perm;
};

extern "C" int ig_neighbours_create(const int64_t* indptr, const int32_t* xk, const float* pk, int32_t n_frags, const int32_t* blacklisted,
                                    int32_t n_black, ig_neighbours** out)
{
    if (!out || !indptr || n_frags <= 0) return ig_fail_msg("ig_neighbours_create: bad arguments");
    ig_neighbours* nb = new ig_neighbours();
    nb->n_frags = n_frags;
    nb->indptr.assign(indptr, indptr + (size_t)n_frags + 1);
    const int64_t tot = indptr[n_frags];
    for (int32_t i = 0; i < n_frags; i++)
        if (indptr[i + 1] < indptr[i] || indptr[i] < 0) {
            delete nb;
            return ig_fail_msg("ig_neighbours_create: indptr must be non-decreasing");
        }
    nb->xk.assign(xk, xk + tot);
    nb->pk.resize((size_t)tot);
    nb->nnz.assign((size_t)n_frags, 0);
    for (int32_t i = 0; i < n_frags; i++)
        for (int64_t k = indptr[i]; k < indptr[i + 1]; k++) {
            if (xk[k] < 0 || xk[k] >= n_frags) {
                delete nb;
                return ig_fail_msg("ig_neighbours_create: partner bin out of range");
            }
            nb->pk[(size_t)k] = (double)pk[k];
            nb->nnz[(size_t)i] += (pk[k] != 0.0f);
        }
    nb->black.assign((size_t)n_frags, 0);
    for (int32_t i = 0; i < n_black; i++) {
        if (blacklisted[i] < 0 || blacklisted[i] >= n_frags) {
            delete nb;
            return ig_fail_msg("ig_neighbours_create: blacklisted id out of range");
        }
        nb->black[(size_t)blacklisted[i]] = 1;
    }
    *out = nb;
    return 0;
}

extern "C" void ig_neighbours_destroy(ig_neighbours* nb) { delete nb; }

/* one return_neighbours + the host clean-up of step_sampler (candidates.sort(), CL:1408; the focal bin itself, reachable
 * only through the uniform draw, is dropped: quirk Q13) -> out[0..n_neighbours), -1 padded */
static void draw_one(ig_neighbours* nb, MT& mt, int32_t A, int32_t n_neighbours, int32_t* out)
{
    const int64_t b = nb->indptr[(size_t)A], e = nb->indptr[(size_t)A + 1];
    const int64_t d = e - b;
    int32_t got[IG_MAX_CANDIDATES];
    int n_got = 0;
    if (d > 0) {
        const int size = std::min<int64_t>(n_neighbours, nb->nnz[(size_t)A]);
        nb->p.assign(nb->pk.begin() + b, nb->pk.begin() + e);
        nb->cdf.resize((size_t)d);
        nb->found.resize((size_t)std::max(size, 1));
        double* p = nb->p.data();
        double* cdf = nb->cdf.data();
        int n_uniq = 0;
        while (n_uniq < size) {
            const int nx = size - n_uniq;
            double x[IG_MAX_CANDIDATES];
            for (int q = 0; q < nx; q++) x[q] = mt.next_double();
            for (int q = 0; q < n_uniq; q++) p[nb->found[(size_t)q]] = 0.0;
            double run = 0.0;
            for (int64_t k = 0; k < d; k++) {
                run += p[k];
                cdf[k] = run;
            }
            const double tot = cdf[d - 1];
            for (int64_t k = 0; k < d; k++) cdf[k] /= tot;
            int64_t neu[IG_MAX_CANDIDATES];
            for (int q = 0; q < nx; q++) neu[q] = std::upper_bound(cdf, cdf + d, x[q]) - cdf; /* searchsorted side='right' */
            for (int q = 0; q < nx; q++) { /* np.unique(return_index) + sort of the indices: first occurrences, draw order */
                bool dup = false;
                for (int r = 0; r < q; r++) dup |= (neu[r] == neu[q]);
                if (!dup) nb->found[(size_t)n_uniq++] = neu[q];
            }
        }
        for (int q = 0; q < size; q++) got[n_got++] = nb->xk[(size_t)(b + nb->found[(size_t)q])];
    } else { /* no hetero contact: choice(n_frags, n, replace=False) = permutation(n_frags)[:n] */
        const int64_t n = nb->n_frags;
        nb->perm.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) nb->perm[(size_t)i] = i;
        for (int64_t i = n - 1; i >= 1; i--) {
            const int64_t j = (int64_t)mt.interval((uint64_t)i);
            std::swap(nb->perm[(size_t)i], nb->perm[(size_t)j]);
        }
        const int size = (int)std::min<int64_t>(n_neighbours, n);
        for (int q = 0; q < size; q++) got[n_got++] = (int32_t)nb->perm[(size_t)q];
    }
    int n_out = 0;
    int32_t keep[IG_MAX_CANDIDATES];
    for (int q = 0; q < n_got; q++)
        if (!nb->black[(size_t)got[q]] && got[q] != A) keep[n_out++] = got[q];
    std::sort(keep, keep + n_out);
    for (int q = 0; q < n_neighbours; q++) out[q] = q < n_out ? keep[q] : -1;
}

extern "C" int ig_neighbours_draw(ig_neighbours* nb, uint32_t* mt_key624, int32_t* mt_pos, const int32_t* frags, int32_t n_moves,
                                  int32_t n_neighbours, int32_t* cands_out)
{
    if (!nb || !mt_key624 || !mt_pos || !frags || !cands_out) return ig_fail_msg("ig_neighbours_draw: NULL argument");
    if (n_neighbours < 1 || n_neighbours > IG_MAX_CANDIDATES) return ig_fail_msg("ig_neighbours_draw: n_neighbours out of 1..16");
    if (*mt_pos < 0 || *mt_pos > 624) return ig_fail_msg("ig_neighbours_draw: MT19937 position out of range");
    /* every fragment is checked before the first draw: an error leaves the caller's generator state (key AND position) untouched */
    for (int32_t i = 0; i < n_moves; i++)
        if (frags[i] < 0 || frags[i] >= nb->n_frags) return ig_fail_msg("ig_neighbours_draw: fragment out of range");
    MT mt{mt_key624, *mt_pos};
    for (int32_t i = 0; i < n_moves; i++) {
        draw_one(nb, mt, frags[i], n_neighbours, cands_out + (size_t)i * n_neighbours);
    }
    *mt_pos = mt.pos;
    return 0;
}

/* The stream of a run of moves WITH nuisance sampling (the loop of instagraal.py:217-262 for cycles > 4): per move the
 * neighbour draw, then the three draws of step_nuisance_parameters (CL:2976-3028) --
 *     np.random.choice(4)           one 32-bit output & 3 (legacy masked rejection, range 3: never rejects)
 *     np.random.normal(0, sigma)    legacy polar Box-Muller with its one-value cache (has_gauss, gauss of get_state());
 *                                   returned here as the STANDARD normal g: the caller forms 0.0 + sigma * g as numpy does
 *     np.random.rand()              the acceptance uniform
 * -- whose consumption does not depend on the parameters being sampled (only `skip_normal_3`: the reference draws no normal
 * for the trans-level proposal when its sigma is <= 0), so the whole stream of a run is drawn up front. */
extern "C" int ig_neighbours_draw_nuisance(ig_neighbours* nb, uint32_t* mt_key624, int32_t* mt_pos, int32_t* has_gauss, double* gauss,
                                           const int32_t* frags, int32_t n_moves, int32_t n_neighbours, int32_t skip_normal_3,
                                           int32_t* cands_out, int32_t* id_modif_out, double* normal_out, double* uniform_out)
{
    if (!nb || !mt_key624 || !mt_pos || !has_gauss || !gauss || !frags || !cands_out || !id_modif_out || !normal_out || !uniform_out)
        return ig_fail_msg("ig_neighbours_draw_nuisance: NULL argument");
    if (n_neighbours < 1 || n_neighbours > IG_MAX_CANDIDATES) return ig_fail_msg("ig_neighbours_draw_nuisance: n_neighbours out of 1..16");
    if (*mt_pos < 0 || *mt_pos > 624) return ig_fail_msg("ig_neighbours_draw_nuisance: MT19937 position out of range");
    for (int32_t i = 0; i < n_moves; i++) /* before the first draw: an error leaves the caller's generator state untouched */
        if (frags[i] < 0 || frags[i] >= nb->n_frags) return ig_fail_msg("ig_neighbours_draw_nuisance: fragment out of range");
    MT mt{mt_key624, *mt_pos};
    int hg = *has_gauss;
    double gz = *gauss;
    for (int32_t i = 0; i < n_moves; i++) {
        draw_one(nb, mt, frags[i], n_neighbours, cands_out + (size_t)i * n_neighbours);
        const int id = (int)(mt.next32() & 3u);
        id_modif_out[i] = id;
        double g = 0.0;
        if (!(id == 3 && skip_normal_3)) {
            if (hg) {
                g = gz;
                hg = 0;
                gz = 0.0;
            } else {
                double f, x1, x2, r2;
                do {
                    x1 = 2.0 * mt.next_double() - 1.0;
                    x2 = 2.0 * mt.next_double() - 1.0;
                    r2 = x1 * x1 + x2 * x2;
                } while (r2 >= 1.0 || r2 == 0.0);
                f = std::sqrt(-2.0 * std::log(r2) / r2);
                gz = f * x1;
                hg = 1;
                g = f * x2;
            }
        }
        normal_out[i] = g;
        uniform_out[i] = mt.next_double();
    }
    *mt_pos = mt.pos;
    *has_gauss = hg;
    *gauss = gz;
    return 0;
}
