/* AddressSanitizer / UBSan harness for the one host-only translation unit of the library, csrc/ig_draw.cpp (the candidate
 * draw on numpy's MT19937 stream): random jump distributions incl. empty rows, rows shorter than the number of neighbours
 * asked for, zero weights and blacklisted bins; every draw is checked for its invariants (sorted, distinct, in range, not the
 * focal bin, not blacklisted, -1 padded); the error paths are taken.  Built and run by tests/test_cpu_abi_and_host.py with
 * g++ -fsanitize=address,undefined (GPU AddressSanitizer is not available on the target pool: sanitizers run on the CPU
 * side only). */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/instagraal_hip.h"

static int g_fails = 0;
int ig_fail_msg(const char* msg)
{
    (void)msg;
    g_fails++;
    return -1;
}

static uint64_t rs = 0x9e3779b97f4a7c15ull;
static uint32_t rnd()
{
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (uint32_t)(rs >> 16);
}

#define CHECK(x)                                                        \
    do {                                                                \
        if (!(x)) {                                                     \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #x); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main()
{
    const int N = 400;
    std::vector<int64_t> indptr(N + 1, 0);
    std::vector<int32_t> xk;
    std::vector<float> pk;
    for (int i = 0; i < N; i++) {
        const int len = (i % 17 == 0) ? 0 : (i % 5 == 0 ? 2 : 3 + (int)(rnd() % 40));
        float tot = 0.0f;
        const size_t at = xk.size();
        const int steps[5] = {3, 7, 9, 11, 13}; /* coprime with N: the partners of a bin are distinct, as the columns of a matrix row are */
        const int start = (int)(rnd() % N), step = steps[rnd() % 5];
        for (int k = 0; k < len; k++) {
            xk.push_back((int32_t)((start + k * step) % N));
            const float w = (rnd() % 7 == 0) ? 0.0f : (float)(1 + rnd() % 100);
            pk.push_back(w);
            tot += w;
        }
        for (size_t k = at; k < xk.size(); k++) pk[k] = tot > 0 ? pk[k] / tot : 0.0f;
        indptr[i + 1] = (int64_t)xk.size();
    }
    std::vector<int32_t> black = {3, 77, 250};
    ig_neighbours* nb = nullptr;
    CHECK(ig_neighbours_create(indptr.data(), xk.data(), pk.data(), N, black.data(), (int32_t)black.size(), &nb) == 0 && nb);
    std::vector<uint32_t> key(624);
    for (auto& k : key) k = rnd();
    int32_t pos = 624;
    for (int n_nb = 1; n_nb <= 16; n_nb += 3) {
        std::vector<int32_t> frags(300), out(300 * (size_t)n_nb, -7);
        for (auto& f : frags) f = (int32_t)(rnd() % N);
        CHECK(ig_neighbours_draw(nb, key.data(), &pos, frags.data(), (int32_t)frags.size(), n_nb, out.data()) == 0);
        CHECK(pos >= 0 && pos <= 624);
        for (size_t i = 0; i < frags.size(); i++) {
            int32_t prev = -1;
            bool padded = false;
            for (int q = 0; q < n_nb; q++) {
                const int32_t v = out[i * n_nb + q];
                if (v < 0) {
                    CHECK(v == -1);
                    padded = true;
                    continue;
                }
                CHECK(!padded && v < N && v > prev && v != frags[i] && v != 3 && v != 77 && v != 250);
                prev = v;
            }
        }
    }
    {
        std::vector<int32_t> frags(200), out(200 * 5), idm(200);
        std::vector<double> g(200), u(200);
        for (auto& f : frags) f = (int32_t)(rnd() % N);
        int32_t hg = 0;
        double gz = 0.0;
        for (int skip = 0; skip < 2; skip++) {
            CHECK(ig_neighbours_draw_nuisance(nb, key.data(), &pos, &hg, &gz, frags.data(), 200, 5, skip, out.data(), idm.data(), g.data(), u.data()) == 0);
            for (int i = 0; i < 200; i++) CHECK(idm[i] >= 0 && idm[i] < 4 && u[i] >= 0.0 && u[i] < 1.0 && g[i] == g[i]);
        }
    }
    /* the error paths */
    const int before = g_fails;
    int32_t bad = N, one = 0, o5[16];
    CHECK(ig_neighbours_draw(nb, key.data(), &pos, &bad, 1, 5, o5) == -1);
    CHECK(ig_neighbours_draw(nb, key.data(), &pos, &one, 1, 0, o5) == -1);
    CHECK(ig_neighbours_draw(nb, key.data(), &pos, &one, 1, 17, o5) == -1);
    int32_t badpos = 700;
    CHECK(ig_neighbours_draw(nb, key.data(), &badpos, &one, 1, 5, o5) == -1);
    CHECK(ig_neighbours_draw(nullptr, key.data(), &pos, &one, 1, 5, o5) == -1);
    ig_neighbours* nb2 = nullptr;
    std::vector<int64_t> ip2 = {0, 2, 1};
    CHECK(ig_neighbours_create(ip2.data(), xk.data(), pk.data(), 2, nullptr, 0, &nb2) == -1);
    int32_t oob = N + 5;
    std::vector<int64_t> ip3 = {0, 1};
    float w1 = 1.0f;
    CHECK(ig_neighbours_create(ip3.data(), &oob, &w1, 1, nullptr, 0, &nb2) == -1);
    CHECK(g_fails == before + 7);
    ig_neighbours_destroy(nb);
    std::puts("draw harness ok");
    return 0;
}
