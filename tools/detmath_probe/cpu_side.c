/* CPU side of the bit-exactness probe: compiled by gcc, NOT by hipcc's host clang. */
#include "../../include/ig_detmath.h"
#include <stddef.h>
void cpu_eval(const float* s, const float* stot, const int* ob, size_t n, ig_params p, const double* lgf,
              float* o_pow, float* o_exp, double* o_log10, float* o_r, float* o_rc, double* o_term, long long* o_q)
{
    for (size_t i = 0; i < n; i++) {
        o_pow[i] = ig_powf(s[i], p.slope, ig_tab());
        o_exp[i] = ig_expf(-s[i] * 0.01f, ig_tab());
        o_log10[i] = ig_log10((double)s[i], ig_tab());
        o_r[i] = ig_rippe(s[i], p, ig_tab());
        o_rc[i] = ig_rippe_circ(s[i], stot[i], p, ig_tab());
        double lg = ig_lgfact(ob[i] > 0 ? ob[i] : 1, lgf, ig_tab());
        o_term[i] = ig_pixel_term(o_r[i], o_rc[i], ob[i], lg, ig_tab());
        o_q[i] = ig_quantize(o_term[i]);
    }
}
