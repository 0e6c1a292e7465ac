/*
 * ig_oracle_ops.c -- genome-state kernels of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Restates, one C function per CUDA kernel, the fragment-order operators of
 *   /root/reference/src/instagraal/kernels/kernel_sparse_adapt.cu  ("KA")
 * Each loop iteration `x` plays one CUDA thread (`id_frag`); the scalars a
 * kernel loads into __shared__ in thread 0 are read once before the loop.
 * Every function cites the KA lines it follows.
 */
#include "ig_oracle.h"
#include <string.h>

/* the 17 members in KA:40-58 order, as an indexable view */
#define NF 17
static inline int32_t** fp(frag* f) { return (int32_t**)f; }
static inline int32_t* const* cfp(const frag* f) { return (int32_t* const*)f; }

/* "copy thread": every kernel's fall-through branch (e.g. KA:912-938) */
static inline void cp(frag* o, const frag* i, int x)
{
    for (int k = 0; k < NF; k++) fp(o)[k][x] = cfp(i)[k][x];
    o->id[x] = x;
}

/* the per-thread register copy of one fragment (KA:786-807 and alike) */
typedef struct {
    int c, p, sp, L, SL, LB, lb, sl, circ, prev, next, sb, ori;
} fr;
static inline fr ld(const frag* f, int x)
{
    fr r;
    r.c = f->id_c[x];
    r.p = f->pos[x];
    r.sp = f->sub_pos[x];
    r.L = f->l_cont[x];
    r.SL = f->sub_l_cont[x];
    r.LB = f->l_cont_bp[x];
    r.lb = f->len_bp[x];
    r.sl = f->sub_len[x];
    r.circ = f->circ[x];
    r.prev = f->prev[x];
    r.next = f->next[x];
    r.sb = f->start_bp[x];
    r.ori = f->ori[x];
    return r;
}
static inline void set_len(frag* o, int x, int L, int SL, int LB)
{
    o->l_cont[x] = L;
    o->sub_l_cont[x] = SL;
    o->l_cont_bp[x] = LB;
}
static inline void set_coord(frag* o, int x, int p, int sp, int sb)
{
    o->pos[x] = p;
    o->sub_pos[x] = sp;
    o->start_bp[x] = sb;
}

/* KA:357-406.  Canonical order of the atomics: ascending thread index
 * (SURVEY Appendix D, Q1).  Block size 512 only shapes the arrival order. */
void igo_select_uniq_id_c(const frag* f, int32_t* list_uniq_id_c, int32_t* list_uniq_len, int32_t* counter, int n_frags)
{
    int n = counter[0];
    for (int x = 0; x < n_frags; x++) {
        if (f->pos[x] == 0) {
            list_uniq_id_c[n] = f->id_c[x];
            list_uniq_len[n] = f->l_cont[x];
            n++;
        }
    }
    counter[0] = n;
}

/* KA:470-482 */
void igo_make_old_2_new_id_c(const int32_t* list_uniq_id_c, int32_t* old_2_new, int n_contigs)
{
    for (int i = 0; i < n_contigs; i++) old_2_new[list_uniq_id_c[i]] = i;
}

/* KA:429-466 */
void igo_count_num(const int32_t* vals, int value, int32_t* counter, int n_values)
{
    int n = 0;
    for (int i = 0; i < n_values; i++) n += (vals[i] == value);
    counter[0] += n;
}

/* KA:4689-4692 (the only part of gl_update_pos that outlives the viewer):
 * int id_c_new = max_id - old_2_new[id_c]  with max_id a float. */
void igo_renumber_id_c(frag* f, const int32_t* old_2_new, int32_t* id_contigs, float max_id, int n_frags)
{
    for (int x = 0; x < n_frags; x++) {
        int id_c = f->id_c[x];
        int id_c_new = (int)(max_id - (float)old_2_new[id_c]);
        f->id_c[x] = id_c_new;
        id_contigs[x] = id_c_new;
    }
}

/* KA:409-426 (ori is NOT reset) */
void igo_explode_genome(frag* f, const int32_t* shuffle_order, int n_frags)
{
    for (int x = 0; x < n_frags; x++) {
        f->pos[x] = 0;
        f->start_bp[x] = 0;
        f->sub_pos[x] = 0;
        f->id_c[x] = shuffle_order[x];
        f->prev[x] = -1;
        f->next[x] = -1;
        f->l_cont[x] = 1;
        f->l_cont_bp[x] = f->len_bp[x];
        f->sub_l_cont[x] = f->sub_len[x];
    }
}

/* KA:4604-4626 */
void igo_simple_copy(frag* out, const frag* in, int n_frags)
{
    for (int x = 0; x < n_frags; x++) cp(out, in, x);
}

/* KA:4566-4591 */
void igo_copy_struct(frag* out, const frag* in, int32_t* id_contigs, int n_frags)
{
    for (int x = 0; x < n_frags; x++) {
        cp(out, in, x);
        id_contigs[x] = in->id_c[x];
    }
}

/* KA:612-670 */
void igo_flip_frag(frag* out, const frag* in, int id_f_flip, int n_frags)
{
    for (int x = 0; x < n_frags; x++) {
        cp(out, in, x);
        if (x == id_f_flip) out->ori[x] = in->ori[x] * -1;
    }
}

/* KA:737-1078 */
void igo_pop_out_frag(frag* out, const frag* in, int32_t* pop_id_contigs, int P, int max_id, int n_frags)
{
    const fr q = ld(in, P);
    for (int x = 0; x < n_frags; x++) {
        const fr a = ld(in, x);
        cp(out, in, x);
        pop_id_contigs[x] = a.c;
        if (q.L > 2 && a.c == q.c) { /* KA:808-911 */
            if (a.p < q.p) {
                out->prev[x] = (x == q.next && q.circ == 1) ? q.prev : a.prev;
                out->next[x] = (a.p == q.p - 1) ? q.next : a.next;
                set_len(out, x, a.L - 1, a.SL - q.sl, a.LB - q.lb);
            } else if (a.p == q.p) {
                goto popped;
            } else {
                set_coord(out, x, a.p - 1, a.sp - q.sl, a.sb - q.lb);
                out->prev[x] = (a.p == q.p + 1) ? q.prev : a.prev;
                out->next[x] = (x == q.prev && q.circ == 1) ? q.next : a.next;
                set_len(out, x, a.L - 1, a.SL - q.sl, a.LB - q.lb);
            }
        } else if (q.L == 2 && a.c == q.c) { /* KA:940-1022 */
            if (a.p < q.p) {
                out->circ[x] = 0;
                out->prev[x] = -1;
                out->next[x] = -1;
                set_len(out, x, a.L - 1, a.SL - q.sl, a.LB - q.lb);
            } else if (a.p == q.p) {
                goto popped;
            } else {
                set_coord(out, x, a.p - 1, a.sp - q.sl, a.sb - q.lb);
                out->circ[x] = 0;
                out->prev[x] = -1;
                out->next[x] = -1;
                set_len(out, x, a.L - 1, a.SL - q.sl, a.LB - q.lb);
            }
        }
        continue;
    popped: /* KA:848-873 / 968-993 */
        set_coord(out, x, 0, 0, 0);
        out->id_c[x] = max_id + 1;
        pop_id_contigs[x] = max_id + 1;
        out->circ[x] = 0;
        out->ori[x] = 1;
        out->prev[x] = -1;
        out->next[x] = -1;
        set_len(out, x, 1, a.sl, a.lb);
    }
}

/* KA:1081-1371  "split insert @ left" */
void igo_pop_in_frag_1(frag* out, const frag* in, int P, int I, int max_id, int ori_pop, int n_frags)
{
    const fr q = ld(in, P), i = ld(in, I);
    const int act = (in->activ[I] == 1) && (in->activ[P] == 1);
    for (int x = 0; x < n_frags; x++) {
        cp(out, in, x);
        if (!act) continue;
        const fr a = ld(in, x);
        if (x == P) { /* KA:1148-1174 */
            set_coord(out, x, 0, 0, 0);
            out->len_bp[x] = q.lb;
            out->sub_len[x] = q.sl;
            out->circ[x] = 0;
            out->ori[x] = ori_pop;
            out->prev[x] = -1;
            out->next[x] = I;
            if (i.circ == 0) {
                out->id_c[x] = max_id + 1;
                set_len(out, x, i.L - i.p + 1, i.SL - i.sp + q.sl, i.LB - i.sb + q.lb);
            } else {
                out->id_c[x] = i.c;
                set_len(out, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
            }
        } else if (a.c == i.c) {
            out->circ[x] = 0;
            if (i.circ == 0) { /* KA:1194-1257 */
                if (a.p < i.p) {
                    out->next[x] = (a.p == i.p - 1) ? -1 : a.next;
                    set_len(out, x, i.p, i.sp, i.sb);
                } else if (a.p == i.p) {
                    set_coord(out, x, 1, q.sl, q.lb);
                    out->id_c[x] = max_id + 1;
                    out->ori[x] = i.ori;
                    out->prev[x] = P;
                    out->next[x] = i.next;
                    set_len(out, x, i.L - i.p + 1, i.SL - i.sp + q.sl, i.LB - i.sb + q.lb);
                } else {
                    set_coord(out, x, a.p - i.p + 1, a.sp - i.sp + q.sl, a.sb - i.sb + q.lb);
                    out->id_c[x] = max_id + 1;
                    set_len(out, x, i.L - i.p + 1, i.SL - i.sp + q.sl, i.LB - i.sb + q.lb);
                }
            } else { /* KA:1258-1327 */
                set_len(out, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
                if (a.p < i.p) {
                    set_coord(out, x, i.L - i.p + a.p + 1, i.SL - i.sp + a.sp + q.sl, i.LB - i.sb + a.sb + q.lb);
                    out->next[x] = (a.p == i.p - 1) ? -1 : a.next;
                } else if (a.p == i.p) {
                    set_coord(out, x, 1, q.sl, q.lb);
                    out->len_bp[x] = i.lb;
                    out->sub_len[x] = i.sl;
                    out->ori[x] = i.ori;
                    out->prev[x] = P;
                    out->next[x] = i.next;
                } else {
                    set_coord(out, x, a.p - i.p + 1, a.sp - i.sp + q.sl, a.sb - i.sb + q.lb);
                    out->next[x] = (x == i.prev) ? -1 : a.next;
                }
            }
        }
    }
}

/* KA:1373-1686  "split insert @ right" */
void igo_pop_in_frag_2(frag* out, const frag* in, int P, int I, int max_id, int ori_pop, int n_frags)
{
    const fr q = ld(in, P), i = ld(in, I);
    const int act = (in->activ[I] == 1) && (in->activ[P] == 1);
    for (int x = 0; x < n_frags; x++) {
        cp(out, in, x);
        if (!act) continue;
        const fr a = ld(in, x);
        if (x == P) { /* KA:1439-1480 */
            out->id_c[x] = i.c;
            out->len_bp[x] = q.lb;
            out->sub_len[x] = q.sl;
            out->circ[x] = 0;
            out->ori[x] = ori_pop;
            out->prev[x] = I;
            out->next[x] = -1;
            if (i.circ == 0) {
                set_coord(out, x, i.p + 1, i.sp + i.sl, i.sb + i.lb);
                set_len(out, x, i.p + 2, i.sp + i.sl + q.sl, i.sb + i.lb + q.lb);
            } else {
                set_coord(out, x, (i.L - (i.p + 1)) + i.p + 1, (i.SL - (i.sp + i.sl)) + i.sp + i.sl,
                          (i.LB - (i.sb + i.lb)) + i.sb + i.lb);
                set_len(out, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
            }
        } else if (a.c == i.c) {
            out->circ[x] = 0;
            if (i.circ == 0) { /* KA:1499-1564 */
                if (a.p < i.p) {
                    set_len(out, x, i.p + 2, i.sp + i.sl + q.sl, i.sb + i.lb + q.lb);
                } else if (a.p == i.p) {
                    out->ori[x] = i.ori;
                    out->prev[x] = i.prev;
                    out->next[x] = P;
                    set_len(out, x, i.p + 2, i.sp + i.sl + q.sl, i.sb + i.lb + q.lb);
                } else {
                    set_coord(out, x, a.p - (i.p + 1), a.sp - (i.sp + i.sl), a.sb - (i.sb + i.lb));
                    out->id_c[x] = max_id + 1;
                    out->prev[x] = (a.p == i.p + 1) ? -1 : a.prev;
                    set_len(out, x, i.L - (i.p + 1), i.SL - (i.sp + i.sl), i.LB - (i.sb + i.lb));
                }
            } else { /* KA:1565-1641 */
                set_len(out, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
                if (a.p < i.p) {
                    set_coord(out, x, (i.L - (i.p + 1)) + a.p, (i.SL - (i.sp + i.sl)) + a.sp,
                              (i.LB - (i.sb + i.lb)) + a.sb);
                    out->prev[x] = (x == i.next) ? -1 : a.prev;
                } else if (a.p == i.p) {
                    set_coord(out, x, (i.L - (i.p + 1)) + i.p, (i.SL - (i.sp + i.sl)) + i.sp,
                              (i.LB - (i.sb + i.lb)) + i.sb);
                    out->len_bp[x] = i.lb;
                    out->sub_len[x] = i.sl;
                    out->prev[x] = i.prev;
                    out->next[x] = P;
                } else {
                    set_coord(out, x, a.p - (i.p + 1), a.sp - (i.sp + i.sl), a.sb - (i.sb + i.lb));
                    out->prev[x] = (a.p == i.p + 1) ? -1 : a.prev;
                }
            }
        }
    }
}

/* KA:1688-1905  "insert @ right of id_f_ins" */
void igo_pop_in_frag_3(frag* out, const frag* in, int P, int I, int max_id, int ori_pop, int n_frags)
{
    (void)max_id;
    const fr q = ld(in, P), i = ld(in, I);
    const int act = (in->activ[I] == 1) && (in->activ[P] == 1);
    for (int x = 0; x < n_frags; x++) {
        cp(out, in, x);
        if (!act) continue;
        const fr a = ld(in, x);
        if (x == P) { /* KA:1754-1772 */
            set_coord(out, x, i.p + 1, i.sp + i.sl, i.sb + i.lb);
            out->id_c[x] = i.c;
            out->len_bp[x] = q.lb;
            out->sub_len[x] = q.sl;
            out->circ[x] = i.circ;
            out->ori[x] = ori_pop;
            out->prev[x] = I;
            out->next[x] = i.next;
            set_len(out, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
        } else if (a.c == i.c) { /* KA:1790-1861 */
            out->circ[x] = i.circ;
            set_len(out, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
            if (a.p < i.p) {
                out->prev[x] = (x == i.next && i.circ == 1) ? P : a.prev;
            } else if (a.p == i.p) {
                out->ori[x] = i.ori;
                out->next[x] = P;
            } else {
                set_coord(out, x, a.p + 1, a.sp + q.sl, a.sb + q.lb);
                out->prev[x] = (a.p == i.p + 1) ? P : a.prev;
            }
        }
    }
}

/* KA:2979-3365 */
void igo_split_contig(frag* out, const frag* in, int32_t* split_id_contigs, int F, int upstream, int max_id, int n_frags)
{
    const fr c = ld(in, F);
    const int act = (in->activ[F] == 1) && (c.L > 1);
    for (int x = 0; x < n_frags; x++) {
        const fr a = ld(in, x);
        cp(out, in, x);
        split_id_contigs[x] = a.c;
        if (!act || a.c != c.c) continue;
        out->circ[x] = 0;
        if (c.circ == 0) {
            if (upstream == 1) { /* KA:3035-3101 */
                if (a.p < c.p) {
                    out->next[x] = (a.p == c.p - 1) ? -1 : a.next;
                    set_len(out, x, c.p, c.sp, c.sb);
                } else if (a.p == c.p) {
                    set_coord(out, x, 0, 0, 0);
                    out->id_c[x] = max_id + 1;
                    split_id_contigs[x] = max_id + 1;
                    out->len_bp[x] = c.lb;
                    out->sub_len[x] = c.sl;
                    out->prev[x] = -1;
                    out->next[x] = c.next;
                    set_len(out, x, c.L - c.p, c.SL - c.sp, c.LB - c.sb);
                } else {
                    set_coord(out, x, a.p - c.p, a.sp - c.sp, a.sb - c.sb);
                    out->id_c[x] = max_id + 1;
                    split_id_contigs[x] = max_id + 1;
                    set_len(out, x, c.L - c.p, c.SL - c.sp, c.LB - c.sb);
                }
            } else { /* KA:3102-3168 */
                if (a.p < c.p) {
                    set_len(out, x, c.p + 1, c.sp + c.sl, c.sb + c.lb);
                } else if (a.p == c.p) {
                    set_coord(out, x, c.p, c.sp, c.sb);
                    out->len_bp[x] = c.lb;
                    out->sub_len[x] = c.sl;
                    out->prev[x] = c.prev;
                    out->next[x] = -1;
                    set_len(out, x, c.p + 1, c.sp + c.sl, c.sb + c.lb);
                } else {
                    set_coord(out, x, a.p - (c.p + 1), a.sp - (c.sp + c.sl), a.sb - (c.sb + c.lb));
                    out->id_c[x] = max_id + 1;
                    split_id_contigs[x] = max_id + 1;
                    out->prev[x] = (a.p == c.p + 1) ? -1 : a.prev;
                    set_len(out, x, c.L - (c.p + 1), c.SL - (c.sp + c.sl), c.LB - (c.sb + c.lb));
                }
            }
        } else { /* circular contig: the ring is opened, id and lengths kept */
            set_len(out, x, c.L, c.SL, c.LB);
            if (upstream == 1) { /* KA:3171-3243 */
                if (a.p < c.p) {
                    set_coord(out, x, c.L - c.p + a.p, c.SL - c.sp + a.sp, c.LB - c.sb + a.sb);
                    out->next[x] = (a.p == c.p - 1) ? -1 : a.next;
                } else if (a.p == c.p) {
                    set_coord(out, x, 0, 0, 0);
                    out->len_bp[x] = c.lb;
                    out->sub_len[x] = c.sl;
                    out->prev[x] = -1;
                    out->next[x] = c.next;
                } else {
                    set_coord(out, x, a.p - c.p, a.sp - c.sp, a.sb - c.sb);
                    out->next[x] = (x == c.prev) ? -1 : a.next;
                }
            } else { /* KA:3244-3320 */
                if (a.p < c.p) {
                    set_coord(out, x, (c.L - (c.p + 1)) + a.p, (c.SL - (c.sp + c.sl)) + a.sp,
                              (c.LB - (c.sb + c.lb)) + a.sb);
                    out->prev[x] = (x == c.next) ? -1 : a.prev;
                } else if (a.p == c.p) {
                    set_coord(out, x, (c.L - (c.p + 1)) + a.p, (c.SL - (c.sp + c.sl)) + c.sp,
                              (c.LB - (c.sb + c.lb)) + c.sb);
                    out->len_bp[x] = c.lb;
                    out->sub_len[x] = c.sl;
                    out->prev[x] = c.prev;
                    out->next[x] = -1;
                } else {
                    set_coord(out, x, a.p - (c.p + 1), a.sp - (c.sp + c.sl), a.sb - (c.sb + c.lb));
                    out->prev[x] = (a.p == c.p + 1) ? -1 : a.prev;
                }
            }
        }
    }
}

/* KA:3367-3693.  When A and B share a contig without closing a ring, members
 * of that contig are NOT written (quirk Q13): `out` keeps what it held. */
void igo_paste_contigs(frag* out, const frag* in, int A, int B, int max_id, int n_frags)
{
    (void)max_id;
    const fr fa = ld(in, A), fb = ld(in, B);
    const int act = (in->activ[A] == 1) && (in->activ[B] == 1);
    for (int x = 0; x < n_frags; x++) {
        const fr a = ld(in, x);
        if (!act) {
            cp(out, in, x);
            continue;
        }
        if (fa.c != fb.c) {
            cp(out, in, x);
            if (a.c == fa.c) { /* KA:3454-3511 */
                out->circ[x] = 0;
                set_len(out, x, fa.L + fb.L, fa.SL + fb.SL, fa.LB + fb.LB);
                if (fa.p == 0) {
                    set_coord(out, x, fa.L - (a.p + 1), fa.SL - (a.sp + a.sl), fa.LB - (a.sb + a.lb));
                    out->ori[x] = a.ori * -1;
                    out->prev[x] = (a.p == fa.L - 1) ? -1 : a.next;
                    out->next[x] = (a.p == fa.p) ? B : a.prev;
                } else {
                    out->next[x] = (a.p == fa.p) ? B : a.next;
                }
            } else if (a.c == fb.c) { /* KA:3512-3566 */
                out->id_c[x] = fa.c;
                out->circ[x] = 0;
                set_len(out, x, fa.L + fb.L, fa.SL + fb.SL, fa.LB + fb.LB);
                if (fb.p == 0) {
                    set_coord(out, x, fa.L + a.p, fa.SL + a.sp, fa.LB + a.sb);
                    out->prev[x] = (a.p == fb.p) ? A : a.prev;
                } else {
                    set_coord(out, x, fa.L + (fb.L - (a.p + 1)), fa.SL + (fb.SL - (a.sp + a.sl)),
                              fa.LB + (fb.LB - (a.sb + a.lb)));
                    out->ori[x] = a.ori * -1;
                    out->prev[x] = (a.p == fb.p) ? A : a.next;
                    out->next[x] = (a.p == 0) ? -1 : a.prev;
                }
            }
        } else { /* KA:3588-3671 */
            if (a.c == fa.c) {
                if ((fa.p == 0) && (fb.p == fa.L - 1)) {
                    cp(out, in, x);
                    out->circ[x] = 1;
                    out->prev[x] = (a.p == fa.p) ? B : a.prev;
                    out->next[x] = (a.p == fa.L - 1) ? A : a.next;
                    set_len(out, x, fa.L, fa.SL, fa.LB);
                } else if ((fa.p == fa.L - 1) && (fb.p == 0)) {
                    cp(out, in, x);
                    out->circ[x] = 1;
                    out->prev[x] = (a.p == fb.p) ? A : a.prev;
                    out->next[x] = (a.p == fa.L - 1) ? B : a.next;
                    set_len(out, x, fa.L, fa.SL, fa.LB);
                }
                /* else: nothing written */
            } else {
                cp(out, in, x);
            }
        }
    }
}

/* KA:2124-2270.  list_valid_insert is written by global thread 0 only; the
 * cut positions are per-block shared copies of the same values. */
void igo_get_bounds(const frag* f, int P, int I, int32_t* list_valid_insert, const int32_t* list_bounds,
                    int32_t* id_f_cut_upstream, int32_t* id_f_cut_downstream, int n_bounds, int n_frags)
{
    int pos_up[IGO_N_TO_CUT], pos_down[IGO_N_TO_CUT];
    const int cP = f->id_c[P], cI = f->id_c[I];
    const int same = (cP == cI);
    const int pP = f->pos[P], pI = f->pos[I];
    const int LP = f->l_cont[P], LI = f->l_cont[I];
    const int ins_is_ext = (pI == 0) || (pI == (LI - 1));
    for (int i = 0; i < n_bounds; i++) {
        int up, down;
        if (i == 0) { /* KA:2161-2180 */
            if (same) {
                if (pI < pP - 1) {
                    up = pI + 1;
                    down = pP;
                } else if (pI > pP + 1) {
                    down = pI - 1;
                    up = pP;
                } else {
                    up = pP;
                    down = pP;
                }
            } else {
                up = pP;
                down = pP;
            }
        } else if (i < n_bounds - 1) { /* KA:2181-2184 */
            up = pP - list_bounds[i - 1] > 0 ? pP - list_bounds[i - 1] : 0;
            down = pP + list_bounds[i - 1] < LP - 1 ? pP + list_bounds[i - 1] : LP - 1;
        } else {
            up = 0;
            down = LP - 1;
        }
        if (same && (pI <= pP) && (pI >= up)) { /* KA:2190-2219 */
            pos_up[i] = -1;
            list_valid_insert[i * 2] = -1;
        } else {
            pos_up[i] = up;
            if (up == 0) {
                int size_extract = pP - up;
                if ((size_extract == 1) || ins_is_ext) {
                    list_valid_insert[i * 2] = -1;
                    pos_up[i] = -1;
                } else {
                    list_valid_insert[i * 2] = 1;
                }
            } else {
                list_valid_insert[i * 2] = 1;
            }
        }
        if (same && (((pI >= pP) && (pI <= down)) || (pI == (pP - 1)))) { /* KA:2221-2248 */
            pos_down[i] = -1;
            list_valid_insert[i * 2 + 1] = -1;
        } else {
            pos_down[i] = down;
            if (down == LP - 1) {
                int size_extract = down - pP;
                if ((size_extract == 1) || ins_is_ext) {
                    list_valid_insert[i * 2 + 1] = -1;
                    pos_down[i] = -1;
                } else {
                    list_valid_insert[i * 2 + 1] = 1;
                }
            } else {
                list_valid_insert[i * 2 + 1] = 1;
            }
        }
    }
    for (int x = 0; x < n_frags; x++) { /* KA:2255-2269 */
        if (f->id_c[x] == cP) {
            for (int i = 0; i < n_bounds; i++) {
                if (f->pos[x] == pos_down[i]) id_f_cut_downstream[i] = x;
                if (f->pos[x] == pos_up[i]) id_f_cut_upstream[i] = x;
            }
        }
    }
}

/* KA:2400-2721 */
void igo_extract_block(frag* out, const frag* in, int32_t* split_id_contigs, int A, const int32_t* list_id_f_cut_b, int id_fb,
                       int upstream, int max_id, int n_frags)
{
    const fr c = ld(in, A); /* c.* = contig of A and A's own coordinates */
    const int activ_a = in->activ[A];
    int activ_b = 0, size = 0, sub_size = 0, size_bp = 0;
    fr b;
    memset(&b, 0, sizeof b);
    const int Bc = list_id_f_cut_b[id_fb];
    if (Bc >= 0) { /* KA:2464-2486 */
        b = ld(in, Bc);
        activ_b = in->activ[Bc];
        if (upstream == 1) {
            size = c.p - b.p + 1;
            sub_size = c.sp - b.sp + c.sl;
            size_bp = c.sb - b.sb + c.lb;
        } else {
            size = b.p - c.p + 1;
            sub_size = b.sp - c.sp + b.sl;
            size_bp = b.sb - c.sb + b.lb;
        }
    }
    const int act = (activ_a == 1) && (activ_b == 1);
    const int lo_p = upstream == 1 ? b.p : c.p;   /* first position of the block */
    const int hi_p = upstream == 1 ? c.p : b.p;   /* last position of the block */
    const int lo_sp = upstream == 1 ? b.sp : c.sp, lo_sb = upstream == 1 ? b.sb : c.sb;
    /* link carried across the gap: what follows / precedes the block */
    const int gap_next = upstream == 1 ? c.next : b.next; /* KA:2526 / 2609 */
    const int gap_prev = upstream == 1 ? b.prev : c.prev; /* KA:2580 / 2663 */
    for (int x = 0; x < n_frags; x++) {
        const fr a = ld(in, x);
        cp(out, in, x);
        split_id_contigs[x] = a.c;
        if (!act || a.c != c.c) continue;
        if (a.p < lo_p) {
            out->circ[x] = c.circ;
            out->next[x] = (a.p == lo_p - 1) ? gap_next : a.next;
            set_len(out, x, c.L - size, c.SL - sub_size, c.LB - size_bp);
        } else if (a.p <= hi_p) {
            set_coord(out, x, a.p - lo_p, a.sp - lo_sp, a.sb - lo_sb);
            out->id_c[x] = max_id + 1;
            split_id_contigs[x] = max_id + 1;
            out->circ[x] = 0;
            out->prev[x] = (a.p == lo_p) ? -1 : a.prev;
            out->next[x] = (a.p == hi_p) ? -1 : a.next;
            set_len(out, x, size, sub_size, size_bp);
        } else {
            set_coord(out, x, a.p - size, a.sp - sub_size, a.sb - size_bp);
            out->circ[x] = c.circ;
            out->prev[x] = (a.p == hi_p + 1) ? gap_prev : a.prev;
            set_len(out, x, c.L - size, c.SL - sub_size, c.LB - size_bp);
        }
    }
}

/* KA:2724-2976 */
void igo_insert_block(frag* out, const frag* o, const frag* init, int P, int I, const int32_t* list_id_bounds,
                      const int32_t* list_valid_insert, int id_mutation, int id_bound, int upstream, int n_frags)
{
    const fr q = ld(o, P), i = ld(o, I);
    const int id_ext = list_id_bounds[id_bound];
    const int ok = (o->activ[I] == 1) && (o->activ[P] == 1) && (i.c != q.c) && (list_valid_insert[id_mutation] != -1);
    for (int x = 0; x < n_frags; x++) {
        if (!ok) { /* KA:2955-2975 */
            cp(out, init, x);
            continue;
        }
        const fr a = ld(o, x);
        cp(out, o, x);
        if (a.c == i.c) { /* KA:2802-2870 */
            out->circ[x] = i.circ;
            set_len(out, x, i.L + q.L, i.SL + q.SL, i.LB + q.LB);
            if (a.p < i.p) {
                out->prev[x] = (x == i.next && i.circ == 1) ? id_ext : a.prev;
            } else if (a.p == i.p) {
                out->ori[x] = i.ori;
                out->next[x] = P;
            } else {
                set_coord(out, x, a.p + q.L, a.sp + q.SL, a.sb + q.LB);
                out->prev[x] = (a.p == i.p + 1) ? id_ext : a.prev;
            }
        } else if (a.c == q.c) { /* KA:2871-2932 */
            out->id_c[x] = i.c;
            out->circ[x] = i.circ;
            set_len(out, x, i.L + q.L, i.SL + q.SL, i.LB + q.LB);
            if (upstream == 0) {
                set_coord(out, x, i.p + 1 + a.p, i.sp + i.sl + a.sp, i.sb + i.lb + a.sb);
                out->prev[x] = (a.p == 0) ? I : a.prev;
                out->next[x] = (a.p == a.L - 1) ? i.next : a.next;
            } else {
                set_coord(out, x, i.p + 1 + (q.L - a.p - 1), i.sp + i.sl + (q.SL - a.sp - a.sl),
                          i.sb + i.lb + (q.LB - a.sb - a.lb));
                out->ori[x] = a.ori * -1;
                out->prev[x] = (a.p == a.L - 1) ? I : a.next;
                out->next[x] = (a.p == 0) ? i.next : a.prev;
            }
        }
    }
}

/* KA:4492-4553 (single thread) */
void igo_extract_uniq_mutations(const frag* f, int frag_a, int frag_b, int32_t* list_uniq_mutations,
                                const int32_t* list_valid_insert, int32_t* n_uniq, int flip_eject)
{
    int n, start, j = 0;
    if (flip_eject == 1) {
        for (int k = 0; k < 4; k++) list_uniq_mutations[k] = k;
        start = 4;
        n = IGO_N_TMP_STRUCT;
    } else {
        list_uniq_mutations[0] = 2;
        list_uniq_mutations[1] = 3;
        start = 2;
        n = IGO_N_TMP_STRUCT - start;
    }
    const int len_ci = f->l_cont[frag_a], len_cj = f->l_cont[frag_b];
    if (len_cj == 1) {
        n -= 4;
    } else {
        for (int k = 0; k < 4; k++) list_uniq_mutations[start + k] = 4 + k;
        start += 4;
    }
    if (len_ci == 1) {
        n -= 4;
    } else {
        for (int k = 0; k < 4; k++) list_uniq_mutations[start + k] = 8 + k;
        start += 4;
    }
    for (int i = 12; i < IGO_N_TMP_STRUCT; i++) {
        if (list_valid_insert[i - 12] != -1) {
            list_uniq_mutations[start + j] = i;
            j += 1;
        } else {
            n -= 1;
        }
    }
    n_uniq[0] = n;
}
