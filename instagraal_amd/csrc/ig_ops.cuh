/*
 * ig_ops.cuh -- fragment-order operators on a LOCAL window (the contigs of A and B only).
 *
 * The reference rewrites the whole genome (17 x N ints) once per operator and per candidate
 * (kernel_sparse_adapt.cu "KA" :612-3693, ~50 launches x 136 N bytes per candidate).  Every
 * operator only changes fragments of the contigs of its two reference fragments, so here a
 * workgroup owns one candidate genome restricted to those contigs and applies the operator
 * chain IN PLACE: reference scalars are read by all lanes, barrier, each lane rewrites the
 * fragments it owns, barrier.  Local index = rank in [contig(A) by pos | contig(B) by pos].
 *
 * Field semantics follow KA exactly (line ranges cited per operator); contig ids are internal
 * (fresh ids come from a counter instead of max+1: only equality is ever observed).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace igd {

/* one candidate genome restricted to the touched contigs; arrays of length n */
struct Loc {
    int* pos;
    int* spos;
    int* cid;
    int* sbp;
    int* circ;
    int* prev;
    int* next;
    int* L;
    int* SL;
    int* LB;
    int* ori;
    const int* gid; /* global fragment id of local slot */
    const int* lb;  /* len_bp  (constant) */
    const int* sl;  /* sub_len (constant) */
    int n;
};

struct FR {
    int c, p, sp, L, SL, LB, lb, sl, circ, prev, next, sb, ori;
};

__device__ __forceinline__ FR ld(const Loc& S, int x)
{
    FR r;
    r.c = S.cid[x];
    r.p = S.pos[x];
    r.sp = S.spos[x];
    r.L = S.L[x];
    r.SL = S.SL[x];
    r.LB = S.LB[x];
    r.lb = S.lb[x];
    r.sl = S.sl[x];
    r.circ = S.circ[x];
    r.prev = S.prev[x];
    r.next = S.next[x];
    r.sb = S.sbp[x];
    r.ori = S.ori[x];
    return r;
}
__device__ __forceinline__ void set_len(const Loc& S, int x, int L, int SL, int LB)
{
    S.L[x] = L;
    S.SL[x] = SL;
    S.LB[x] = LB;
}
__device__ __forceinline__ void set_coord(const Loc& S, int x, int p, int sp, int sb)
{
    S.pos[x] = p;
    S.spos[x] = sp;
    S.sbp[x] = sb;
}

#define IG_FOR_OWN(x) for (int x = threadIdx.x; x < S.n; x += blockDim.x)

/* KA:612-670 */
__device__ inline void op_flip(const Loc& S, int F)
{
    if (threadIdx.x == 0) S.ori[F] = S.ori[F] * -1;
    __syncthreads();
}

/* KA:737-1078.  P is a local index; `fresh` the id of the singleton contig P becomes. */
__device__ inline void op_pop_out(const Loc& S, int P, int fresh)
{
    const FR q = ld(S, P);
    const int gP = S.gid[P];
    (void)gP;
    __syncthreads();
    if (q.L >= 2) {
        IG_FOR_OWN(x)
        {
            const FR a = ld(S, x);
            if (a.c != q.c) continue;
            if (x == P) {
                set_coord(S, x, 0, 0, 0);
                S.cid[x] = fresh;
                S.circ[x] = 0;
                S.ori[x] = 1;
                S.prev[x] = -1;
                S.next[x] = -1;
                set_len(S, x, 1, a.sl, a.lb);
            } else if (q.L > 2) {
                const int g = S.gid[x];
                if (a.p < q.p) {
                    S.prev[x] = (g == q.next && q.circ == 1) ? q.prev : a.prev;
                    S.next[x] = (a.p == q.p - 1) ? q.next : a.next;
                } else {
                    set_coord(S, x, a.p - 1, a.sp - q.sl, a.sb - q.lb);
                    S.prev[x] = (a.p == q.p + 1) ? q.prev : a.prev;
                    S.next[x] = (g == q.prev && q.circ == 1) ? q.next : a.next;
                }
                set_len(S, x, a.L - 1, a.SL - q.sl, a.LB - q.lb);
            } else { /* q.L == 2: the survivor becomes a linear singleton */
                if (a.p > q.p) set_coord(S, x, a.p - 1, a.sp - q.sl, a.sb - q.lb);
                S.circ[x] = 0;
                S.prev[x] = -1;
                S.next[x] = -1;
                set_len(S, x, a.L - 1, a.SL - q.sl, a.LB - q.lb);
            }
        }
    }
    __syncthreads();
}

/* KA:1081-1371 "split insert @ left" */
__device__ inline void op_pop_in_1(const Loc& S, int P, int I, int fresh, int ori_pop)
{
    const FR q = ld(S, P), i = ld(S, I);
    const int gP = S.gid[P], gI = S.gid[I];
    __syncthreads();
    IG_FOR_OWN(x)
    {
        const FR a = ld(S, x);
        if (x == P) {
            set_coord(S, x, 0, 0, 0);
            S.circ[x] = 0;
            S.ori[x] = ori_pop;
            S.prev[x] = -1;
            S.next[x] = gI;
            if (i.circ == 0) {
                S.cid[x] = fresh;
                set_len(S, x, i.L - i.p + 1, i.SL - i.sp + q.sl, i.LB - i.sb + q.lb);
            } else {
                S.cid[x] = i.c;
                set_len(S, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
            }
        } else if (a.c == i.c) {
            S.circ[x] = 0;
            if (i.circ == 0) {
                if (a.p < i.p) {
                    S.next[x] = (a.p == i.p - 1) ? -1 : a.next;
                    set_len(S, x, i.p, i.sp, i.sb);
                } else if (a.p == i.p) {
                    set_coord(S, x, 1, q.sl, q.lb);
                    S.cid[x] = fresh;
                    S.prev[x] = gP;
                    set_len(S, x, i.L - i.p + 1, i.SL - i.sp + q.sl, i.LB - i.sb + q.lb);
                } else {
                    set_coord(S, x, a.p - i.p + 1, a.sp - i.sp + q.sl, a.sb - i.sb + q.lb);
                    S.cid[x] = fresh;
                    set_len(S, x, i.L - i.p + 1, i.SL - i.sp + q.sl, i.LB - i.sb + q.lb);
                }
            } else {
                set_len(S, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
                if (a.p < i.p) {
                    set_coord(S, x, i.L - i.p + a.p + 1, i.SL - i.sp + a.sp + q.sl, i.LB - i.sb + a.sb + q.lb);
                    S.next[x] = (a.p == i.p - 1) ? -1 : a.next;
                } else if (a.p == i.p) {
                    set_coord(S, x, 1, q.sl, q.lb);
                    S.prev[x] = gP;
                } else {
                    set_coord(S, x, a.p - i.p + 1, a.sp - i.sp + q.sl, a.sb - i.sb + q.lb);
                    S.next[x] = (S.gid[x] == i.prev) ? -1 : a.next;
                }
            }
        }
    }
    __syncthreads();
}

/* KA:1373-1686 "split insert @ right" */
__device__ inline void op_pop_in_2(const Loc& S, int P, int I, int fresh, int ori_pop)
{
    const FR q = ld(S, P), i = ld(S, I);
    const int gP = S.gid[P], gI = S.gid[I];
    __syncthreads();
    IG_FOR_OWN(x)
    {
        const FR a = ld(S, x);
        if (x == P) {
            S.cid[x] = i.c;
            S.circ[x] = 0;
            S.ori[x] = ori_pop;
            S.prev[x] = gI;
            S.next[x] = -1;
            if (i.circ == 0) {
                set_coord(S, x, i.p + 1, i.sp + i.sl, i.sb + i.lb);
                set_len(S, x, i.p + 2, i.sp + i.sl + q.sl, i.sb + i.lb + q.lb);
            } else {
                set_coord(S, x, i.L, i.SL, i.LB);
                set_len(S, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
            }
        } else if (a.c == i.c) {
            S.circ[x] = 0;
            if (i.circ == 0) {
                if (a.p < i.p) {
                    set_len(S, x, i.p + 2, i.sp + i.sl + q.sl, i.sb + i.lb + q.lb);
                } else if (a.p == i.p) {
                    S.next[x] = gP;
                    set_len(S, x, i.p + 2, i.sp + i.sl + q.sl, i.sb + i.lb + q.lb);
                } else {
                    set_coord(S, x, a.p - (i.p + 1), a.sp - (i.sp + i.sl), a.sb - (i.sb + i.lb));
                    S.cid[x] = fresh;
                    S.prev[x] = (a.p == i.p + 1) ? -1 : a.prev;
                    set_len(S, x, i.L - (i.p + 1), i.SL - (i.sp + i.sl), i.LB - (i.sb + i.lb));
                }
            } else {
                set_len(S, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
                if (a.p < i.p) {
                    set_coord(S, x, (i.L - (i.p + 1)) + a.p, (i.SL - (i.sp + i.sl)) + a.sp, (i.LB - (i.sb + i.lb)) + a.sb);
                    S.prev[x] = (S.gid[x] == i.next) ? -1 : a.prev;
                } else if (a.p == i.p) {
                    set_coord(S, x, (i.L - (i.p + 1)) + i.p, (i.SL - (i.sp + i.sl)) + i.sp, (i.LB - (i.sb + i.lb)) + i.sb);
                    S.next[x] = gP;
                } else {
                    set_coord(S, x, a.p - (i.p + 1), a.sp - (i.sp + i.sl), a.sb - (i.sb + i.lb));
                    S.prev[x] = (a.p == i.p + 1) ? -1 : a.prev;
                }
            }
        }
    }
    __syncthreads();
}

/* KA:1688-1905 "insert @ right of I" */
__device__ inline void op_pop_in_3(const Loc& S, int P, int I, int ori_pop)
{
    const FR q = ld(S, P), i = ld(S, I);
    const int gP = S.gid[P], gI = S.gid[I];
    __syncthreads();
    IG_FOR_OWN(x)
    {
        const FR a = ld(S, x);
        if (x == P) {
            set_coord(S, x, i.p + 1, i.sp + i.sl, i.sb + i.lb);
            S.cid[x] = i.c;
            S.circ[x] = i.circ;
            S.ori[x] = ori_pop;
            S.prev[x] = gI;
            S.next[x] = i.next;
            set_len(S, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
        } else if (a.c == i.c) {
            S.circ[x] = i.circ;
            set_len(S, x, i.L + 1, i.SL + q.sl, i.LB + q.lb);
            if (a.p < i.p) {
                S.prev[x] = (S.gid[x] == i.next && i.circ == 1) ? gP : a.prev;
            } else if (a.p == i.p) {
                S.next[x] = gP;
            } else {
                set_coord(S, x, a.p + 1, a.sp + q.sl, a.sb + q.lb);
                S.prev[x] = (a.p == i.p + 1) ? gP : a.prev;
            }
        }
    }
    __syncthreads();
}

/* KA:2979-3365 */
__device__ inline void op_split(const Loc& S, int F, int upstream, int fresh)
{
    const FR c = ld(S, F);
    __syncthreads();
    if (c.L > 1) {
        IG_FOR_OWN(x)
        {
            const FR a = ld(S, x);
            if (a.c != c.c) continue;
            S.circ[x] = 0;
            if (c.circ == 0) {
                if (upstream == 1) {
                    if (a.p < c.p) {
                        S.next[x] = (a.p == c.p - 1) ? -1 : a.next;
                        set_len(S, x, c.p, c.sp, c.sb);
                    } else {
                        set_coord(S, x, a.p - c.p, a.sp - c.sp, a.sb - c.sb);
                        S.cid[x] = fresh;
                        if (a.p == c.p) S.prev[x] = -1;
                        set_len(S, x, c.L - c.p, c.SL - c.sp, c.LB - c.sb);
                    }
                } else {
                    if (a.p <= c.p) {
                        if (a.p == c.p) S.next[x] = -1;
                        set_len(S, x, c.p + 1, c.sp + c.sl, c.sb + c.lb);
                    } else {
                        set_coord(S, x, a.p - (c.p + 1), a.sp - (c.sp + c.sl), a.sb - (c.sb + c.lb));
                        S.cid[x] = fresh;
                        S.prev[x] = (a.p == c.p + 1) ? -1 : a.prev;
                        set_len(S, x, c.L - (c.p + 1), c.SL - (c.sp + c.sl), c.LB - (c.sb + c.lb));
                    }
                }
            } else { /* ring opened; id and lengths kept */
                const int g = S.gid[x];
                if (upstream == 1) {
                    if (a.p < c.p) {
                        set_coord(S, x, c.L - c.p + a.p, c.SL - c.sp + a.sp, c.LB - c.sb + a.sb);
                        S.next[x] = (a.p == c.p - 1) ? -1 : a.next;
                    } else if (a.p == c.p) {
                        set_coord(S, x, 0, 0, 0);
                        S.prev[x] = -1;
                    } else {
                        set_coord(S, x, a.p - c.p, a.sp - c.sp, a.sb - c.sb);
                        S.next[x] = (g == c.prev) ? -1 : a.next;
                    }
                } else {
                    if (a.p < c.p) {
                        set_coord(S, x, (c.L - (c.p + 1)) + a.p, (c.SL - (c.sp + c.sl)) + a.sp, (c.LB - (c.sb + c.lb)) + a.sb);
                        S.prev[x] = (g == c.next) ? -1 : a.prev;
                    } else if (a.p == c.p) {
                        set_coord(S, x, (c.L - (c.p + 1)) + a.p, (c.SL - (c.sp + c.sl)) + c.sp, (c.LB - (c.sb + c.lb)) + c.sb);
                        S.next[x] = -1;
                    } else {
                        set_coord(S, x, a.p - (c.p + 1), a.sp - (c.sp + c.sl), a.sb - (c.sb + c.lb));
                        S.prev[x] = (a.p == c.p + 1) ? -1 : a.prev;
                    }
                }
            }
        }
    }
    __syncthreads();
}

/* KA:3367-3693.  (A and B in one contig without closing a ring writes nothing -- unreachable for A != B.) */
__device__ inline void op_paste(const Loc& S, int A, int B)
{
    const FR fa = ld(S, A), fb = ld(S, B);
    const int gA = S.gid[A], gB = S.gid[B];
    __syncthreads();
    IG_FOR_OWN(x)
    {
        const FR a = ld(S, x);
        if (fa.c != fb.c) {
            if (a.c == fa.c) {
                S.circ[x] = 0;
                set_len(S, x, fa.L + fb.L, fa.SL + fb.SL, fa.LB + fb.LB);
                if (fa.p == 0) {
                    set_coord(S, x, fa.L - (a.p + 1), fa.SL - (a.sp + a.sl), fa.LB - (a.sb + a.lb));
                    S.ori[x] = a.ori * -1;
                    S.prev[x] = (a.p == fa.L - 1) ? -1 : a.next;
                    S.next[x] = (a.p == fa.p) ? gB : a.prev;
                } else {
                    S.next[x] = (a.p == fa.p) ? gB : a.next;
                }
            } else if (a.c == fb.c) {
                S.cid[x] = fa.c;
                S.circ[x] = 0;
                set_len(S, x, fa.L + fb.L, fa.SL + fb.SL, fa.LB + fb.LB);
                if (fb.p == 0) {
                    set_coord(S, x, fa.L + a.p, fa.SL + a.sp, fa.LB + a.sb);
                    S.prev[x] = (a.p == fb.p) ? gA : a.prev;
                } else {
                    set_coord(S, x, fa.L + (fb.L - (a.p + 1)), fa.SL + (fb.SL - (a.sp + a.sl)),
                              fa.LB + (fb.LB - (a.sb + a.lb)));
                    S.ori[x] = a.ori * -1;
                    S.prev[x] = (a.p == fb.p) ? gA : a.next;
                    S.next[x] = (a.p == 0) ? -1 : a.prev;
                }
            }
        } else if (a.c == fa.c) {
            if ((fa.p == 0) && (fb.p == fa.L - 1)) {
                S.circ[x] = 1;
                S.prev[x] = (a.p == fa.p) ? gB : a.prev;
                S.next[x] = (a.p == fa.L - 1) ? gA : a.next;
            } else if ((fa.p == fa.L - 1) && (fb.p == 0)) {
                S.circ[x] = 1;
                S.prev[x] = (a.p == fb.p) ? gA : a.prev;
                S.next[x] = (a.p == fa.L - 1) ? gB : a.next;
            }
        }
    }
    __syncthreads();
}

/* KA:2400-2721.  A, Bc local indices (Bc < 0: no-op). */
__device__ inline void op_extract_block(const Loc& S, int A, int Bc, int upstream, int fresh)
{
    if (Bc < 0) {
        __syncthreads();
        return;
    }
    const FR c = ld(S, A), b = ld(S, Bc);
    __syncthreads();
    int size, sub_size, size_bp;
    if (upstream == 1) {
        size = c.p - b.p + 1;
        sub_size = c.sp - b.sp + c.sl;
        size_bp = c.sb - b.sb + c.lb;
    } else {
        size = b.p - c.p + 1;
        sub_size = b.sp - c.sp + b.sl;
        size_bp = b.sb - c.sb + b.lb;
    }
    const int lo_p = upstream == 1 ? b.p : c.p, hi_p = upstream == 1 ? c.p : b.p;
    const int lo_sp = upstream == 1 ? b.sp : c.sp, lo_sb = upstream == 1 ? b.sb : c.sb;
    const int gap_next = upstream == 1 ? c.next : b.next;
    const int gap_prev = upstream == 1 ? b.prev : c.prev;
    IG_FOR_OWN(x)
    {
        const FR a = ld(S, x);
        if (a.c != c.c) continue;
        if (a.p < lo_p) {
            S.circ[x] = c.circ;
            S.next[x] = (a.p == lo_p - 1) ? gap_next : a.next;
            set_len(S, x, c.L - size, c.SL - sub_size, c.LB - size_bp);
        } else if (a.p <= hi_p) {
            set_coord(S, x, a.p - lo_p, a.sp - lo_sp, a.sb - lo_sb);
            S.cid[x] = fresh;
            S.circ[x] = 0;
            S.prev[x] = (a.p == lo_p) ? -1 : a.prev;
            S.next[x] = (a.p == hi_p) ? -1 : a.next;
            set_len(S, x, size, sub_size, size_bp);
        } else {
            set_coord(S, x, a.p - size, a.sp - sub_size, a.sb - size_bp);
            S.circ[x] = c.circ;
            S.prev[x] = (a.p == hi_p + 1) ? gap_prev : a.prev;
            set_len(S, x, c.L - size, c.SL - sub_size, c.LB - size_bp);
        }
    }
    __syncthreads();
}

/* KA:2724-2976.  Returns without touching S when the insertion is not admissible; the caller
 * guarantees S still equals the current genome in that case (extraction was a no-op). */
__device__ inline void op_insert_block(const Loc& S, int P, int I, int g_ext, int valid, int upstream)
{
    const FR q = ld(S, P), i = ld(S, I);
    const int gP = S.gid[P], gI = S.gid[I];
    __syncthreads();
    const bool ok = (i.c != q.c) && (valid != -1);
    if (ok) {
        IG_FOR_OWN(x)
        {
            const FR a = ld(S, x);
            if (a.c == i.c) {
                S.circ[x] = i.circ;
                set_len(S, x, i.L + q.L, i.SL + q.SL, i.LB + q.LB);
                if (a.p < i.p) {
                    S.prev[x] = (S.gid[x] == i.next && i.circ == 1) ? g_ext : a.prev;
                } else if (a.p == i.p) {
                    S.next[x] = gP;
                } else {
                    set_coord(S, x, a.p + q.L, a.sp + q.SL, a.sb + q.LB);
                    S.prev[x] = (a.p == i.p + 1) ? g_ext : a.prev;
                }
            } else if (a.c == q.c) {
                S.cid[x] = i.c;
                S.circ[x] = i.circ;
                set_len(S, x, i.L + q.L, i.SL + q.SL, i.LB + q.LB);
                if (upstream == 0) {
                    set_coord(S, x, i.p + 1 + a.p, i.sp + i.sl + a.sp, i.sb + i.lb + a.sb);
                    S.prev[x] = (a.p == 0) ? gI : a.prev;
                    S.next[x] = (a.p == a.L - 1) ? i.next : a.next;
                } else {
                    set_coord(S, x, i.p + 1 + (q.L - a.p - 1), i.sp + i.sl + (q.SL - a.sp - a.sl),
                              i.sb + i.lb + (q.LB - a.sb - a.lb));
                    S.ori[x] = a.ori * -1;
                    S.prev[x] = (a.p == a.L - 1) ? gI : a.next;
                    S.next[x] = (a.p == 0) ? i.next : a.prev;
                }
            }
        }
    }
    __syncthreads();
}

}  // namespace igd
