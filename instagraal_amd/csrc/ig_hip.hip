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
 * Environment knobs (tests, fault injection and tuning only; the table in INTEGRATION.md section 4 is the reference): IG_BATCH_W, IG_WINDOW
 * (widths), IG_POOL_ENTRIES, IG_WIDE_LISTS, IG_NO_HOST_FLAG, IG_POISON_ALLOC / IG_POISON_ONLY (force the rare paths), IG_SCREEN,
 * IG_SCREEN_VERIFY, IG_ABLATE, IG_WINDOW_CHECK, IG_FUSED_COMMIT, IG_STEP_DRAW_FAST (the same results another way: what the
 * path-against-path tests and fuzzers toggle), IG_NUIS_W, IG_NUIS_SCREEN, IG_NUIS_HIST, IG_NUIS_SCREEN_VERIFY, IG_NUIS_SCREEN_NOCHECK,
 * IG_NUIS_HIST_TRACE, IG_NUIS_CHAIN, IG_NUIS_ASYNC (the nuisance step's tiers and its helper thread).
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

/* ================================================================== host side (one translation unit, five parts) */
#include "ig_host_core.inc"
#include "ig_host_upload.inc"
#include "ig_host_batch.inc"
#include "ig_host_nuis.inc"
#include "ig_host_debug.inc"
