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

/* ================================================================== host side (one translation unit, five parts) */
#include "ig_host_core.inc"
#include "ig_host_upload.inc"
#include "ig_host_batch.inc"
#include "ig_host_nuis.inc"
#include "ig_host_debug.inc"
