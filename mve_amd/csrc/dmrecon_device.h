/* Host-callable launchers of the kernels in dmrecon_device.hip (internal). */
#ifndef MI_DMRECON_DEVICE_H
#define MI_DMRECON_DEVICE_H

#include <hip/hip_runtime.h>
#include "dmrecon_types.h"

/*
 * The kernels that depend on the filter width (mvs::Settings::filterWidth: 3, 5 or 7 -> 9 / 25 / 49 samples per
 * patch) are compiled once per width (dmrecon_device.hip with -DMI_FW=...); mi_device_api() returns the launchers
 * of a width, or null.
 *
 * optimize -- lanes_per_view: 1 = 16 patches per wavefront (throughput), 16 = one patch per wavefront (latency).
 *   The launch only acts if min_work <= n < max_work, n = *n_work_ptr (if given) or n_work.
 *   follow_out != null: one optimisation attempt per entry; entries with further candidate hypotheses are appended to
 *   follow_out.  follow_in != null: continue the entries work[follow_in[..]] -- with ONE more attempt each if the launch has a
 *   follow_out of its own, else with all the attempts they have left.  A follow-up list is MI_FOLLOW_SEGS segments of follow_seg
 *   entries, one per XCD (workgroups b with equal b % 8), each with its own counter follow_*_n[segment] (OptArgs::follow_seg).
 * generate -- k_generate scans MI_GEN_TILE_W x MI_GEN_TILE_H pixel tiles; max_tiles = max over the jobs of their tile count.
 *   A view's entries go to `work` (count round_work[round]) while the view is in the throughput layout, to `work_lat`
 *   (round_work_lat[round]) once it has handed over: for good, from the round after the first one in which the VIEW's
 *   own list had fewer than `handover` entries -- decided on the device from view_count ([3][n_jobs], rotating per round;
 *   the caller presets slot 0 = what "the round before round 1" counts as, slot 1 = 0) and recorded in view_mode
 *   ([n_jobs], preset 0; the round of the hand-over).
 * tail -- one fused tail round: candidates from (prev_work, prev_results, round_work[round - 1]) -> this round's list,
 *   results and pixel-state writes (second state slot, see DevJob).  speculative: workgroups of four wavefronts, a
 *   pixel's candidate hypotheses tried at the same time; else one wavefront per pixel tries them in turn.
 * front -- the end of the tail, one persistent workgroup per reference view (k_front): the accepted entries of
 *   (list, list_results, *list_n) = round first_round - 1 are dealt out to per-view lists (view j's at job_off[j] of
 *   work0 / results0, its size in job_count[j], zeroed by the caller), then every view runs its own rounds
 *   first_round, first_round + 1, ... until its front is empty (lists alternate between the two buffer pairs).
 *   job_stats[4 j ..]: rounds run, attempts run, sum of list sizes, 100 MHz ticks (zeroed by the caller).
 *   Same maps as one `tail` launch per round.
 *   team > 1: that many workgroups per view (k_front<TEAM>): the attempts of a round are dealt out over them, every
 *   member runs the whole round otherwise (same numbering, same state writes) and fetches the others' results from the
 *   view's mailbox: mail = MI_FRONT_MAIL_WORDS 8-byte words per view, team_flags = MI_FRONT_FLAG_STRIDE words per view,
 *   zeroed by the caller; team_filled = one word per view, zeroed.  The members wait for each other at every pass, at
 *   most spin_ticks (100 MHz ticks): if one does not show up (all n_jobs * team workgroups must be resident at once; the
 *   caller keeps it <= the CUs, but cannot know what else holds them) the team GIVES UP -- error_flags bit 5 (32) is set
 *   and job_resume[j] (zeroed by the caller of a first launch) says from where view j goes on: round << 32 | entries << 1
 *   | list buffer, or >= MI_FRONT_DONE_HOST for a view that ran to its end.  A second launch with job_start = that array
 *   (no dealing out, the views start where it says) and team = 1 finishes them; same maps either way.
 */
#define MI_FOLLOW_SEGS 8            /* segments of a follow-up list = XCDs (MI_XCDS of dmrecon_device.hip) */
#define MI_FRONT_DONE_HOST 0xFFFFFFFF00000000ull
#define MI_FRONT_TEAM_MAX 32
#define MI_FRONT_FLAG_STRIDE 40     /* team_flags words per view: MI_FRONT_TEAM_MAX pass flags, the XCC registration word, "several XCDs" */
#define MI_FRONT_MAIL_WORDS (2 * 1024 * 12)
struct MiDeviceApi {
    int filter_width;
    void (*optimize)(hipStream_t s, int lanes_per_view, unsigned grid_blocks, const DevJob* jobs, const DevView* views,
                     const float* lut, const DevSettings& st, const DevEntry* work, const DevHyp* hyp,
                     DevResult* results, const unsigned* n_work_ptr, unsigned n_work, unsigned min_work,
                     unsigned max_work, int round, DevCounters* counters, const unsigned* follow_in,
                     const unsigned* follow_in_n, unsigned* follow_out, unsigned* follow_out_n, unsigned follow_seg,
                     unsigned follow_seg_in /* the segments of follow_in (0: one list) */,
                     unsigned long long* follow_mask /* first launch of a large round: a mask per wavefront unit instead of
                                                      * appending (OptArgs::follow_mask; mi_launch_follow_compact makes the list) */);
    void (*patch_eval)(hipStream_t s, const DevJob* job, const DevView* views, const float* lut,
                       const DevSettings& st, int x, int y, float depth, float dzI, float dzJ,
                       float* master, float* ncc, int32_t* ok, float* col, float* deriv, int32_t* level);
    void (*generate)(hipStream_t s, const DevJob* jobs, int n_jobs, int max_tiles, DevEntry* work, DevEntry* work_lat,
                     unsigned* round_work, unsigned* round_work_lat, unsigned* view_count, unsigned* view_mode,
                     unsigned handover, int round, unsigned* items, unsigned* round_items,
                     int self_round /* 1: the entries are the pixels written in round - 1 themselves (the seed re-optimisation
                                     * round, DevSettings::self_round), not their neighbours */);
    void (*tail)(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevView* views, const float* lut,
                 const DevSettings& st, const DevEntry* prev_work, const DevResult* prev_results, DevEntry* work,
                 DevResult* results, unsigned* round_work, int round, DevCounters* counters, bool speculative);
    void (*front)(hipStream_t s, int n_jobs, const DevJob* jobs, const DevView* views, const float* lut, const DevSettings& st,
                  const DevEntry* list, const DevResult* list_results, const unsigned* list_n,
                  DevEntry* work0, DevResult* results0, DevEntry* work1, DevResult* results1,
                  const unsigned* job_off, unsigned* job_count, unsigned* job_stats, int first_round, int max_rounds,
                  DevCounters* counters, int team, unsigned long long* mail, unsigned* team_flags,
                  const unsigned long long* job_start, unsigned long long* job_resume, unsigned* team_filled, unsigned spin_ticks,
                  int fault /* test hook: member | round << 8 of the team member that vanishes (member 255: none), bit 24: the
                             * team writes through its L2s as if found on several XCDs; -1 = none */,
                  int n_xcd /* teams: XCDs of the device; a view's team is confined to the blocks of one (b % n_xcd) */,
                  unsigned* host_done /* page-locked host memory, n_jobs zeroed words, or null: set to 1 per view that has run to its
                                       * end, after its state has been written back to memory */,
                  const unsigned* block_map /* teams: per block of the grid job | member << 16 | team size << 24, 0xFFFFFFFF = none;
                                             * the blocks of a team share b % n_xcd */,
                  unsigned grid_blocks,
                  const unsigned* job_order /* one workgroup per view: the view every block runs (a permutation of 0 .. n_jobs - 1: the
                                             * views the host expects to run longest first), or null: block b runs view b */);
    /* optimize_spec -- a round of the throughput layout with every (entry, candidate rank) pair on a quad of its own: `items`
     * (entry << 2 | rank, *n_items of them: written by `generate` when given an item list) are the attempts, spec holds
     * one record per item; mi_launch_apply_spec applies the reference's sequential rule to the records and writes the
     * pixels back.  Same maps and counters as `optimize` + mi_launch_apply. */
    void (*optimize_spec)(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevView* views, const float* lut,
                          const DevSettings& st, const DevEntry* work, DevSpec* spec, const unsigned* items, const unsigned* n_items,
                          const unsigned* n_work_ptr, unsigned n_work,
                          unsigned min_work, unsigned max_work, int round, DevCounters* counters);
};
const MiDeviceApi* mi_device_api(int filter_width);
extern unsigned long long* mi_debug_tbuf;

/* ---- kernels that do not depend on the filter width (defined once, in the width-5 object) ---- */
#define MI_GEN_TILE_W 64
#define MI_GEN_TILE_H 32
/* (both act only if min_work <= n < max_work, n = *n_work_ptr if given) */
void mi_launch_apply(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevEntry* work, const DevResult* results,
                     const unsigned* n_work_ptr, unsigned n_work, unsigned min_work, unsigned max_work, int round, DevCounters* counters);
void mi_launch_apply_spec(hipStream_t s, unsigned grid_blocks, const DevJob* jobs, const DevEntry* work, const DevSpec* spec,
                          const unsigned* items, const unsigned* n_items, const unsigned* n_work_ptr, unsigned n_work, unsigned min_work, unsigned max_work, int round, DevCounters* counters);
/* maps: [slot 0: depth | conf | dz x2 | normal x3][slot 1: same], imaps: [views | upd][views1 | upd1], per batch */
/* eight_views: imaps also holds [views_hi | views1_hi] behind them (nrReconNeighbors > 4) */
/* ... for the pixels [first, first + count) of the batch */
void mi_launch_flatten(hipStream_t s, float* maps, uint32_t* imaps, size_t total_px, bool eight_views, size_t first, size_t count);
/* the pixels [first, first + count) of the batch that were written from round r0 on, appended as records of `stride` words
 * ({view, pixel - first, depth, conf, dzI, dzJ [, nx, ny, nz]}) to `out` (page-locked host memory, `cap` records; *d_count counts
 * them all, also those that did not fit) */
void mi_launch_emit_changed(hipStream_t s, const float* maps, const uint32_t* imaps, size_t total_px, size_t first, size_t count,
                            int r0, unsigned view, unsigned stride, unsigned* d_count, unsigned cap, uint32_t* out);
/* seed_reopt: the records come from the re-optimising seed launch (DevSettings::seed_reopt: first confidence in `accepted`,
 * "propagates" in `tried`) -- the pixels are stamped round 1 / round 0 accordingly, and seed_count[job] (zeroed by the caller)
 * counts the pixels written: the propagation then starts with round 2 */
void mi_launch_apply_seeds(hipStream_t s, const DevJob* jobs, const DevEntry* work, const DevResult* results,
                           unsigned n_work, DevCounters* counters, unsigned long long* seed_keys,
                           const unsigned* key_off, int seed_reopt = 0, unsigned* seed_count = nullptr);
/* one dispatch instead of four small copies: a[0..n_a) | b[0..n_b) -> out_rw, *counters -> *out_hc, per job (flags, n_filled,
 * view_count[job] or 0) -> out_dyn (12 bytes per job); the out pointers are page-locked host memory */
void mi_launch_round_report(hipStream_t s, const unsigned* a, int n_a, const unsigned* b, int n_b, const DevCounters* counters,
                            const DevJob* jobs, int n_jobs, unsigned* out_rw, DevCounters* out_hc, void* out_dyn, const unsigned* view_count);
/* the job records of a batch from their packed upload (words_per_job 32-bit words each: the part of DevJob in use, the
 * same for every job of the batch) to their places in jobs[] */
void mi_launch_unpack_jobs(hipStream_t s, const uint32_t* packed, unsigned words_per_job, DevJob* jobs, int n_jobs);
/* an empty one-lane kernel: where profiles are cut (mi_dmrecon_debug_region_mark) */
void mi_launch_region_mark(hipStream_t s, unsigned tag);
unsigned mi_quad_words(void);     /* 32-bit words per footprint element (2: column pairs; 4 / 12 in the -DMI_QUAD_RECORDS / -DMI_EMU_LIN48 experiment builds) */
/* the first follow-up list of a round from the masks of its first launch, in list order (k_follow_count + k_follow_scatter);
 * blk_sum: 1024 words of scratch; acts like the launches of the round only if min_work <= *n_work_ptr < max_work */
void mi_launch_follow_compact(hipStream_t s, const unsigned long long* mask, const unsigned* n_work_ptr, unsigned min_work, unsigned max_work,
                              unsigned ppw, unsigned lpp, unsigned* blk_sum, unsigned* out, unsigned* out_n);
/* dst: w*h records of 16 bytes (texels (x,y) (x+1,y) (x,y+1) (x+1,y+1), edge-clamped) */
void mi_launch_quadify(hipStream_t s, const uint32_t* src, uint32_t* dst, int w, int h);
void mi_launch_pack_rgba(hipStream_t s, const uint8_t* src, uint32_t* dst, int n, int channels);
void mi_launch_unpack_rgb(hipStream_t s, const uint32_t* src, uint8_t* dst, int n);
void mi_launch_pyramid(hipStream_t s, const uint32_t* src, uint32_t* dst, int iw, int ih, int ow, int oh,
                       float w1, float w2, float w3);
#endif
