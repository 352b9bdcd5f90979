/*
 * mi_dmrecon_debug.h -- test and development hooks of libmi_dmrecon.so.
 *
 * NOT part of the drop-in boundary (include/mi_dmrecon.h is; nothing here replaces anything of the reference): these six
 * entry points exist for tests/ and tools/ -- fault injection, the host-only halves of the planning (so that they can be
 * checked against the oracle without a GPU), pool introspection and the probe buffer of -DMI_PROBE builds.  They are
 * declared here so that the library exports nothing that no header declares (the link uses a version script: only
 * mi_dmrecon_* symbols leave the library, tests/test_abi_and_host.py checks that every one of them is declared in one of
 * the two headers).  A production caller never needs them; they may change without notice.
 */
#ifndef MI_DMRECON_DEBUG_H
#define MI_DMRECON_DEBUG_H

#include "mi_dmrecon.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The reference view with this id gets a non-positive pixel footprint in the reconstruct calls that follow (-1: none): the
 * path of patch_sampler.cc:78-82 (std::out_of_range -> that view's MI_DMRECON_EFOOTPRINT), which no valid camera reaches. */
void mi_dmrecon_debug_inject_footprint(int view_id);

/* The block map of a front launch with teams for n_views views on a device of n_cus compute units (`want`: the largest
 * team allowed; empty: per view the pixels not filled at the hand-over, or NULL): map_out gets min(grid, cap) words
 * view | member << 16 | team size << 24 (0xFFFFFFFF: a block without work).  A pure function: needs no GPU. */
int mi_dmrecon_debug_front_teams(int32_t n_views, int32_t n_cus, int32_t want, const int64_t* empty, uint32_t* map_out, int32_t cap,
                                 int32_t* grid_out, int32_t* team_min_out, int32_t* team_max_out);

/* The HOST half of a reconstruct call's planning on cameras and features alone -- global view selection of reference view
 * `ref` (DMRecon::globalViewSelection, global_view_selection.cc:33-101; tables = 1: from the dense scene tables, 0: directly)
 * and, if n_seeds_out is given, the view's seeds (the host half of DMRecon::processFeatures, dmrecon.cc:243-296): exactly the
 * code a call runs, checked against the oracle without a GPU.  ms_out (optional): the time of `repeats` selections. */
int mi_dmrecon_debug_plan_views_host(int32_t n_views, const mi_dmrecon_camera* cams, const int32_t* widths, const int32_t* heights,
                                     int32_t n_feat, const float* pos, const int32_t* off, const int32_t* ids,
                                     const mi_dmrecon_settings* st, int32_t ref, int32_t tables, int32_t repeats,
                                     int32_t* ids_out, int32_t* n_out, double* ms_out,
                                     int32_t seed_cap, int32_t* seed_xy_out, float* seed_depth_out, int32_t* n_seeds_out);

/* The scratch sets of the context's scene that no call holds at the moment (return value) and the pixel capacity of the
 * largest of them. */
int mi_dmrecon_debug_scratch_sets(mi_dmrecon_ctx* ctx, long long* pixels_max);

/* Launches an empty one-lane kernel (`k_region_mark`) on the context's stream and waits for it: bench.py brackets every timed
 * region with one, so that a rocprofv3 kernel trace can be cut to the timed regions (tools/trace_regions.py). */
int mi_dmrecon_debug_region_mark(mi_dmrecon_ctx* ctx, int tag);

/* The debug buffer of -DMI_PROBE builds (tools/patch_probe.py): the first call allocates n 8-byte words on the device; later
 * calls copy up to n words out and clear the buffer. */
int mi_dmrecon_debug_buffer(unsigned long long* out, int n);

#ifdef __cplusplus
}
#endif
#endif
