/*
 * Internal POD layouts shared by the host pipeline (dmrecon_host.cpp) and the
 * device kernels (dmrecon_device.hip).  Not part of the public ABI.
 */
#ifndef MI_DMRECON_TYPES_H
#define MI_DMRECON_TYPES_H

#include <stdint.h>

#define MI_MAX_LEVELS 16
#define MI_MAX_GLOBAL 128       /* global views per reference view (Settings::globalVSMax): one bit each in the availability mask
                                 * (MI_AVAIL_WORDS 64-bit words per patch); a view set holds 8-bit indices into the list */
#define MI_AVAIL_WORDS (MI_MAX_GLOBAL / 64)
#define MI_MAX_LOCAL 16         /* local views per patch (Settings::nrReconNeighbors): four, eight or sixteen view slots */
#define MI_MAX_FW 7         /* filter widths 3, 5, 7: the device code is compiled once per width (dmrecon_device.hip) */
#define MI_PATCHES_PER_WAVE 16
#define MI_VIEW_NONE 0xFFu
#define MI_MAX_ROUNDS 8192      /* per-round work counters kept on the device */

/* One pyramid level of one view, as the sampler needs it (ImagePyramidLevel,
 * libs/dmrecon/image_pyramid.h:28-54; K = [ax 0 cx; 0 ay cy; 0 0 1]). */
struct DevLevel {
    float ax, ay, cx, cy;
    float inv0;              /* invproj[0] = 1/ax */
    int32_t w, h;
    uint32_t tex_off;        /* offset (in texels) of this level inside DevView::img */
};

/* One view resident in HBM (SingleView + its ImagePyramid). */
struct DevView {
    float cam_pos[3];
    float w2c[12];           /* rows of [R|t] */
    int32_t n_levels;
    const uint32_t* img;     /* all levels back to back, RGBA8 (A unused), row-major */
    const uint32_t* quad;    /* same levels as footprint elements: element (x, y) = texels (x,y) (x,y+1), 8 bytes, edge-clamped --
                              * the whole bilinear footprint of a sample = elements x, x + 1 of row y in ONE 16-byte gather
                              * (mi_quad_words() words per element: 2; the 16-byte records of rounds 2-5 with -DMI_QUAD_RECORDS) */
    DevLevel lv[MI_MAX_LEVELS];
};

/* A global neighbour view as a job sees it: everything a sampling pass needs before it knows the mip level, in one
 * record indexed by the view's position in the job's global list (no global_ids -> DevView indirection). */
struct DevJobView {
    /* H = R_n R_ref^T K_s^-1 (row-major; K_s^-1 of the reference level): the pixel (x + .5, y + .5, 1) of the reference
     * image as a direction in this view's camera frame; sc = R_n C_ref + t_n: the reference camera centre there.  With
     * them a patch sample projects as s_C + (t g) H (pixel) -- see NView in dmrecon_device.hip.  Computed in double on
     * the host, rounded once. */
    float H[9];
    float sc[3];
    float w2c_z[4];          /* third row of [R|t] (SingleView::footPrint of a world point: the view selection) */
    float inv0;              /* invproj[0] of level 0 (SingleView::footPrint) */
    int32_t maxl;            /* number of pyramid levels - 1 */
    int32_t view;            /* index into the DevView table (levels, texels) */
    float cam_pos[3];
    int32_t pad[2];
};

/* One reference view being reconstructed (one mvs::DMRecon instance).  The per-view records of its global views come LAST:
 * a batch uploads, per job, only the part of the struct its views use (BatchRun::upload: the bytes up to gv[max n_global]). */
struct DevJob {
    /* the two words the device writes and the host polls (copied back as a strided 8-byte column) */
    int32_t flags;                      /* MI_JOB_* */
    uint32_t n_filled;                  /* pixels that went from confidence 0 to > 0 (Progress::filled) */
    int32_t ref_view, scale, w, h;
    float inv_a, inv_c, inv_b, inv_d;   /* invproj at `scale`: x' = inv_a*x + inv_c, y' = inv_b*y + inv_d */
    float rot_t[9];                     /* R^T (camera -> world rotation) */
    float cam_pos[3];
    float w2c_z[4];                     /* third row of [R|t] of the reference view */
    float inv0_s;                       /* footPrintScaled factor */
    int32_t n_global;
    /* per-pixel state maps (device), zero = unfilled (single_view.cc:78-81) */
    float* depth;
    float* dz;        /* 2 ch */
    float* conf;
    float* normal;    /* 3 ch */
    uint32_t* views;  /* 4 x 8-bit indices into global_ids, MI_VIEW_NONE padded */
    int32_t* upd;     /* round in which the pixel was last written, -1 = never */
    uint32_t* views_hi;   /* view slots 4..7 of the set (nrReconNeighbors > 4 only, else null) */
    /* Sixteen view slots (nrReconNeighbors > 8; else all three null): slots 8..15 of a set live apart from the rest, two words
     * each -- per pixel, per entry of the round's list (what an optimisation found, k_apply / k_apply_seeds copy it to the
     * pixel), per explicit hypothesis (null: none propagated).  The last two are the batch's, the same in every job.  Such
     * views stay in the throughput layout: there is no second state slot of them. */
    uint32_t* views_x;
    uint32_t* results_x;
    const uint32_t* hyp_x;
    /* Second slot of the pixel state, used by the fused tail rounds only (k_tail): a write of round r goes to
     * the slot that does NOT hold the pixel's state as of the end of round r-1, so the optimisations of a round
     * keep reading the frozen state of the previous round without a separate write-back launch.  The state
     * of a pixel is the slot with the larger stamp; k_flatten folds slot 1 back into slot 0 at the end. */
    float* depth1;
    float* dz1;
    float* conf1;
    float* normal1;
    uint32_t* views1;
    int32_t* upd1;
    uint32_t* views1_hi;
    int32_t global_ids[MI_MAX_GLOBAL];  /* ascending view ids (GlobalViewSelection result) */
    DevJobView gv[MI_MAX_GLOBAL];       /* ... and what the sampler needs of each of them */
};

#define MI_JOB_EFOOTPRINT 1u   /* device: non-positive master footprint in this view (patch_sampler.cc:78-82 throws) */
#define MI_JOB_DEAD       2u   /* host: the view failed or was cancelled -- its entries are skipped from now on */

/* Settings as the kernels see them. */
struct DevSettings {
    float minNCC, minParallax, acceptNCC, minRefineDiff;
    int32_t maxIterations, K, useColorScale;
    /* Not a setting of the reference: 1 in the ONE round in which every pixel written by a seed is re-optimised from its own
     * converged state, as the reference does when it pops a seed (it pushes the seed's OWN pixel, dmrecon.cc:316-326, and
     * propagates from it only if that re-optimisation strictly raised its confidence, :365-398) -- MI_DMRECON_SEED_REOPT. */
    int32_t self_round;
    /* (not settings either) floats per patch of the view selection's NCC table in the throughput kernels' dynamic shared memory:
     * 64, or MI_MAX_GLOBAL when globalVSMax is above 64 */
    int32_t ncc_stride;
    /* (nor this) 1 in the launch of the SEEDS when every seed that succeeds is re-optimised from its own converged state in the
     * same launch -- what the reference does when it pops the seed (dmrecon.cc:365-398) -- and propagates only if that strictly
     * raised its confidence (k_optimize<..., SEED>; the default form of the reference's seed semantics) */
    int32_t seed_reopt;
};

/* Work list entry + result of one patch optimisation attempt chain. */
struct DevEntry {
    int32_t job;
    int32_t xy;              /* x | y << 16 */
};
struct DevHyp {              /* explicit hypothesis (seeds / parity hook) */
    float depth, dzI, dzJ;
    uint32_t views;          /* packed global indices or all MI_VIEW_NONE */
    uint32_t views_hi;       /* ... of view slots 4..7 */
};
struct DevResult {
    float conf, depth, dzI, dzJ;
    float nx, ny, nz;
    uint32_t views;
    uint32_t views_hi;
    int32_t iters;
    int32_t accepted;        /* propagate mode: 1 if the pixel state must be overwritten */
    uint32_t tried;          /* propagate mode: neighbours (bit k: left, right, up, down) whose hypothesis has been consumed */
};

/* One candidate hypothesis of a work-list entry optimised SPECULATIVELY (small throughput rounds, k_optimize_spec): the
 * entry's up to four candidates run at the same time on four quads, the reference's sequential rule (dmrecon.cc:371,378,391)
 * is applied afterwards from these records (k_apply_spec).  One record per item of the round's item list (an entry's items
 * are consecutive, rank 0 first). */
struct DevSpec {
    float conf, depth, dzI, dzJ, nx, ny, nz;
    uint32_t views, views_hi;
    int32_t iters;
    float bc;                /* the candidate's source confidence: order of trial and pop-time test */
    float own;               /* the pixel's own confidence, frozen at the start of the round */
    uint32_t n_eval, n_pass; /* what this attempt counted (all view slots) */
    int32_t n_cand;          /* rank 0 only: candidates of the entry (0: the view has ended) */
    int32_t pad;             /* Run::deferred: what the attempt noted instead of doing (footprint exception, views replaced) */
};

struct DevCounters {
    unsigned long long n_patch, n_eval, n_pass, n_filled, n_seeds_ok;
    unsigned long long n_stage;        /* -DMI_ACTIVITY builds: turns of patch optimisations in the throughput layout ... */
    unsigned long long n_gather_pass;  /* ... and the turns their wavefronts ran x patches per wavefront (lane activity = the ratio) */
    unsigned long long n_view_replaced; /* local views dropped by replaceViews (patch_optimization.cc:218-228), attempts of all kinds */
    unsigned long long n_iter14;        /* ... of which only because they were still moving at iteration 14 */
    unsigned int error_flags;  /* bit0: non-positive master footprint (patch_sampler.cc:78-82) */
    unsigned int pad;
    /* The same work counts per KERNEL TEMPLATE (mi_dmrecon_stats::n_eval_by_kernel ...; MI_KIND_*): evaluations / passes / patches as
     * counted for n_eval / n_pass / n_patch (speculative forms: the attempts the reference's rule consumes), and the passes a
     * template EXECUTED (the attempts a speculative form runs and discards included) -- what a profile's per-kernel time is
     * divided by. */
    unsigned long long k_eval[8], k_pass[8], k_patch[8], k_pass_exec[8];
    /* shader clock while the patch kernels run: every 1024th wavefront of a k_optimize launch adds the shader cycles
     * (s_memtime) and the constant-rate ticks (s_memrealtime) of its own life: their ratio is the clock the kernels ran at */
    unsigned long long clk_shader, clk_real;
};
enum { MI_KIND_FAST = 0, MI_KIND_FOLLOW = 1, MI_KIND_SEED = 2, MI_KIND_LOOP = 3, MI_KIND_SPEC = 4, MI_KIND_LAT = 5, MI_KIND_TAIL = 6, MI_KIND_FRONT = 7 };

#endif
