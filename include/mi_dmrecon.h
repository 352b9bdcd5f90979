/*
 * mi_dmrecon.h -- C ABI of the MI355X-native depth-map reconstruction library
 * (libmi_dmrecon.so), the drop-in replacement for the compute inside MVE's
 * libs/dmrecon.  Paths below are relative to the reference tree
 * (simonfuhrmann/mve).
 *
 * The reference has no FFI: its boundary is the C++ class mvs::DMRecon
 * (libs/dmrecon/dmrecon.h:40-68) driven by apps/dmrecon/dmrecon.cc:53-65.
 * The host shim mve_amd/host/ re-implements that class (same headers, same
 * names, same exceptions) on top of the entry points declared here; see
 * INTEGRATION.md.  Every entry point cites the reference code it replaces.
 *
 * Conventions: plain pointers and sizes only; all buffers are caller-owned
 * host memory unless stated; every function returns 0 on success and a
 * negative MI_DMRECON_E* code on failure (message: mi_dmrecon_last_error(),
 * thread-local); nothing throws across this boundary.  A context is bound to
 * one GPU and must be used by one host thread at a time; distinct contexts are
 * independent (one per GPU = the multi-GPU sharding unit).
 */
#ifndef MI_DMRECON_H
#define MI_DMRECON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_DMRECON_OK            0
#define MI_DMRECON_EINVAL       -1   /* std::invalid_argument in the reference (dmrecon.cc:37-46,74-75) */
#define MI_DMRECON_EGVS         -2   /* std::runtime_error("Global View Selection failed") (dmrecon.cc:223) */
#define MI_DMRECON_EDEVICE      -3   /* HIP runtime error */
#define MI_DMRECON_ECANCELLED   -4   /* progress.cancelled observed (dmrecon.cc:101-105) */
#define MI_DMRECON_EFOOTPRINT   -5   /* std::out_of_range("Negative pixel footprint") (patch_sampler.cc:78-82) */
#define MI_DMRECON_ENOIMAGE     -6   /* a view the global view selection picked has no image (it was registered with its camera
                                      * only, mi_dmrecon_set_view with pixels = NULL): the reference fails there with the
                                      * exception of the image's loader (dmrecon.cc:236-240 loads the selected views) */

#define MI_DMRECON_MAX_GLOBAL_VIEWS 128  /* Settings::globalVSMax (apps/dmrecon -n, default 20) */
#define MI_DMRECON_MAX_LOCAL_VIEWS  16   /* Settings::nrReconNeighbors (apps/dmrecon --local-neighbors, default 4) */

typedef struct mi_dmrecon_ctx mi_dmrecon_ctx;

/* The mve::CameraInfo fields the path reads (libs/mve/camera.h; camera.cc:34-200). */
typedef struct mi_dmrecon_camera {
    float flen;
    float paspect;
    float ppoint[2];
    float rot[9];     /* world -> camera, row-major */
    float trans[3];
} mi_dmrecon_camera;

/* POD mirror of the algorithmic fields of mvs::Settings (libs/dmrecon/settings.h:22-52).
 * refViewNr is passed per call; imageEmbedding / ply / keep* flags stay in the host shim. */
typedef struct mi_dmrecon_settings {
    int32_t filterWidth;        /* 3, 5, 7, 9 or 11 (apps/dmrecon --filter-width; default 5, the only width for which the
                                 * reference's hard-coded derivative sample 12 is the centre, patch_sampler.cc:96) */
    float   minNCC;             /* 0.3 */
    float   minParallax;        /* 10 */
    float   acceptNCC;          /* 0.6 */
    float   minRefineDiff;      /* 0.001 */
    int32_t maxIterations;      /* 20 */
    int32_t nrReconNeighbors;   /* 4; 1..MI_DMRECON_MAX_LOCAL_VIEWS (above 4 a patch runs in the eight-view lane layouts, above 8 in the
                                 * sixteen-view one, which has host-visible rounds only: correct, not fast) */
    int32_t globalVSMax;        /* 20; <= MI_DMRECON_MAX_GLOBAL_VIEWS */
    int32_t scale;              /* 0 */
    int32_t useColorScale;      /* 1 */
    float   aabbMin[3];         /* -FLT_MAX */
    float   aabbMax[3];         /* +FLT_MAX */
} mi_dmrecon_settings;

/* mvs::ReconStatus / mvs::Progress (libs/dmrecon/progress.h:17-43).  The caller may
 * poll the fields from another thread and may set `cancelled` (as UMVE does). */
enum { MI_RECON_IDLE = 0, MI_RECON_GLOBALVS, MI_RECON_FEATURES, MI_RECON_QUEUE, MI_RECON_SAVING, MI_RECON_CANCELLED };
typedef struct mi_dmrecon_progress {
    volatile int32_t  status;
    volatile uint64_t filled;
    volatile uint64_t queueSize;    /* size of the current propagation work list */
    volatile uint64_t start_time;
    volatile int32_t  cancelled;
} mi_dmrecon_progress;

/* Output maps of one reference view: what DMRecon::start() hands to View::set_image
 * (dmrecon.cc:119-145).  Row-major, interleaved, W_s x H_s (mi_dmrecon_level_size at
 * settings.scale).  Unfilled pixels are exactly 0.  Any pointer may be NULL. */
typedef struct mi_dmrecon_maps {
    float*   depth;     /* 1 channel  "depth-L<s>"  */
    float*   normal;    /* 3 channels (SingleView::normalImg) */
    float*   dz;        /* 2 channels "dz-L<s>"     */
    float*   conf;      /* 1 channel  "conf-L<s>"   */
    int32_t* views;     /* local view ids of the accepted patch, ascending, -1 padded (QueueData::localViewIDs): 4 channels
                         * for nrReconNeighbors <= 4, 8 channels up to 8, 16 above (mi_dmrecon_local_view_channels) */
} mi_dmrecon_maps;

/* Work counters (device-counted) and timings of the last reconstruct call.  The struct only ever grows at its end (from the
 * layout of mi_dmrecon_abi_version() 5 on: that one put struct_size in front of everything), and
 * it carries its own size: the CALLER sets struct_size = sizeof(mi_dmrecon_stats) of the header it was built with before
 * every call, the library fills min(struct_size, its own sizeof) bytes and writes that number back -- a caller built
 * against an older (shorter) header is never overrun, one built against a newer header sees from struct_size which
 * fields it got.  A struct_size below 8 or absurdly large is MI_DMRECON_EINVAL (an object that was never initialised).
 * (The reference has no such struct: nothing of this is constrained by its interface.) */
typedef struct mi_dmrecon_stats {
    int64_t struct_size;    /* in: sizeof(mi_dmrecon_stats) as the caller knows it; out: the bytes the library filled */
    int64_t n_patch;        /* patch optimisations started (PatchOptimization objects) */
    int64_t n_eval;         /* patch-view evaluations: 25 bilinear samples of one neighbour view (SURVEY 8d unit) */
    int64_t n_filled;       /* pixels with depth (progress.filled) */
    int64_t n_seeds;        /* features processed (dmrecon.cc:287) */
    int64_t n_seeds_ok;     /* features whose optimisation succeeded (dmrecon.cc:301) */
    int64_t n_rounds;       /* propagation sweeps */
    int64_t n_launches;     /* launches of the optimisation kernel */
    double  ms_total;       /* wall time of the call, host clock */
    double  ms_opt_kernel;  /* hipEvent durations of the optimisation kernels on the context's stream, summed
                             * (= ms_bulk_kernel + ms_tail_kernel + ms_front_kernel) */
    double  ms_sweep_kernels; /* sum of hipEvent durations of the timed generate/apply kernels */
    /* the two halves of ms_opt_kernel, with their launch counts */
    double  ms_bulk_kernel; /* host-visible rounds (throughput layout; seeds; the hand-over round) */
    double  ms_tail_kernel; /* blind tail rounds (latency layout, one launch per round): every 8th round is bracketed by
                             * events (an event pair costs ~12 us of queue time per round) and their mean is multiplied by
                             * the number of tail launches that had work */
    int64_t n_bulk_launches;
    int64_t n_tail_launches;
    int64_t n_pass;         /* fused sampling passes actually run (each gathers the 100 texels of one patch-view
                             * once; a pass can stand for two of the reference's evaluations, see n_eval) */
    int64_t truncated;      /* 1 if the propagation ran out of round counters (the call fails with EDEVICE) */
    int64_t n_eval_bulk, n_patch_bulk, n_filled_bulk;   /* the share of n_eval / n_patch / n_filled of the host-visible rounds */
    int64_t n_view_replaced; /* local views dropped by replaceViews (patch_optimization.cc:218-228), by the attempts the reference's rule makes */
    int64_t n_iter14;        /* ... of which only by the iteration-14 rule (still moving at iterationCount == 14) */
    int64_t gvs_on_device;   /* 1 if the global view selection of this call ran on the GPU (gvs_device.hip) */
    double  ms_plan_gvs;     /* host clock: global view selection of all reference views of the call */
    double  ms_plan_seeds;   /* host clock: feature seeds of all reference views (dmrecon.cc:232-258) */
    int64_t n_merged_calls;  /* calls this execution served (concurrent calls on one scene with equal settings are merged
                              * into one batch; the call that ran it carries the statistics) */
    int64_t merged_into_other_call; /* 1: this call's views were reconstructed in another call's batch (all other fields 0) */
    /* the end of the tail in the front kernel (one persistent workgroup per reference view, its own rounds) */
    double  ms_front_kernel;     /* hipEvent duration of the front launch (= its slowest view) */
    double  ms_front_view_max;   /* the slowest view's own clock inside that launch */
    int64_t n_front_launches;    /* 0 or 1 */
    int64_t front_first_round;   /* the round at which the views went their own ways */
    int64_t n_front_views;       /* views that still had a front then */
    int64_t n_front_rounds_max;  /* rounds of the view that needed most (n_rounds = front_first_round + this) */
    int64_t n_front_rounds_sum;  /* ... summed over the views */
    int64_t n_front_attempts;    /* patch optimisations run there (speculative ones included) */
    int64_t n_front_entries;     /* list entries summed over all rounds of all views */
    int64_t front_team;          /* workgroups per view in that launch (> 1 only for a call that has the GPU to itself): the
                                  * smallest team; the views with the longest lists may have larger ones (front_team_max) */
    int64_t front_fallbacks;     /* 1: the teams gave up (a member found no compute unit in time: the GPU is shared with
                                  * something the library cannot see) and the views finished with one workgroup each */
    int64_t n_latency_rounds;    /* host-visible rounds in which some view was already in the latency layout */
    double  ms_wall_setup, ms_wall_rounds, ms_wall_front, ms_wall_download;   /* host clock: uploads, host-visible + tail rounds, front phase, download */
    int64_t n_patch_turns, n_wave_turns;   /* development builds (-DMI_ACTIVITY) only: turns of the patch optimisations of the throughput
                                  * layout, and turns of their wavefronts x patches per wavefront: the ratio = lanes at work */
    int64_t front_team_max;      /* the largest team of the front launch (the views with the longest lists get the teams of the
                                  * XCDs that hold fewer views) */
    int64_t n_sparse_records;    /* large batches: the maps went back as a snapshot taken at the hand-over (copied while the front
                                  * kernel ran) plus this many pixels the front changed afterwards (0: the maps were copied in
                                  * full; -1: the list outgrew its buffer and they were copied in full after all) */
    /* (round 6) the work counts once more, per KERNEL TEMPLATE -- index: 0 k_optimize<.., FAST> (first attempts of the large rounds,
     * and the follow-up launches that run in it), 1 k_optimize<.., SINGLE> (single-attempt follow-up launches of the general
     * kernel), 2 the seed launch, 3 k_optimize general (an entry's attempts in a row), 4 k_optimize_spec, 5 k_optimize in the latency
     * layout on a host-visible list, 6 k_tail, 7 k_front: what a profile's per-kernel time is divided by (profiles/r6_roofline_check.md) */
    int64_t n_eval_by_kernel[8], n_pass_by_kernel[8], n_patch_by_kernel[8];
    int64_t n_pass_executed_by_kernel[8];   /* ... and the passes a template EXECUTED: for k_optimize_spec the attempts the rule discards
                                             * included; elsewhere = n_pass_by_kernel (k_tail / k_front: the counted passes only) */
    double  ms_latency_rounds;   /* hipEvent durations of the latency-layout launches of the host-visible rounds (part of ms_bulk_kernel):
                                  * views that have handed over while others of the batch have not */
    int64_t n_latency_entries;   /* ... and the list entries they ran */
    double  shader_clock_mhz;    /* the shader clock the k_optimize launches of this call ran at: shader cycles over constant-rate
                                  * ticks of every 1024th wavefront's life (0: none sampled) -- the clock the VALU-issue roof is priced at */
    int64_t clk_shader_cycles, clk_real_ticks;   /* ... its two sums (they add up over calls) and the rate of the constant clock */
    double  clk_real_mhz;
} mi_dmrecon_stats;

/* The version of this header's binary interface (MI_DMRECON_ABI_VERSION of the header the library was built from).  It changes
 * whenever a struct changes in a way `struct_size` cannot express or a function changes its signature: 5 = the layout in which
 * mi_dmrecon_stats gained `struct_size` as its FIRST member (round 5: every other field moved by 8 bytes -- a caller built against
 * the header before that must be recompiled; the append-only rule of mi_dmrecon_stats holds from that layout on); 6 = this header
 * (statistics appended: the per-kernel-template counts, the latency-round and shader-clock fields).  A caller compares it with the
 * MI_DMRECON_ABI_VERSION it was compiled with: a library that reports less than 5, or that does not export the function, has the
 * old statistics layout. */
#define MI_DMRECON_ABI_VERSION 6
int  mi_dmrecon_abi_version(void);
int  mi_dmrecon_device_count(void);
/* channels of mi_dmrecon_maps::views / of the local-view arguments of mi_dmrecon_patch_optimize: 4 or 8 */
int  mi_dmrecon_local_view_channels(int32_t nrReconNeighbors);
const char* mi_dmrecon_last_error(void);
void mi_dmrecon_settings_default(mi_dmrecon_settings* s);              /* settings.h:25-51 */

int  mi_dmrecon_ctx_create(int device, mi_dmrecon_ctx** out);
void mi_dmrecon_ctx_destroy(mi_dmrecon_ctx* ctx);
/* A sibling context on the same GPU that SHARES the parent's resident views and features (read-only)
 * but has its own HIP stream and scratch memory.  This is how several mvs::DMRecon instances run
 * concurrently, as the OpenMP loop of apps/dmrecon/dmrecon.cc:285-318 does with one shared mve::Scene:
 * one fork per host thread, their kernels overlap on the GPU.  Views/features must not be changed
 * while forks are reconstructing.  Destroy forks and parent in any order. */
int  mi_dmrecon_ctx_fork(mi_dmrecon_ctx* parent, mi_dmrecon_ctx** out);
/* Optional page-locked host buffers for mi_dmrecon_maps (any host memory works; pinned memory makes the
 * final device->host copy of the maps run at PCIe rate).  Returns NULL on failure. */
void* mi_dmrecon_host_alloc(size_t bytes);
void  mi_dmrecon_host_free(void* p);
/* The HIP stream all kernels of this context are launched on (hipStream_t as void*). */
void* mi_dmrecon_ctx_stream(mi_dmrecon_ctx* ctx);

/* Replaces SingleView::SingleView + ImagePyramidCache::get / buildPyramid / ensureImages
 * (single_view.cc:24-50, image_pyramid.cc:21-132): registers view `view_id` with its camera
 * and level-0 image (uint8, channels 1..4: grey is expanded, alpha dropped, image_pyramid.cc:65-73),
 * builds the Gaussian pyramid on the device (image_tools.h:619-690) and keeps every level resident in HBM. */
int  mi_dmrecon_set_view(mi_dmrecon_ctx* ctx, int32_t view_id, const mi_dmrecon_camera* cam,
                         int32_t width, int32_t height, int32_t channels, const uint8_t* pixels);
/* pixels = NULL registers the view with its camera and image size ONLY: the reference creates a SingleView for every view
 * with a valid camera and an image of the embedding (dmrecon.cc:62-79) and loads images lazily -- a view whose image cannot be
 * decoded is still a candidate of everybody's global view selection, and only a reconstruction that SELECTS it fails
 * (dmrecon.cc:236-240).  Such a view takes part in mi_dmrecon_global_view_selection as any other; a reference view that
 * selects it ends with MI_DMRECON_ENOIMAGE (its own status; the others of a batch finish); as a reference view itself it is
 * "Invalid master view". */
/* Same, without waiting: the host->device copy, the RGBA pack and the pyramid kernels are only enqueued.
 * `pixels` must stay valid (and should be page-locked, mi_dmrecon_host_alloc, for the copy to be truly
 * asynchronous) until mi_dmrecon_sync() -- the staging path for scenes whose level-0 images do not fit
 * the time budget of a blocking upload (BASELINE config 5: 100 x 4032x3024). */
int  mi_dmrecon_set_view_async(mi_dmrecon_ctx* ctx, int32_t view_id, const mi_dmrecon_camera* cam,
                               int32_t width, int32_t height, int32_t channels, const uint8_t* pixels);
int  mi_dmrecon_sync(mi_dmrecon_ctx* ctx);
int  mi_dmrecon_evict_view(mi_dmrecon_ctx* ctx, int32_t view_id);     /* ImagePyramidCache::cleanup */

/* mve::Bundle::Features (libs/mve/bundle.h:46-56) in CSR form: feature i has position pos[3i..3i+2]
 * and is referenced by views ref_view_ids[ref_offsets[i] .. ref_offsets[i+1]). */
int  mi_dmrecon_set_features(mi_dmrecon_ctx* ctx, int32_t n_features, const float* pos,
                             const int32_t* ref_offsets, const int32_t* ref_view_ids);

int  mi_dmrecon_num_levels(mi_dmrecon_ctx* ctx, int32_t view_id);
int  mi_dmrecon_level_size(mi_dmrecon_ctx* ctx, int32_t view_id, int32_t level, int32_t* width, int32_t* height);
/* Reads a pyramid level back (rgb: w*h*3, proj/invproj: 9 floats; any may be NULL) -- the
 * "undist-L<s>" embedding (dmrecon.cc:135-140) and a test hook for the pyramid kernel. */
int  mi_dmrecon_get_level(mi_dmrecon_ctx* ctx, int32_t view_id, int32_t level, uint8_t* rgb,
                          float* proj, float* invproj);

/* DMRecon::analyzeFeatures + DMRecon::globalViewSelection (dmrecon.cc:178-241,
 * global_view_selection.cc:33-101).  ids_out: >= MI_DMRECON_MAX_GLOBAL_VIEWS entries, ascending. */
int  mi_dmrecon_global_view_selection(mi_dmrecon_ctx* ctx, const mi_dmrecon_settings* st, int32_t ref_view,
                                      int32_t* ids_out, int32_t* n_out);

/* DMRecon::start() for n_refs reference views at once (dmrecon.cc:89-172: analyzeFeatures,
 * globalViewSelection, processFeatures, processQueue).  The views are independent; batching
 * them fills the GPU.  maps[i] / progress[i] belong to ref_views[i]; progress may be NULL.
 * Views end individually, as mvs::DMRecon instances do: status_out[i] (may be NULL) receives the view's own
 * outcome -- 0, MI_DMRECON_EGVS, MI_DMRECON_ENOIMAGE (a selected view was registered without pixels),
 * MI_DMRECON_EFOOTPRINT (patch_sampler.cc:78-82 throws for that view only) or
 * MI_DMRECON_ECANCELLED (progress[i].cancelled was set, before or during the run: dmrecon.cc:101-105,353) -- and
 * the maps of a view that did not finish are left untouched (a view is only ever written once it has ended well), with ONE
 * exception: a call of 48 or more views without a progress array copies the state of its views as of the hand-over to the
 * per-view front kernel into the callers' buffers while that kernel runs (and the pixels it changed afterwards); a view that
 * ends with an error AFTER that point -- with valid cameras only a device error can make one: a propagated hypothesis has a
 * positive depth, and there is no progress array to cancel through -- has had that intermediate state written to its buffers.
 * Its status says so: maps of a view whose status is not 0 are void, whatever they hold (MI_DMRECON_SPARSE_MAPS=0 keeps the
 * strict form).  progress[i].filled counts that view's pixels.
 * Return value: 0 if at least one view finished; with a single view (or when every view failed) the failing
 * view's own code.
 * Calls without a progress array that arrive at the same time on contexts of one scene (ctx_fork) with equal
 * settings are run as ONE batch by one of the callers (the others wait for their maps): statuses, return codes AND maps
 * are those of the separate calls, bit for bit -- everything that decides the last bits of a view's maps (the round at
 * which the view leaves the throughput lane layout, whose sums run in another order) is decided per view from the view's
 * own history, never from the batch (as apps/dmrecon's all-views mode writes what `-m ID` writes,
 * apps/dmrecon/dmrecon.cc:285-318).  The statistics go to the call that ran the batch (stats.n_merged_calls; the
 * others get zeros and stats.merged_into_other_call = 1).  MI_DMRECON_MERGE_CALLS=0 switches the merging off.
 * A call that has the GPU to itself runs the end of its propagation with several persistent workgroups per view that
 * wait for each other every pass (stats.front_team).  Only one call per GPU does so at a time, across processes (an
 * advisory lock file named after the GPU's PCI address in /dev/shm), and a team whose members are not all given a
 * compute unit in time -- something the library cannot see holds them -- gives up and the views finish with one
 * workgroup each (stats.front_fallbacks): slower, the same maps, never an error. */
int  mi_dmrecon_reconstruct(mi_dmrecon_ctx* ctx, const mi_dmrecon_settings* st, int32_t n_refs,
                            const int32_t* ref_views, mi_dmrecon_maps* maps,
                            mi_dmrecon_progress* progress, int32_t* status_out, mi_dmrecon_stats* stats);

/* Patch-level entry (parity hook; = constructing mvs::PatchOptimization, doAutoOptimization,
 * computeConfidence -- patch_optimization.cc:21-78,170-242,114-142) for n hypotheses in
 * reference view ref_view: xy[2n]; hyp[3n] = depth,dzI,dzJ; local[Cn] view ids (-1 = none, may be NULL),
 * C = mi_dmrecon_local_view_channels(nrReconNeighbors).
 * lanes_per_view: the lane layout to run them in (1 = throughput layout, 16 patches per wavefront; 16 = latency
 * layout, one patch per wavefront -- the two layouts of the product path, same mathematics; more than eight local views
 * exist in the throughput layout only: 16 is MI_DMRECON_EINVAL then).
 * out[8n] = conf, depth, dzI, dzJ, nx, ny, nz, iterationCount; out_local[Cn] ascending ids, -1 padded. */
int  mi_dmrecon_patch_optimize(mi_dmrecon_ctx* ctx, const mi_dmrecon_settings* st, int32_t ref_view, int32_t n,
                               const int32_t* xy, const float* hyp, const int32_t* local, int32_t lanes_per_view,
                               float* out, int32_t* out_local);

/* Patch-level evaluation hook (PatchSampler::getFastNCC + fastColAndDeriv, patch_sampler.cc:64-163)
 * of ONE hypothesis against every global view g (order of mi_dmrecon_global_view_selection):
 * master[5] = ok, masterMeanCol, normal; ncc[g]; ok[g]; level[g]; col/deriv [g][filterWidth^2][3]. */
int  mi_dmrecon_patch_eval(mi_dmrecon_ctx* ctx, const mi_dmrecon_settings* st, int32_t ref_view,
                           int32_t x, int32_t y, float depth, float dzI, float dzJ,
                           float* master, float* ncc, int32_t* ok, float* col, float* deriv, int32_t* level);

/* ---- next stage downstream: depth map -> oriented point set ------------------------------- */

/* Options of apps/scene2pset (scene2pset.cc:44-63 AppSettings, :262-356). */
typedef struct mi_dmrecon_pointset_options {
    float   dd_factor;        /* 5.0  depth-discontinuity factor of depthmap_triangulate (depthmap.cc:323,372) */
    float   scale_factor;     /* 2.5  "--scale-factor" (scene2pset.cc:58, :355) */
    int32_t conf_iterations;  /* 4    depthmap_mesh_confidences(mesh, 4) (scene2pset.cc:329) */
} mi_dmrecon_pointset_options;

/* The per-view body of apps/scene2pset (scene2pset.cc:262-356) for one depth map: triangulate the depth map
 * (mve::geom::depthmap_triangulate, libs/mve/depthmap.cc:210-399), compute angle-weighted vertex normals
 * (TriangleMesh::recalc_normals, libs/mve/mesh.cc:25-160), border-distance confidences
 * (depthmap_mesh_confidences, depthmap.cc:497-546) and per-vertex scale (scene2pset.cc:343-356), and return the
 * vertices.  depth: w*h floats (0 = no depth); color: w*h*color_channels bytes (1 or 3 channels) or NULL;
 * cam: the view's camera (the intrinsics are applied to w x h, i.e. w,h may be a pyramid level).
 * Outputs (each may be NULL) hold `capacity` entries; a vertex's record is pixel (y*w+x), pos[3] (world),
 * normal[3], color[3] (0..1), scale, conf.  Order: ascending pixel index (the reference numbers vertices by first
 * use while scanning 2x2 blocks; as a set the output is the same).  *n_out = number of vertices found (also when
 * > capacity, in which case only `capacity` are written).  opt == NULL: the defaults above. */
int  mi_dmrecon_pointset(mi_dmrecon_ctx* ctx, const mi_dmrecon_camera* cam, int32_t w, int32_t h,
                         const float* depth, const uint8_t* color, int32_t color_channels,
                         const mi_dmrecon_pointset_options* opt, int32_t capacity,
                         int32_t* pixel, float* pos, float* normal, float* color_out, float* scale, float* conf,
                         int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* MI_DMRECON_H */
