/*
 * TEST INFRASTRUCTURE ONLY.  C interface of the CPU restatement of MVE's
 * libs/dmrecon (see dmrecon_oracle.cc).  Nothing on the product path may
 * include, link or load this; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, as the checker.
 */
#ifndef DMRECON_ORACLE_H
#define DMRECON_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float flen, paspect, ppoint[2], rot[9], trans[3];   /* mve::CameraInfo */
} orc_camera;

/* POD mirror of mvs::Settings (libs/dmrecon/settings.h:22-52). */
typedef struct {
    int32_t refViewNr;
    int32_t filterWidth;        /* only 5 is supported, as in the HIP path */
    float minNCC, minParallax, acceptNCC, minRefineDiff;
    int32_t maxIterations, nrReconNeighbors, globalVSMax, scale;
    int32_t useColorScale;
    float aabbMin[3], aabbMax[3];
} orc_settings;

typedef struct {
    int64_t n_patch;      /* PatchOptimization objects constructed */
    int64_t n_eval;       /* patch-view evaluations (fastColAndDeriv + computeNeighColorSamples) */
    int64_t n_filled;     /* progress.filled */
    int64_t n_seeds_ok;   /* features that succeeded optimisation */
    int64_t n_seeds;      /* features processed */
} orc_stats;

void* orc_scene_create(int n_views);
void orc_scene_destroy(void* scene);
void orc_settings_default(orc_settings* s);
/* rgb: level-0 image, HxWx3 uint8.  The pyramid is built immediately. */
void orc_scene_set_view(void* scene, int id, const orc_camera* cam, int w, int h, const uint8_t* rgb);
void orc_scene_set_features(void* scene, int n, const float* pos, const int32_t* ref_off,
                            const int32_t* ref_views);
int orc_pyramid_levels(void* scene, int view);
/* any out pointer may be null */
void orc_pyramid_get(void* scene, int view, int level, int32_t* wh, uint8_t* rgb, float* proj, float* invproj);

/* analyzeFeatures + GlobalViewSelection; returns count, ids ascending. */
int orc_global_vs(void* scene, const orc_settings* st, int32_t* ids_out);

/* Whole DMRecon::start() for st->refViewNr.  Maps are W_s x H_s (orc_pyramid_get). Any map may be null. */
int orc_reconstruct(void* scene, const orc_settings* st, float* depth, float* normal, float* dz,
                    float* conf, orc_stats* stats);

/* n hypotheses: xy[2n], hyp[3n] = depth,dzI,dzJ, local[ORC_MAX_LOCAL n] view ids (-1 = none; the hook carries up to
 * ORC_MAX_LOCAL local views, the algorithm itself has no limit).
 * out[8n] = conf, depth, dzI, dzJ, nx, ny, nz, iterations; out_local[ORC_MAX_LOCAL n]. */
#define ORC_MAX_LOCAL 16
int orc_patch_optimize(void* scene, const orc_settings* st, int n, const int32_t* xy, const float* hyp,
                       const int32_t* local, float* out, int32_t* out_local);

/* One hypothesis, every global view g (order of orc_global_vs):
 * ncc[g]; ok[g]; col[g][25][3]; deriv[g][25][3]; level[g].  master[0] = ok, [1] = masterMeanCol, [2..4] = normal */
int orc_patch_eval(void* scene, const orc_settings* st, int x, int y, float depth, float dzI, float dzJ,
                   float* master, float* ncc, int32_t* ok, float* col, float* deriv, int32_t* level);

#ifdef __cplusplus
}
#endif
#endif
