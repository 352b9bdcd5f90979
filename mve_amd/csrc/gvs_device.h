/* Internal: launcher of gvs_device.hip -- the global view selection on the device. */
#ifndef MI_GVS_DEVICE_H
#define MI_GVS_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_GVS_MAX_OUT 128         /* = MI_MAX_GLOBAL: ids written per reference view */

/* The scene tables of SceneGeom (dmrecon_host.cpp) in device memory, plus what benefitFromView needs per view. */
struct GvsScene {
    int32_t nv, nf;
    const uint8_t* sees;           /* [nv][nf] */
    const float* zcam;             /* [nv][nf] */
    const float* plx;              /* [nv][nv][nf] degrees */
    const float* fpos;             /* [nf][3] */
    const float* inv0;             /* [nv] invproj[0] of level 0 (SingleView::footPrint) */
    const uint8_t* valid;          /* [nv] */
};

struct GvsRef {                    /* one per reference view of the call */
    int32_t ref;
    float inv_m;                   /* invproj[0] of the reference level (footPrintScaled) */
};

struct GvsArgs {
    GvsScene sc;
    const GvsRef* refs;
    float minParallax;
    int32_t globalVSMax;
    float aabb_min[3], aabb_max[3];
    int32_t use_box;
    /* scratch, per reference view (blockIdx.x): */
    int32_t* feat;                 /* [n_refs][nf] attached features, ascending */
    float* base;                   /* [n_refs][nv][nf] indexed by position in feat */
    float* benefit;                /* [n_refs][nv] */
    /* output */
    int32_t* out_ids;              /* [n_refs][MI_GVS_MAX_OUT] ascending */
    int32_t* out_n;                /* [n_refs] */
};

void mi_gvs_launch(hipStream_t s, const GvsArgs& a, int n_refs);
#endif
