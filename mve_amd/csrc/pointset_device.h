/* Internal: layouts + launcher of pointset_device.hip. */
#ifndef MI_POINTSET_DEVICE_H
#define MI_POINTSET_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PsParams {
    int32_t w, h;
    float inv[9];            /* CameraInfo::fill_inverse_calibration for w x h */
    float ctw[12];           /* CameraInfo::fill_cam_to_world, first three rows */
    float dd_factor, scale_factor;
    int32_t conf_iterations;
};

struct PsVertex {            /* one per pixel */
    float pos[3], nrm[3], scale;
    uint16_t adj;            /* adjacent vertices as a mask over the 3x3 neighbourhood */
    uint8_t used, vclass;    /* MeshInfo::VertexClass: 0 SIMPLE, 1 BORDER, 2 COMPLEX, 3 UNREF */
    int8_t level;            /* hop distance to the mesh border, -1 = further than conf_iterations - 1 */
    uint8_t pad[3];
};

void mi_ps_launch(hipStream_t s, const PsParams& p, const float* depth, uint8_t* cells, PsVertex* verts);
#endif
