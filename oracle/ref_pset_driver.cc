/*
 * TEST INFRASTRUCTURE ONLY (oracle/): runs the UNMODIFIED reference functions behind apps/scene2pset for one
 * view -- mve::geom::depthmap_triangulate (libs/mve/depthmap.cc:377-399), TriangleMesh::ensure_normals
 * (libs/mve/mesh.cc:25-173), mve::geom::depthmap_mesh_confidences (depthmap.cc:497-546) and the per-vertex
 * scale of apps/scene2pset/scene2pset.cc:343-356 -- and dumps the vertices in binary, each with the pixel it
 * came from, so that order-independent comparisons are possible.  This file is ours; it is compiled against
 * the reference headers and linked with the reference objects by oracle/Makefile.
 *
 * usage: ref_pset_driver SCENE VIEWID DMNAME IMAGENAME SCALEFACTOR OUT
 * OUT: int32 n, then n records of { int32 pixel; float pos[3], normal[3], color[3], scale, conf; }
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "mve/depthmap.h"
#include "mve/mesh.h"
#include "mve/mesh_info.h"
#include "mve/scene.h"

int main (int argc, char** argv)
{
    if (argc != 7) { std::fprintf(stderr, "usage: %s SCENE VIEWID DMNAME IMAGENAME SCALEFACTOR OUT\n", argv[0]); return 2; }
    mve::Scene::Ptr scene = mve::Scene::create(argv[1]);
    mve::View::Ptr view = scene->get_view_by_id(std::atoi(argv[2]));
    if (view == nullptr) { std::fprintf(stderr, "no such view\n"); return 1; }
    mve::FloatImage::Ptr dm = view->get_float_image(argv[3]);
    mve::ByteImage::Ptr ci = view->get_byte_image(argv[4]);
    if (dm == nullptr) { std::fprintf(stderr, "no depth map\n"); return 1; }
    float const scale_factor = std::atof(argv[5]);
    mve::CameraInfo const& cam = view->get_camera();

    mve::Image<unsigned int> vertex_ids;
    mve::TriangleMesh::Ptr mesh = mve::geom::depthmap_triangulate(dm, ci, cam, mve::geom::DD_FACTOR_DEFAULT, &vertex_ids);
    if (ci == nullptr) {
        /* without a colour image the reference returns before it hands out the id map (depthmap.cc:341-342):
         * fetch the (deterministic) numbering from the invproj-level overload it calls internally */
        math::Matrix3f invproj;
        cam.fill_inverse_calibration(*invproj, dm->width(), dm->height());
        mve::geom::depthmap_triangulate(dm, invproj, mve::geom::DD_FACTOR_DEFAULT, &vertex_ids);
    }
    mesh->ensure_normals();
    mve::geom::depthmap_mesh_confidences(mesh, 4);
    mve::TriangleMesh::VertexList const& verts(mesh->get_vertices());
    mve::TriangleMesh::NormalList const& norms(mesh->get_vertex_normals());
    mve::TriangleMesh::ColorList const& cols(mesh->get_vertex_colors());
    mve::TriangleMesh::ConfidenceList const& confs(mesh->get_vertex_confidences());
    std::vector<float> scale(verts.size(), 0.0f);
    mve::MeshInfo mesh_info(mesh);
    for (std::size_t j = 0; j < mesh_info.size(); ++j) {
        mve::MeshInfo::VertexInfo const& vinf = mesh_info[j];
        for (std::size_t k = 0; k < vinf.verts.size(); ++k)
            scale[j] += (verts[j] - verts[vinf.verts[k]]).norm();
        scale[j] /= static_cast<float>(vinf.verts.size());
        scale[j] *= scale_factor;
    }
    std::vector<int> pixel_of(verts.size(), -1);
    for (int i = 0; i < vertex_ids.get_pixel_amount(); ++i)
        if (vertex_ids[i] != MATH_MAX_UINT) pixel_of[vertex_ids[i]] = i;

    std::ofstream out(argv[6], std::ios::binary);
    int32_t n = (int32_t)verts.size();
    out.write((char const*)&n, 4);
    for (std::size_t i = 0; i < verts.size(); ++i) {
        int32_t px = pixel_of[i];
        float rec[11] = { verts[i][0], verts[i][1], verts[i][2], norms[i][0], norms[i][1], norms[i][2],
            cols.empty() ? 0.f : cols[i][0], cols.empty() ? 0.f : cols[i][1], cols.empty() ? 0.f : cols[i][2],
            scale[i], confs[i] };
        out.write((char const*)&px, 4);
        out.write((char const*)rec, sizeof(rec));
    }
    return 0;
}
