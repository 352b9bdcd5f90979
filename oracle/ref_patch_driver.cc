/*
 * TEST INFRASTRUCTURE ONLY (oracle/): patch-level driver around the UNMODIFIED
 * reference classes.  This file is ours; it is compiled against the reference
 * headers under /root/reference/libs and linked with the reference objects
 * built by oracle/Makefile into oracle/_ref/ref_patch_driver.
 *
 * It exposes what the reference's own apps never print: the per-patch values
 * of mvs::PatchSampler / mvs::PatchOptimization (libs/dmrecon/patch_sampler.cc,
 * patch_optimization.cc) for a list of hypotheses, so that the CPU restatement
 * (oracle/dmrecon_oracle.cc) and the HIP kernels can be pinned per function
 * rather than only per depth map.
 *
 * usage: ref_patch_driver SCENE REFVIEW SCALE LOCALNEIGH MODE SEEDS OUT
 *   MODE = gvs    -> OUT: "G id0 id1 ..." (global view selection result)
 *   MODE = eval   -> per seed, per global view: NCC at the hypothesis, and
 *                    colour / derivative samples of fastColAndDeriv
 *   MODE = opt    -> per seed: result of doAutoOptimization + computeConfidence
 * SEEDS: text, one hypothesis per line: x y depth dzI dzJ nlocal [ids...]
 * Environment: REF_FILTER_WIDTH = mvs::Settings::filterWidth (default 5), REF_GLOBAL_VS_MAX = globalVSMax (default 20).
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "mve/scene.h"
#include "mve/bundle.h"
#include "math/octree_tools.h"
#include "dmrecon/settings.h"
#include "dmrecon/single_view.h"
#include "dmrecon/global_view_selection.h"
#include "dmrecon/patch_sampler.h"
#include "dmrecon/patch_optimization.h"

struct Seed { int x, y; float depth, dzI, dzJ; mvs::IndexSet local; };

int main (int argc, char** argv)
{
    if (argc != 8) {
        std::fprintf(stderr, "usage: %s SCENE REFVIEW SCALE LOCALNEIGH MODE SEEDS OUT\n", argv[0]);
        return 2;
    }
    std::string scene_path = argv[1];
    mvs::Settings st;
    st.refViewNr = std::atoi(argv[2]);
    st.scale = std::atoi(argv[3]);
    st.nrReconNeighbors = std::atoi(argv[4]);
    st.quiet = true;
    if (char const* fw = std::getenv("REF_FILTER_WIDTH")) st.filterWidth = std::atoi(fw);   /* apps/dmrecon --filter-width */
    if (char const* gm = std::getenv("REF_GLOBAL_VS_MAX")) st.globalVSMax = std::atoi(gm);  /* apps/dmrecon -n / --neighbors */
    std::string mode = argv[5];

    mve::Scene::Ptr scene = mve::Scene::create(scene_path);
    mve::Bundle::ConstPtr bundle = scene->get_bundle();
    mve::Scene::ViewList const& mv(scene->get_views());

    /* Same construction sequence as mvs::DMRecon::DMRecon (dmrecon.cc:62-79). */
    std::vector<mvs::SingleView::Ptr> views(mv.size());
    for (std::size_t i = 0; i < mv.size(); ++i) {
        if (mv[i] == nullptr || !mv[i]->is_camera_valid()
            || !mv[i]->has_image(st.imageEmbedding, mve::IMAGE_TYPE_UINT8))
            continue;
        views[i] = mvs::SingleView::create(scene, mv[i], st.imageEmbedding);
    }
    mvs::SingleView::Ptr refV = views[st.refViewNr];
    if (refV == nullptr) { std::fprintf(stderr, "invalid master view\n"); return 1; }
    refV->loadColorImage(st.scale);
    refV->prepareMasterView(st.scale);

    /* Same as DMRecon::analyzeFeatures (dmrecon.cc:178-208). */
    mve::Bundle::Features const& features = bundle->get_features();
    for (std::size_t i = 0; i < features.size(); ++i) {
        if (!features[i].contains_view_id(st.refViewNr)) continue;
        math::Vec3f p(features[i].pos);
        if (!refV->pointInFrustum(p)) continue;
        if (!math::geom::point_box_overlap(p, st.aabbMin, st.aabbMax)) continue;
        for (std::size_t j = 0; j < features[i].refs.size(); ++j) {
            int id = features[i].refs[j].view_id;
            if (id < 0 || id >= (int)views.size() || views[id] == nullptr) continue;
            if (views[id]->pointInFrustum(p)) views[id]->addFeature(i);
        }
    }
    /* Same as DMRecon::globalViewSelection (dmrecon.cc:210-241). */
    mvs::GlobalViewSelection gvs(views, features, st);
    gvs.performVS();
    mvs::IndexSet neigh = gvs.getSelectedIDs();
    for (mvs::IndexSet::const_iterator it = neigh.begin(); it != neigh.end(); ++it)
        views[*it]->loadColorImage(0);

    std::ofstream out(argv[7]);
    out.precision(9);
    out << "G";
    for (mvs::IndexSet::const_iterator it = neigh.begin(); it != neigh.end(); ++it)
        out << " " << *it;
    out << "\n";
    if (mode == "gvs") return 0;

    std::vector<Seed> seeds;
    {
        std::ifstream in(argv[6]);
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            std::istringstream ss(line);
            Seed s; int nl = 0;
            ss >> s.x >> s.y >> s.depth >> s.dzI >> s.dzJ >> nl;
            for (int k = 0; k < nl; ++k) { std::size_t id; ss >> id; s.local.insert(id); }
            seeds.push_back(s);
        }
    }

    for (std::size_t si = 0; si < seeds.size(); ++si) {
        Seed const& s = seeds[si];
        if (mode == "eval") {
            mvs::PatchSampler::Ptr smp = mvs::PatchSampler::create(views, st,
                s.x, s.y, s.depth, s.dzI, s.dzJ);
            bool ok = smp->success[st.refViewNr];
            out << "S " << si << " " << (ok ? 1 : 0);
            if (ok) {
                math::Vec3f n = smp->getPatchNormal();
                out << " " << smp->getMasterMeanColor() << " " << n[0] << " " << n[1] << " " << n[2];
            }
            out << "\n";
            if (!ok) continue;
            for (mvs::IndexSet::const_iterator it = neigh.begin(); it != neigh.end(); ++it) {
                float ncc = smp->getFastNCC(*it);
                mvs::Samples col, der;
                smp->fastColAndDeriv(*it, col, der);
                bool okv = smp->success[*it];
                out << "V " << *it << " " << ncc << " " << (okv ? 1 : 0);
                if (okv) {
                    for (std::size_t i = 0; i < col.size(); ++i)
                        for (int c = 0; c < 3; ++c) out << " " << col[i][c];
                    for (std::size_t i = 0; i < der.size(); ++i)
                        for (int c = 0; c < 3; ++c) out << " " << der[i][c];
                }
                out << "\n";
            }
        } else {
            mvs::PatchOptimization patch(views, st, s.x, s.y, s.depth, s.dzI, s.dzJ,
                neigh, s.local);
            patch.doAutoOptimization();
            float conf = patch.computeConfidence();
            out << "P " << si << " " << conf << " " << patch.getDepth() << " "
                << patch.getDzI() << " " << patch.getDzJ();
            math::Vec3f n(0.f);
            if (conf > 0.f) n = patch.getNormal();
            out << " " << n[0] << " " << n[1] << " " << n[2];
            mvs::IndexSet const& loc = patch.getLocalViewIDs();
            out << " " << loc.size();
            for (mvs::IndexSet::const_iterator it = loc.begin(); it != loc.end(); ++it)
                out << " " << *it;
            out << "\n";
        }
    }
    return 0;
}
