/* mvs::Settings -- field-for-field the public struct of libs/dmrecon/settings.h:22-52 (same names,
 * types and defaults: callers assign these fields directly, apps/dmrecon/dmrecon.cc:166-222).
 *
 * Interface attribution: the names, field order and default values mirrored here are those of MVE's
 * libs/dmrecon public headers, Copyright (C) 2015 Simon Fuhrmann, Ronny Klowsky, TU Darmstadt, distributed under
 * the BSD 3-Clause license (LICENSE.txt of simonfuhrmann/mve).  Only the declarations a caller compiles against
 * are mirrored; the implementation behind them is this repository's.
 */
#ifndef MI_DMRECON_SHIM_SETTINGS_H
#define MI_DMRECON_SHIM_SETTINGS_H

#include <limits>
#include <string>

#include "math/vector.h"
#include "dmrecon/defines.h"

MVS_NAMESPACE_BEGIN

struct Settings
{
    std::size_t refViewNr = 0;
    std::string imageEmbedding = "undistorted";
    unsigned int filterWidth = 5;
    float minNCC = 0.3f;
    float minParallax = 10.0f;
    float acceptNCC = 0.6f;
    float minRefineDiff = 0.001f;
    unsigned int maxIterations = 20;
    unsigned int nrReconNeighbors = 4;
    unsigned int globalVSMax = 20;
    int scale = 0;
    bool useColorScale = true;
    bool writePlyFile = false;
    math::Vec3f aabbMin = math::Vec3f(-std::numeric_limits<float>::max());
    math::Vec3f aabbMax = math::Vec3f(std::numeric_limits<float>::max());
    std::string plyPath;
    bool keepDzMap = false;
    bool keepConfidenceMap = false;
    bool quiet = false;
};

MVS_NAMESPACE_END

#endif
