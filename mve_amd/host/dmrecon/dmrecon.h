/* mvs::DMRecon -- same public surface as libs/dmrecon/dmrecon.h:40-68 (constructor, getRefViewNr,
 * getProgress x2, start); the private part talks to libmi_dmrecon.so instead of owning the
 * reference's priority queue.
 *
 * Interface attribution: the names, field order and default values mirrored here are those of MVE's
 * libs/dmrecon public headers, Copyright (C) 2015 Simon Fuhrmann, Ronny Klowsky, TU Darmstadt, distributed under
 * the BSD 3-Clause license (LICENSE.txt of simonfuhrmann/mve).  Only the declarations a caller compiles against
 * are mirrored; the implementation behind them is this repository's.
 */
#ifndef MI_DMRECON_SHIM_DMRECON_H
#define MI_DMRECON_SHIM_DMRECON_H

/* the reference header pulls these in and its callers rely on that (fancy_progress_printer.cc uses
 * std::cout without including <iostream> itself) */
#include <fstream>
#include <memory>
#include <iostream>
#include <queue>
#include <string>
#include <vector>

#include "mve/bundle.h"
#include "mve/image.h"
#include "mve/scene.h"
#include "dmrecon/defines.h"
#include "dmrecon/progress.h"
#include "dmrecon/settings.h"

struct mi_dmrecon_ctx;

MVS_NAMESPACE_BEGIN

class DMRecon
{
public:
    DMRecon(mve::Scene::Ptr scene, Settings const& settings);

    std::size_t getRefViewNr() const { return settings.refViewNr; }
    Progress const& getProgress() const { return progress; }
    Progress& getProgress() { return progress; }
    void start();

private:
    mve::Scene::Ptr scene;
    Settings settings;
    Progress progress;
    std::shared_ptr<void> slot;   /* the resident scene's generation and the GPU slot of it this instance submits to */
    int width, height;
};

MVS_NAMESPACE_END

#endif
