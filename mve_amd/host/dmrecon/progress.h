/* mvs::ReconStatus / mvs::Progress -- the public struct of libs/dmrecon/progress.h:17-43.  The
 * progress printer thread of apps/dmrecon polls it; UMVE sets `cancelled` from another thread.
 *
 * Interface attribution: the names, field order and default values mirrored here are those of MVE's
 * libs/dmrecon public headers, Copyright (C) 2015 Simon Fuhrmann, Ronny Klowsky, TU Darmstadt, distributed under
 * the BSD 3-Clause license (LICENSE.txt of simonfuhrmann/mve).  Only the declarations a caller compiles against
 * are mirrored; the implementation behind them is this repository's.
 */
#ifndef MI_DMRECON_SHIM_PROGRESS_H
#define MI_DMRECON_SHIM_PROGRESS_H

#include <cstddef>

#include "dmrecon/defines.h"

MVS_NAMESPACE_BEGIN

enum ReconStatus
{
    RECON_IDLE,
    RECON_GLOBALVS,
    RECON_FEATURES,
    RECON_QUEUE,
    RECON_SAVING,
    RECON_CANCELLED
};

struct Progress
{
    ReconStatus status = RECON_IDLE;
    std::size_t filled = 0;
    std::size_t queueSize = 0;
    std::size_t start_time = 0;
    bool cancelled = false;
};

MVS_NAMESPACE_END

#endif
