/*
 * Drop-in header set for MVE's libs/dmrecon public interface, backed by the MI355X library
 * (include/mi_dmrecon.h).  apps/dmrecon/dmrecon.cc and UMVE include "dmrecon/settings.h" and
 * "dmrecon/dmrecon.h" (apps/dmrecon/dmrecon.cc:14-15); putting this directory before the
 * reference's libs/ on the include path swaps the implementation without touching the callers.
 *
 * Interface attribution: the names, field order and default values mirrored here are those of MVE's
 * libs/dmrecon public headers, Copyright (C) 2015 Simon Fuhrmann, Ronny Klowsky, TU Darmstadt, distributed under
 * the BSD 3-Clause license (LICENSE.txt of simonfuhrmann/mve).  Only the declarations a caller compiles against
 * are mirrored; the implementation behind them is this repository's.
 */
#ifndef MI_DMRECON_SHIM_DEFINES_H
#define MI_DMRECON_SHIM_DEFINES_H

#include <set>
#include <vector>

#include "math/vector.h"

#define MVS_NAMESPACE_BEGIN namespace mvs {
#define MVS_NAMESPACE_END }

MVS_NAMESPACE_BEGIN
typedef std::set<std::size_t> IndexSet;
MVS_NAMESPACE_END

#endif
