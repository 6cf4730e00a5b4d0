// mevp_fused.hip -- placeholder until the fused marching kernel lands (variant 1).
#include "nsdg_internal.h"

int nsdg_launch_mevp_fused(nsdg_ctx*, int, int, int, double, double*, double*, double*, const double*, const double*, double*,
    double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*,
    const double*, const double*)
{
    nsdg_set_error("nsdg_mevp_iterate: variant 1 (fused march) is not built into this library");
    return NSDG_ERR_STATE;
}
