// fresnel_lds.hip -- LDS-resident FFT-convolution engine for the Fresnel propagator (not built yet: the plan
// falls back to the rocFFT engine).
#include "fresnel_plan.hpp"

namespace psx {

bool lds_engine_supported(int, int, int) { return false; }
int lds_engine_create(psx_fresnel_plan *) { return fail(PSX_E_UNSUPPORTED, "LDS engine not available in this build"); }
void lds_engine_destroy(psx_fresnel_plan *) {}
int lds_engine_propagate(psx_fresnel_plan *, const PropArgs &) {
    return fail(PSX_E_UNSUPPORTED, "LDS engine not available in this build");
}

}  // namespace psx
