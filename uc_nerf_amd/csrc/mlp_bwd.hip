// K6 backward + the fused render backward.  (Placeholder bodies: the symbols exist so the ABI is complete;
// they report an error until the backward kernels land.)
#include "common.h"

using namespace ucnerf;

extern "C" {

int64_t ucnerf_mlp_bwd_workspace_floats(const ucnerf_mlp_config*, int32_t) { return 0; }
int ucnerf_mlp_bwd(const ucnerf_mlp_bwd_params*, void*) { return fail(UCNERF_EINVAL, "mlp_bwd: not implemented yet"); }
int64_t ucnerf_render_bwd_workspace_floats(int32_t, int32_t, int32_t) { return 0; }
int ucnerf_render_fused_bwd(const ucnerf_render_bwd_params*, void*) { return fail(UCNERF_EINVAL, "render_fused_bwd: not implemented yet"); }

}
