// todo.hip -- entry points declared in include/ov2slam_hip.h whose kernels are not
// written yet.  They fail loudly (OV2_EUNSUPPORTED); there is no CPU fallback.
#include "common.hpp"
extern "C" {
void ov2_ba_default_options(ov2_ba_options *o)
{
    if (!o) return;
    o->max_iter = 5; o->function_tolerance = 1e-3; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->huber_delta = 2.4477540725; o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->min_relative_decrease = 1e-3; o->jacobi_scaling = 1;
    o->max_consecutive_invalid_steps = 5;
}
int ov2_ba_solve(ov2_ctx *, const ov2_ba_problem *, const ov2_ba_options *, ov2_ba_result *)
{ ov2_set_error("ov2_ba_solve: not implemented yet"); return OV2_EUNSUPPORTED; }
}
