// ptmi_render.hip -- wavefront integrator (generate / extend / shade / connect / accumulate).
#include "ptmi_ctx.hpp"

namespace ptmi {
void free_scene(ptmi_ctx *) {}
}  // namespace ptmi
