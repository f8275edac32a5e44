#include "bcx_internal.h"
int bcx_ensure_gram(bcx_solver* s, int64_t) { return BCX_OK; }
int bcx_launch_apply_omp(bcx_solver* s, const double*) { s->err = "OMP not built"; return BCX_ERR_STATE; }
int bcx_launch_optimize(bcx_solver* s, double) { s->err = "optimize not built"; return BCX_ERR_STATE; }
