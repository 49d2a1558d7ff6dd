/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle for the boxtree hot path (tree build + FMM traversal): plain C
 * restatement of the reference algorithm, instantiated for float and double.
 * See boxtree_oracle_impl.h for the provenance/"PARITY UNPINNED" statement.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the resulting shared object.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#define COORD_T double
#define SFX(name) name##_f64
#define COORD_SQRT sqrt
#define COORD_EPS DBL_EPSILON
#define COORD_MAX DBL_MAX
#include "boxtree_oracle_impl.h"
#include "boxtree_oracle_trav_impl.h"
#include "boxtree_oracle_aq_impl.h"
#undef COORD_T
#undef SFX
#undef COORD_SQRT
#undef COORD_EPS
#undef COORD_MAX

#define COORD_T float
#define SFX(name) name##_f32
#define COORD_SQRT sqrtf
#define COORD_EPS FLT_EPSILON
#define COORD_MAX FLT_MAX
#include "boxtree_oracle_impl.h"
#include "boxtree_oracle_trav_impl.h"
#include "boxtree_oracle_aq_impl.h"
#undef COORD_T
#undef SFX
#undef COORD_SQRT
#undef COORD_EPS
#undef COORD_MAX

void orc_free(void *p) { free(p); }

int orc_abi_version(void) { return 1; }

/* threads used by the loops (1 in the sequential build) */
int orc_max_threads(void) { return ORC_NTHREADS(); }
#ifdef _OPENMP
__attribute__((constructor)) static void orc_no_dynamic_teams(void) { omp_set_dynamic(0); }
#endif
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_dynamic(0);            /* the chunking assumes the team it asks for */
    if (n > 0) omp_set_num_threads(n);
#else
    (void) n;
#endif
}
