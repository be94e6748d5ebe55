// Library-level entry points: version and per-thread error text.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void fea_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int fea_version(void) { return 100; }
extern "C" const char* fea_last_error(void) { return g_err; }

// Work-split factor of the persistent kernels (conv_p.hip, wgrad2.hip): they launch `factor` x 256 workgroups, each with
// 1 / factor of the tile range, instead of one workgroup per CU.  1 (default) is fastest on a GPU the step has to itself;
// with a collective's channel kernels holding CUs (data-parallel training: the gradient all-reduce overlaps backward) a
// workgroup that finds no free CU waits for a peer to finish, which stretches a one-per-CU launch towards 2x -- with
// factor 4 to 8 the hardware dispatcher hands the smaller pieces to whichever CUs are free (scripts/cu_steal_test.py).
static int g_wg_split = 1;
extern "C" int fea_set_persistent_split(int factor) {
    FEA_CHECK(factor >= 1 && factor <= 16, FEA_E_BADARG, "fea_set_persistent_split: factor %d not in 1..16", factor);
    g_wg_split = factor;
    return FEA_OK;
}
extern "C" int fea_get_persistent_split(void) { return g_wg_split; }
int fea_wg_split(void) { return g_wg_split; }
