// Thread-local error text for the C-ABI (include/mreserve_hip.h: mr_last_error).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mreserve_hip.h"

static thread_local char g_err[512] = "";

void mr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mr_last_error(void) { return g_err; }
// 2: mr_gemm_args grew colsum / ldcs, mr_attention_bwd / mr_unit_norm_scale_bwd / mr_contrastive_lse gained arguments (round 2);
// 3: mr_transpose_leaves, mr_set_option("gemm3") (round 3)
extern "C" int mr_version(void) { return 3; }

// ---- process-wide knobs (include/mreserve_hip.h: mr_set_option) ----
int g_mr_opt_tile_n = 0;
int g_mr_opt_v1_only = 0;
int g_mr_opt_group_tile_n = 0;
int g_mr_opt_gemm3 = 1;
int g_mr_opt_gemm3_ph = 0;
int g_mr_opt_gemm4 = -1;
int g_mr_opt_gemm_cus = 0;
int g_mr_opt_gemm5 = -1;
int g_mr_opt_gemm5_stagger = -1;
extern "C" int mr_set_option(const char* name, int value) {
    if (name && !strcmp(name, "gemm_tile_n")) { g_mr_opt_tile_n = value; return MR_OK; }
    if (name && !strcmp(name, "gemm_v1_only")) { g_mr_opt_v1_only = value; return MR_OK; }
    if (name && !strcmp(name, "gemm_group_tile_n")) { g_mr_opt_group_tile_n = value; return MR_OK; }
    if (name && !strcmp(name, "gemm3")) { g_mr_opt_gemm3 = value; return MR_OK; }
    if (name && !strcmp(name, "gemm3_phases")) { g_mr_opt_gemm3_ph = value; return MR_OK; }
    if (name && !strcmp(name, "gemm4")) { g_mr_opt_gemm4 = value; return MR_OK; }
    if (name && !strcmp(name, "gemm_cus")) { g_mr_opt_gemm_cus = value; return MR_OK; }
    if (name && !strcmp(name, "gemm5")) { g_mr_opt_gemm5 = value; return MR_OK; }
    if (name && !strcmp(name, "gemm5_stagger")) { g_mr_opt_gemm5_stagger = value; return MR_OK; }
    mr_set_error("mr_set_option: unknown option '%s'", name ? name : "(null)");
    return MR_EINVAL;
}
