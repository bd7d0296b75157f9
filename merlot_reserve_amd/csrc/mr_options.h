// Tuning / diagnostic option set of the library (include/mreserve_hip.h: mr_create, mr_set_option).  One set per handle; a thread
// launches under the options of ITS current handle (mr_make_current), or under the process-wide defaults when it has none -- which is
// what the deprecated mr_set_option(name, value) writes when the calling thread has no current handle.  Host code only.
#pragma once
#include <stdint.h>

struct MrOptions {
    int tile_n = 0;            // "gemm_tile_n"
    int v1_only = 0;           // "gemm_v1_only"
    int group_tile_n = 0;      // "gemm_group_tile_n"
    int gemm3 = 1;             // "gemm3"
    int gemm3_ph = 0;          // "gemm3_phases"
    int gemm4 = -1;            // "gemm4"
    int gemm_cus = 0;          // "gemm_cus"
    int gemm5 = -1;            // "gemm5"
    int gemm5_stagger = -1;    // "gemm5_stagger"
    int attn_onepass = -1;     // "attn_onepass"
    int attn_tile_modes = 1;   // "attn_tile_modes"
    int xpx = 0;               // "gemm_xpx": force the XCD partition px x (8 / px) of the tile grid (1, 2, 4, 8; 0 = the cost model)
    int xpanel = 0;            // "gemm_xpanel": tile columns per panel of a cell's walk (0 = 8)
    int ln_impl = 1;           // "ln_impl": 1 = the round-6 LayerNorm kernels (whole grid resident), 0 = round 5's
    int trace = 0;             // "gemm_trace": record the kernel every GEMM launch is routed to (mr_last_gemm_kernel)
};

struct mr_handle_s {
    int device;
    int current_on;            // number of threads this handle is current on (mr_make_current): mr_destroy refuses while another thread holds it
    MrOptions opt;
    void* ws;                  // split-K workspace owned by the handle (mr_create's ws_bytes; NULL if 0): used by mr_gemm when the
    int64_t ws_bytes;          // caller's mr_gemm_args.workspace is NULL and this handle is current
};

MrOptions& mr_opts();
mr_handle_s* mr_current_handle();
void mr_note_route(const char* fmt, ...);
// Environment knobs of the experiment scripts (MR_GEMM3, MR_G3_PH, ...): read once per process in DEBUG builds of the library
// (MR_DEBUG_ENV=1 python -m merlot_reserve_amd.build -> -DMR_DEBUG_ENV); the product build ignores the environment.
int mr_env_int(const char* name, int dflt);

#define g_mr_opt_tile_n (mr_opts().tile_n)
#define g_mr_opt_v1_only (mr_opts().v1_only)
#define g_mr_opt_group_tile_n (mr_opts().group_tile_n)
#define g_mr_opt_gemm3 (mr_opts().gemm3)
#define g_mr_opt_gemm3_ph (mr_opts().gemm3_ph)
#define g_mr_opt_gemm4 (mr_opts().gemm4)
#define g_mr_opt_gemm_cus (mr_opts().gemm_cus)
#define g_mr_opt_gemm5 (mr_opts().gemm5)
#define g_mr_opt_gemm5_stagger (mr_opts().gemm5_stagger)
