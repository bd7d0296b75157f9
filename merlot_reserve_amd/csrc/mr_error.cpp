// Thread-local error text, option sets and handles of the C-ABI (include/mreserve_hip.h: mr_last_error, mr_create / mr_destroy /
// mr_make_current, mr_set_option / mr_get_option, mr_last_gemm_kernel).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mreserve_hip.h"
#include "mr_options.h"

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
// 4: handles (mr_create / mr_destroy / mr_make_current / mr_handle_set_option / mr_handle_get_option), mr_get_option,
//    mr_last_gemm_kernel, options "gemm5" / "gemm5_stagger" / "gemm_trace"; environment knobs only in MR_DEBUG_ENV builds (round 4)
// 5: mr_attention_fwd_dense_mask, mr_masked_lm_xent, mr_crc32c / mr_crc32c_masked / mr_tfrecord_scan, option "attn_tile_modes"; mr_destroy refuses a handle
//    another thread holds (round 5)
// 6: mr_attention_bwd_dense_mask (+ _workspace), options "ln_impl" / "gemm_xpx" / "gemm_xpanel"; the dynamic symbol table is the header's names only (round 6)
extern "C" int mr_version(void) { return 6; }

// ---- option sets: one per handle + the process-wide defaults ----
static MrOptions g_default_opts;
static thread_local mr_handle_s* g_current = nullptr;

MrOptions& mr_opts() { return g_current ? g_current->opt : g_default_opts; }
mr_handle_s* mr_current_handle() { return g_current; }

static int* opt_field(MrOptions& o, const char* name) {
    if (!name) return nullptr;
    if (!strcmp(name, "gemm_tile_n")) return &o.tile_n;
    if (!strcmp(name, "gemm_v1_only")) return &o.v1_only;
    if (!strcmp(name, "gemm_group_tile_n")) return &o.group_tile_n;
    if (!strcmp(name, "gemm3")) return &o.gemm3;
    if (!strcmp(name, "gemm3_phases")) return &o.gemm3_ph;
    if (!strcmp(name, "gemm4")) return &o.gemm4;
    if (!strcmp(name, "gemm_cus")) return &o.gemm_cus;
    if (!strcmp(name, "gemm5")) return &o.gemm5;
    if (!strcmp(name, "gemm5_stagger")) return &o.gemm5_stagger;
    if (!strcmp(name, "attn_onepass")) return &o.attn_onepass;
    if (!strcmp(name, "attn_tile_modes")) return &o.attn_tile_modes;
    if (!strcmp(name, "ln_impl")) return &o.ln_impl;
    if (!strcmp(name, "gemm_xpx")) return &o.xpx;
    if (!strcmp(name, "gemm_xpanel")) return &o.xpanel;
    if (!strcmp(name, "gemm_trace")) return &o.trace;
    return nullptr;
}

extern "C" int mr_create(int32_t device, int64_t ws_bytes, mr_handle* out) {
    if (!out) { mr_set_error("mr_create: null output"); return MR_EINVAL; }
    *out = nullptr;
    if (device < 0 || ws_bytes < 0) { mr_set_error("mr_create: bad device %d / workspace size %ld", (int)device, (long)ws_bytes); return MR_EINVAL; }
    mr_handle_s* h = new mr_handle_s();
    h->device = device;
    h->current_on = 0;
    h->opt = g_default_opts;            // a new handle starts from the process defaults as they are now
    h->ws = nullptr;
    h->ws_bytes = 0;
    if (ws_bytes > 0) {
        int prev = 0;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipMalloc(&h->ws, (size_t)ws_bytes);
        (void)hipSetDevice(prev);
        if (e != hipSuccess) {
            mr_set_error("mr_create: %ld-byte workspace on device %d: %s", (long)ws_bytes, (int)device, hipGetErrorString(e));
            delete h;
            return MR_ELAUNCH;
        }
        h->ws_bytes = ws_bytes;
    }
    *out = h;
    return MR_OK;
}

// A handle may be current on several threads (each launches under its options); `current_on` counts them under g_handle_mu so that mr_destroy
// cannot free a handle (and its workspace) that another thread's next launch would dereference.
static std::mutex g_handle_mu;

extern "C" int mr_destroy(mr_handle h) {
    if (!h) return MR_OK;
    {
        std::lock_guard<std::mutex> lk(g_handle_mu);
        const int others = h->current_on - (g_current == h ? 1 : 0);
        if (others > 0) {
            mr_set_error("mr_destroy: the handle is current on %d other thread(s); mr_make_current(NULL) there first", others);
            return MR_EINVAL;
        }
        if (g_current == h) g_current = nullptr;
    }
    // the caller has synchronised the streams that used the handle; a hipGraph captured while the handle's workspace was substituted into
    // mr_gemm holds that pointer: destroy such graphs first
    if (h->ws) (void)hipFree(h->ws);
    delete h;
    return MR_OK;
}

extern "C" int mr_make_current(mr_handle h) {
    std::lock_guard<std::mutex> lk(g_handle_mu);
    if (g_current == h) return MR_OK;
    if (g_current) --g_current->current_on;
    if (h) ++h->current_on;
    g_current = h;
    return MR_OK;
}
extern "C" mr_handle mr_get_current(void) { return g_current; }

extern "C" int mr_handle_set_option(mr_handle h, const char* name, int32_t value) {
    int* f = opt_field(h ? h->opt : g_default_opts, name);
    if (!f) { mr_set_error("mr_handle_set_option: unknown option '%s'", name ? name : "(null)"); return MR_EINVAL; }
    *f = value;
    return MR_OK;
}

extern "C" int mr_handle_get_option(mr_handle h, const char* name, int32_t* value) {
    int* f = opt_field(h ? h->opt : g_default_opts, name);
    if (!f || !value) { mr_set_error("mr_handle_get_option: unknown option '%s'", name ? name : "(null)"); return MR_EINVAL; }
    *value = *f;
    return MR_OK;
}

// Deprecated shims: the calling thread's current handle, or the process defaults when it has none.
extern "C" int mr_set_option(const char* name, int value) {
    int* f = opt_field(mr_opts(), name);
    if (!f) { mr_set_error("mr_set_option: unknown option '%s'", name ? name : "(null)"); return MR_EINVAL; }
    *f = value;
    return MR_OK;
}
extern "C" int mr_get_option(const char* name, int32_t* value) {
    int* f = opt_field(mr_opts(), name);
    if (!f || !value) { mr_set_error("mr_get_option: unknown option '%s'", name ? name : "(null)"); return MR_EINVAL; }
    *value = *f;
    return MR_OK;
}

// ---- which kernel the last GEMM launch of this thread was routed to (option "gemm_trace") ----
static thread_local char g_route[128] = "";
void mr_note_route(const char* fmt, ...) {
    if (!mr_opts().trace) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_route, sizeof(g_route), fmt, ap);
    va_end(ap);
}
extern "C" const char* mr_last_gemm_kernel(void) { return g_route; }

int mr_env_int(const char* name, int dflt) {
#ifdef MR_DEBUG_ENV
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
