// api_common.h -- what every translation unit of the u16 flavour's host side starts with: the status macros and `fail`.
#pragma once
#include <string>

#include "context.h"

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
            return e__ == hipErrorOutOfMemory ? SARPRO_HIP_ERR_OOM : SARPRO_HIP_ERR_HIP;          \
        }                                                                                         \
    } while (0)

#define RETCHK(expr)                                   \
    do {                                               \
        int rc__ = (expr);                             \
        if (rc__ != SARPRO_HIP_OK) return rc__;        \
    } while (0)

static inline int fail(sarpro_hip_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}
