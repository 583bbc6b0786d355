// What the device layer's translation units share besides the public ABI (include/rtx.h).
#pragma once
#include "../../include/rtx.h"

int rtx_internal_device(const rtx_context* ctx);                                    // HIP device of the context
rtx_status rtx_internal_fail(rtx_context* ctx, rtx_status st, const char* message); // records the message, returns st
