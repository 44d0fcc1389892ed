// Host-side helpers shared by the translation units of liblcf_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "lcf.h"

namespace lcf {

extern thread_local std::string g_err;
lcf_status fail(lcf_status st, const std::string& msg);

#define LCF_HIP(call)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return lcf::fail(e_ == hipErrorOutOfMemory ? LCF_ERR_OUT_OF_MEMORY : LCF_ERR_HIP,              \
                             std::string(#call) + ": " + hipGetErrorString(e_));                           \
    } while (0)

template <class T>
lcf_status upload(const std::vector<T>& h, T** d, std::vector<void*>& owned) {
    *d = nullptr;
    if (h.empty()) return LCF_OK;
    LCF_HIP(hipMalloc((void**)d, h.size() * sizeof(T)));
    owned.push_back(*d);
    LCF_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return LCF_OK;
}

}  // namespace lcf
