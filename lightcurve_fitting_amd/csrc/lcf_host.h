// Host-side helpers shared by the translation units of liblcf_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
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

// The immutable arrays of an engine go to the device as ONE block with ONE copy: every array of lcf_engine_create used to
// be an allocation and a host-to-device copy of its own (about forty per engine -- 1475 copy kernels in the profile of a
// population of 32 transients).  put() hands out the array's place in the block at once (callers store the pointer in
// the problem) and keeps the bytes in a host image of the block; flush() sends the image.  A block that is full is
// followed by another one.
struct UploadArena {
    std::vector<void*>& owned;   // the device blocks end up here (freed with the engine)
    struct Block { char* dev; std::vector<char> host; size_t used; };
    std::vector<Block> blocks;
    size_t block_bytes;
    explicit UploadArena(std::vector<void*>& o, size_t first_block = 1 << 20) : owned(o), block_bytes(first_block) {}
    lcf_status put(const void* src, size_t bytes, void** dst) {
        *dst = nullptr;
        if (bytes == 0) return LCF_OK;
        const size_t need = (bytes + 255) & ~size_t(255);
        if (blocks.empty() || blocks.back().used + need > blocks.back().host.size()) {
            const size_t cap = std::max(block_bytes, need);
            char* dev = nullptr;
            LCF_HIP(hipMalloc((void**)&dev, cap));
            owned.push_back(dev);
            blocks.push_back(Block{dev, std::vector<char>(cap), 0});
        }
        Block& b = blocks.back();
        std::memcpy(b.host.data() + b.used, src, bytes);
        *dst = b.dev + b.used;
        b.used += need;
        return LCF_OK;
    }
    lcf_status flush() {   // (complete on return)
        for (Block& b : blocks)
            if (b.used) LCF_HIP(hipMemcpy(b.dev, b.host.data(), b.used, hipMemcpyHostToDevice));
        blocks.clear();
        return LCF_OK;
    }
};

template <class T>
lcf_status upload(const std::vector<T>& h, T** d, UploadArena& arena) {
    return arena.put(h.data(), h.size() * sizeof(T), (void**)d);
}

}  // namespace lcf
