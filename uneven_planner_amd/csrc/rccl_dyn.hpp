// RCCL, bound at first use.  libunevenhip.so has no link-time dependency on librccl: the single-GPU path never needs it, and a host
// process that already carries an RCCL (a PyTorch-ROCm wheel bundles its own librccl.so next to its own HIP runtime) must keep using
// THAT copy -- two RCCLs on two HIP runtimes in one process do not work.  So: first look for an RCCL that is already mapped
// (RTLD_NOLOAD), only then load the system one.  Everything the sharded map build needs is five entry points.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace uph {

struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    void* handle = nullptr;
    std::string origin;       // which library was bound (diagnostics)
};

// nullptr (and `why` filled) when no RCCL can be found
inline const RcclApi* rcclApi(std::string& why) {
    static RcclApi api;
    static bool ok = false;
    static std::string err;
    static std::once_flag once;          // (several host threads may reach their first multi-GPU call together)
    std::call_once(once, [&]() {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* nm : names) {
            if ((api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD)) != nullptr) { api.origin = std::string(nm) + " (already mapped)"; break; }
        }
        if (!api.handle) {
            const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
            for (const char* nm : paths) {
                if ((api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) != nullptr) { api.origin = nm; break; }
            }
        }
        if (!api.handle) { const char* e = dlerror(); err = std::string("librccl.so not found: ") + (e ? e : "dlopen failed"); }
        else {
#define UPH_RCCL_SYM(field, name) api.field = (decltype(api.field))dlsym(api.handle, name)
            UPH_RCCL_SYM(CommInitAll, "ncclCommInitAll"); UPH_RCCL_SYM(CommDestroy, "ncclCommDestroy"); UPH_RCCL_SYM(AllGather, "ncclAllGather");
            UPH_RCCL_SYM(GroupStart, "ncclGroupStart"); UPH_RCCL_SYM(GroupEnd, "ncclGroupEnd"); UPH_RCCL_SYM(GetErrorString, "ncclGetErrorString");
            UPH_RCCL_SYM(GetVersion, "ncclGetVersion");
#undef UPH_RCCL_SYM
            ok = api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd && api.GetErrorString;
            if (!ok) err = "librccl.so (" + api.origin + ") lacks an expected entry point";
        }
    });
    if (!ok) { why = err; return nullptr; }
    return &api;
}

}  // namespace uph
