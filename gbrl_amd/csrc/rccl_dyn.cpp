#include "rccl_dyn.h"

#include <dlfcn.h>

namespace gbrl {

const RcclApi &rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        void *lib = nullptr;
        // 1. whatever the process already has (PyTorch's bundled RCCL when torch was imported first), 2. the system RCCL
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib)
            for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
        if (!lib) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(lib, "ncclAllReduce"));
        a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(lib, "ncclReduceScatter"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy;
        return a;
    }();
    return api;
}

}  // namespace gbrl
