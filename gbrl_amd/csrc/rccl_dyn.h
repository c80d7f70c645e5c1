// rccl_dyn.h -- RCCL bound at run time (dlopen), so that libgbrl_hip.so has no link-time dependency on a particular RCCL
// and shares the library the process already loaded (PyTorch-ROCm ships its own librccl.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace gbrl {

struct RcclApi {
    // mirrors of the RCCL declarations used (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclReduceScatter, ncclCommDestroy)
    struct UniqueId { char internal[128]; };
    using Comm = void *;
    enum DataType { kInt64 = 4, kFloat32 = 7, kFloat64 = 8 };
    enum RedOp { kSum = 0, kMax = 2, kMin = 3 };
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void *, void *, size_t /*recv count*/, int, int, Comm, hipStream_t) = nullptr;   // optional
    int (*CommDestroy)(Comm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

// Loads RCCL once per process (prefers an already loaded librccl.so); api.ok == false when none is available.
const RcclApi &rccl_api();

}  // namespace gbrl
