// RCCL through dlopen: the entry points the sharded indexes use (rxgpu_sharded.hip: float_vector shards; rxgpu_ft_capi.hip: ft_fast
// document-range shards), resolved once by the first index that asks.
#pragma once

#include <string>

#include <rccl/rccl.h>   // types and prototypes only

namespace rxgpu {

// RCCL is opened lazily (dlopen) by the first sharded index that asks for the device-side exchange: a single-GPU deployment neither links
// nor needs librccl.so, and a node where the library is missing or cannot initialise (no peer access, no /dev/shm in the container, ...)
// keeps working on the host-merge path.  The entry points keep their nccl* names below.
struct RcclApi {
	decltype(&::ncclCommInitAll) ncclCommInitAll = nullptr;
	decltype(&::ncclCommDestroy) ncclCommDestroy = nullptr;
	decltype(&::ncclAllGather) ncclAllGather = nullptr;
	decltype(&::ncclAllReduce) ncclAllReduce = nullptr;
	decltype(&::ncclGroupStart) ncclGroupStart = nullptr;
	decltype(&::ncclGroupEnd) ncclGroupEnd = nullptr;
	decltype(&::ncclGetErrorString) ncclGetErrorString = nullptr;
	std::string why;   // non-empty: not available, and why
};

const RcclApi& rccl_api();   // rxgpu_sharded.hip

}  // namespace rxgpu
