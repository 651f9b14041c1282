// RCCL through dlopen: the entry points the sharded indexes use (rxgpu_sharded.hip: float_vector shards; rxgpu_ft_capi.hip: ft_fast
// document-range shards), resolved once by the first index that asks.
#pragma once

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>   // types and prototypes only

namespace rxgpu {

// RCCL is opened lazily (dlopen) by the first sharded index that asks for the device-side exchange: a single-GPU deployment neither links
// nor needs librccl.so, and a node where the library is missing or cannot initialise (no peer access, no /dev/shm in the container, ...)
// keeps working on the host-merge path.  The entry points keep their nccl* names below.
struct RcclApi {
	decltype(&::ncclCommInitAll) ncclCommInitAll = nullptr;
	decltype(&::ncclAllGather) ncclAllGather = nullptr;
	decltype(&::ncclGroupStart) ncclGroupStart = nullptr;
	decltype(&::ncclGroupEnd) ncclGroupEnd = nullptr;
	decltype(&::ncclGetErrorString) ncclGetErrorString = nullptr;
	std::string why;   // non-empty: not available, and why
};

const RcclApi& rccl_api();   // rxgpu_sharded.hip

// The communicators over one list of distinct devices, shared by every sharded index of the process over that list and kept until the
// process ends: ncclCommInitAll costs hundreds of milliseconds, proxy threads and device buffers per communicator, and namespaces (and
// tests) create and drop indexes all the time — an index takes the set that is there instead of building and tearing down its own.
// Collectives of one communicator are enqueued in one order: `mtx` is held from ncclGroupStart to ncclGroupEnd (the enqueue, not the run).
struct RcclCommSet {
	std::vector<int> devices;
	std::vector<ncclComm_t> comms;   // comms[r] on devices[r]
	std::mutex mtx;
};
// nullptr + *why when RCCL is not there or ncclCommInitAll fails (the caller falls back to the host path)
std::shared_ptr<RcclCommSet> rccl_comm_set(const std::vector<int>& devices, std::string* why);   // rxgpu_sharded.hip

}  // namespace rxgpu
