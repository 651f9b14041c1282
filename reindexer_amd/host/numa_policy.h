// ScopedInterleave — page placement for the builder's big arrays on multi-socket hosts.
//
// The concurrent HNSW build (HnswGraph::AddPointConcurrent, the reference's HierarchicalNSWMT build, hnsw.h:60-91) is bound by random
// 3 KB row gathers from every inserting thread.  std::vector zero-fills on the constructing thread, so with the kernel's default
// first-touch policy the whole vector array lands on ONE NUMA node and every thread of the other socket(s) pulls its rows over the
// inter-socket links.  While an instance of this guard lives, pages first touched by the calling thread are interleaved over all
// memory nodes the process may use (set_mempolicy(MPOL_INTERLEAVE)); the previous policy is restored on destruction.  Raw syscalls: the
// image has no libnuma.  Every failure (single node, seccomp, non-Linux) leaves the default policy in place.
// RXGPU_NUMA_INTERLEAVE=0 disables it.
#pragma once

#include <cstddef>
#include <cstdlib>
#include <cstring>

#if defined(__linux__)
#include <sys/syscall.h>
#include <unistd.h>
#endif

namespace rxgpu::host {

class ScopedInterleave {
public:
	explicit ScopedInterleave(size_t bytes, size_t threshold = size_t(256) << 20) noexcept {
#if defined(__linux__) && defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
		if (bytes < threshold) return;
		if (const char* e = std::getenv("RXGPU_NUMA_INTERLEAVE"); e && e[0] == '0') return;
		unsigned long allowed[kWords];
		std::memset(allowed, 0, sizeof(allowed));
		if (syscall(SYS_get_mempolicy, &prevMode_, prevMask_, kBits, nullptr, 0UL) != 0) return;
		if (syscall(SYS_get_mempolicy, nullptr, allowed, kBits, nullptr, kMemsAllowed) != 0) return;
		int nodes = 0;
		for (unsigned long w : allowed) nodes += __builtin_popcountl(w);
		if (nodes < 2) return;
		active_ = syscall(SYS_set_mempolicy, kInterleave, allowed, kBits) == 0;
#else
		(void)bytes;
		(void)threshold;
#endif
	}
	~ScopedInterleave() {
#if defined(__linux__) && defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
		if (active_) syscall(SYS_set_mempolicy, prevMode_, prevMode_ == 0 ? nullptr : prevMask_, prevMode_ == 0 ? 0UL : kBits);
#endif
	}
	ScopedInterleave(const ScopedInterleave&) = delete;
	ScopedInterleave& operator=(const ScopedInterleave&) = delete;
	bool Active() const noexcept { return active_; }

private:
	static constexpr unsigned long kBits = 1024, kWords = kBits / (8 * sizeof(unsigned long));
	static constexpr int kInterleave = 3;              // MPOL_INTERLEAVE
	static constexpr unsigned long kMemsAllowed = 4;   // MPOL_F_MEMS_ALLOWED
	int prevMode_ = 0;
	unsigned long prevMask_[kWords] = {};
	bool active_ = false;
};

}  // namespace rxgpu::host
