// Scans shared by the ft_fast kernels (ft_merge.hip, ft_phrase.hip): wavefront scans and the ordered prefix over all workgroups of a
// launch (decoupled look-back in ticket order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rxgpu_internal.h"

namespace rxgpu {
namespace {

constexpr unsigned long long kLbPrefix = 1ull << 63;
constexpr unsigned long long kLbAggregate = 1ull << 62;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t o = __shfl_up(v, off, 64);
		if (lane >= off) v += o;
	}
	return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

// Exclusive prefix of `count` over ALL threads of ALL workgroups in ticket order (256 threads per workgroup).
// lookback[] is zeroed before the launch; *grand_incl = inclusive total up to and including this workgroup.
__device__ inline uint32_t ordered_prefix(uint32_t count, uint32_t ticket, unsigned long long* lookback, uint32_t* error_flag, uint32_t* grand_incl) {
	__shared__ uint32_t s_wave_tot[4];
	__shared__ uint32_t s_block_excl;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t incl = wave_inclusive_scan(count, lane);
	if (lane == 63) s_wave_tot[wave] = incl;
	__syncthreads();
	uint32_t before = 0;
	for (int w = 0; w < wave; ++w) before += s_wave_tot[w];
	const uint32_t block_total = s_wave_tot[0] + s_wave_tot[1] + s_wave_tot[2] + s_wave_tot[3];
	if (wave == 0) {
		if (lane == 0) {
			__hip_atomic_store(&lookback[ticket], (ticket == 0 ? kLbPrefix : kLbAggregate) | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		uint32_t excl = 0;
		long long j = (long long)ticket - 1;   // nearest predecessor
		while (j >= 0) {
			const long long idx = j - lane;
			unsigned long long st = 0;
			if (idx >= 0) {
				uint32_t spins = 0;
				do {
					st = __hip_atomic_load(&lookback[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (st) break;
					__builtin_amdgcn_s_sleep(1);
					if ((++spins & 1023u) == 0 &&
						(spins > (1u << 24) || __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
						__hip_atomic_store(error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // never hang the GPU: bail out, the host reports it
						st = kLbPrefix;
						break;
					}
				} while (true);
			}
			const unsigned long long pm = __ballot(idx >= 0 && (st & kLbPrefix));
			const int first = pm ? __ffsll((long long)pm) - 1 : 63;
			excl += wave_sum((idx >= 0 && lane <= first) ? uint32_t(st & 0xFFFFFFFFull) : 0u);
			if (pm) break;
			j -= 64;
		}
		if (lane == 0) {
			if (ticket != 0) __hip_atomic_store(&lookback[ticket], kLbPrefix | (unsigned long long)(excl + block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s_block_excl = excl;
		}
	}
	__syncthreads();
	const uint32_t be = s_block_excl;
	*grand_incl = be + block_total;
	__syncthreads();   // the shared words are reused by the caller's next call
	return be + before + (incl - count);
}

__device__ __forceinline__ uint32_t grab_ticket(uint32_t* ticket) {
	__shared__ uint32_t s_ticket;
	if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
	__syncthreads();
	return s_ticket;
}

// block of a posting-side grid -> its sub-term (every entry owns at least one block; entries ascend by block_base)
__device__ __forceinline__ FtGridEntry grid_entry(const FtGridEntry* g, uint32_t n, uint32_t block) {
	uint32_t lo = 0, hi = n - 1;
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (g[mid].block_base <= block) {
			lo = mid;
		} else {
			hi = mid - 1;
		}
	}
	return g[lo];
}


}  // namespace
}  // namespace rxgpu
