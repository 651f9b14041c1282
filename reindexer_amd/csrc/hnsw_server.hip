// The resident HNSW search kernel (protocol: HnswServer, knn_kernels.hip.h; host side: rxgpu_hnsw_server.hip).
#include "hnsw_search_core.hip.h"

namespace rxgpu {

// The RESIDENT form of the team search (HnswServer, knn_kernels.hip.h): the kernel stays on the chip and workgroup w serves the requests
// planner threads post into slot w of a mailbox in pinned host memory.  The reference's concurrency is T threads with one SearchKnn each
// (float_vector_index.cc:258-294); as launches that is T kernels, T completion signals and — between them — a batcher with its wake-ups:
// 9.9 k q/s at T = 16 over 10M rows where the reference's 16 cores make 18.6 k, although one search is as fast as one core's.  Served from
// the mailbox a call costs what the search costs.
//   * one writer per word (host: post, req, query, stop; device: done, leaving, results), sequence numbers instead of flags: nothing is
//     ever reset, a word is either old or new;
//   * the kernel always ends by itself: workgroup 0 decides — the host's stop word (the index is about to change), no request for
//     idle_ticks, or life_ticks since the start — writes the generation number to `leaving` and raises the leave flag the other workgroups
//     poll in device memory.  No wait in here depends on another kernel or on the host making progress;
//   * the next generation is an ordinary launch on the SAME stream: it starts when this one is gone, so a slot has one server at a time and
//     a request posted while a generation leaves is simply the next one's first.
template <int kMetric, int NB, int kSorted, bool kDel, int kTeam>
__global__ __launch_bounds__(64 * kTeam) void hnsw_server_kernel(HnswParams p, HnswServer sv) {
	__shared__ HnswTeamBox box;
	__shared__ uint32_t s_cmd[4];   // [0] 1 = a request, 2 = leave; [1] its sequence number; [2] k; [3] ef
	const uint32_t slot = blockIdx.x;
	if (threadIdx.x >= 64) {   // the other wavefronts of the team: distance batches of every search until the workgroup leaves
		for (;;) {
			__syncthreads();
			if (s_cmd[0] == 2u) return;
			hnsw_team_serve<kMetric, NB, kTeam>(p, &box);
		}
	}
	const int lane = threadIdx.x;
	const unsigned long long t_start = wall_clock64();
	uint32_t last = __hip_atomic_load(&sv.done[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the host keeps it across generations
	for (;;) {
		if (lane == 0) {
			uint32_t cmd = 0u, seq = last, quiet = 0u;
			while (!cmd) {
				seq = __hip_atomic_load(&sv.post[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				if (seq != last) {
					cmd = 1u;
					break;
				}
				const unsigned long long now = wall_clock64();
				if (slot == 0u) {
					const unsigned long long seen = __hip_atomic_load(&sv.dev[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const unsigned long long since = seen > t_start ? seen : t_start;
					if (__hip_atomic_load(sv.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u || now - t_start > sv.life_ticks ||
						(now > since && now - since > sv.idle_ticks)) {
						__hip_atomic_store(sv.leaving, sv.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
						__hip_atomic_store(&sv.dev[0], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
						cmd = 2u;
					}
				} else if (__hip_atomic_load(&sv.dev[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull || now - t_start > 2ull * sv.life_ticks) {
					cmd = 2u;   // (twice the lifetime: workgroup 0 never came to decide — e.g. it was not resident yet; the kernel ends anyway)
				}
				// every look is a read across PCIe: close together while requests keep coming, ~7 us apart once the slot has been quiet for a while
				if (!cmd) {
					if (++quiet < 64u) {
						__builtin_amdgcn_s_sleep(24);
					} else {
						__builtin_amdgcn_s_sleep(127);
						__builtin_amdgcn_s_sleep(127);
					}
				}
			}
			if (cmd == 1u) {
				s_cmd[2] = __hip_atomic_load(&sv.req[2 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				s_cmd[3] = __hip_atomic_load(&sv.req[2 * slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				(void)__hip_atomic_fetch_max(&sv.dev[1], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			s_cmd[1] = seq;
			s_cmd[0] = cmd;
		}
		__syncthreads();
		if (s_cmd[0] == 2u) return;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // the query the host wrote in front of the sequence number, not a cached line of the slot's last one
		const uint32_t seq = s_cmd[1];
		const unsigned long long t_search = wall_clock64();
		HnswParams pl = p;
		pl.k = s_cmd[2];
		pl.ef = s_cmd[3];
		pl.queries = p.queries + size_t(slot) * p.dim;
		pl.out_dist = p.out_dist + size_t(slot) * sv.kcap;
		pl.out_row = p.out_row + size_t(slot) * sv.kcap;
		pl.out_count = p.out_count + slot;
		hnsw_search_one<kMetric, false, NB, true, false, kSorted, kDel, kTeam>(pl, slot, 0u, &box);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // this wavefront's result stores are out before the sequence number
		if (lane == 0) {
			__hip_atomic_store(&sv.took[slot], uint32_t(wall_clock64() - t_search), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(&sv.done[slot], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			box.cnt = -1;
		}
		last = seq;
		__syncthreads();   // the team leaves hnsw_team_serve
	}
}

// The resident kernel exists for the embedding sizes with a fixed-dimension distance batch and ef <= 256 (224 with deleted nodes); everything
// else keeps the launches.  p: ef_cap / lds_cand_cap as for a team launch, vis_lds_log2 = the size of a hash set in LDS (0: the set is in HBM).
// dynamic LDS of a server workgroup: heaps + query fragment + visited set, then (what fits under 60 KB) the link-block area and the look-ahead area
constexpr size_t kServerLdsLimit = size_t(150) << 10;   // one workgroup per CU: it may take most of the CU's 160 KB (the launcher raises the kernel's limit)
static size_t server_layout(const HnswParams& p, uint32_t* nbl_off, uint32_t* spec_off) {
	size_t at = (size_t(p.ef_cap) + p.lds_cand_cap) * 8 + size_t(p.dim / 64) * 256 + (p.vis_lds_log2 ? (size_t(4) << p.vis_lds_log2) : 0);
	*nbl_off = *spec_off = 0u;
	if (p.nbl && !p.spec && p.maxM0 < 64u && at + kHnswNblBytes <= kServerLdsLimit) {   // (the look-ahead experiment keeps link blocks of its own)
		*nbl_off = uint32_t(at);
		at += kHnswNblBytes;
	}
	if (p.spec && p.bare && p.maxM0 < 64u && at + kHnswSpecBytes <= kServerLdsLimit) {
		*spec_off = uint32_t(at);
		at += kHnswSpecBytes;
	}
	return at;
}
size_t hnsw_server_lds_bytes(const HnswParams& p) {
	uint32_t a, b;
	return server_layout(p, &a, &b);
}
template <int NB, int kSorted, bool kDel>
static void launch_hnsw_server_nb(int metric, const HnswParams& p, const HnswServer& sv, uint32_t slots, hipStream_t s) {
	HnswParams ps = p;
	const size_t lds = server_layout(p, &ps.nbl_off, &ps.spec_off);
#define RX_SERVER(M)                                                                                                                       \
	do {                                                                                                                                   \
		static std::atomic<uint64_t> raised{0};                                                                                            \
		if (lds > (size_t(60) << 10)) (void)raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&hnsw_server_kernel<M, NB, kSorted, kDel, 4>), kServerLdsLimit); \
		hipLaunchKernelGGL((hnsw_server_kernel<M, NB, kSorted, kDel, 4>), dim3(slots), dim3(256), lds, s, ps, sv);                          \
	} while (0)
	switch (metric) {
		case kL2: RX_SERVER(kL2); break;
		case kIP: RX_SERVER(kIP); break;
		default: RX_SERVER(kCos); break;
	}
#undef RX_SERVER
}
template <int NB>
static void launch_hnsw_server_dim(int metric, const HnswParams& p, const HnswServer& sv, uint32_t slots, hipStream_t s) {
	const bool wide = p.ef_cap > 128u;   // the second mailbox of an index: lists of four entries a lane
	if (p.bare) {
		wide ? launch_hnsw_server_nb<NB, 4, false>(metric, p, sv, slots, s) : launch_hnsw_server_nb<NB, 2, false>(metric, p, sv, slots, s);
	} else {
		wide ? launch_hnsw_server_nb<NB, 4, true>(metric, p, sv, slots, s) : launch_hnsw_server_nb<NB, 2, true>(metric, p, sv, slots, s);
	}
}
// Two classes of resident kernel (an index may have one of each, serving a mailbox of its own): ef <= 128 (96 with deleted nodes) — lists of
// two entries a lane, the visited set in LDS — and ef <= 256 (224) — four entries a lane, the visited hash set of a slot in HBM
// (p.visited: [slots][visited_words], zeroed by the search itself).  p.ef_cap says which: 128 or 256.
bool launch_hnsw_server(int metric, const HnswParams& p, const HnswServer& sv, uint32_t slots, hipStream_t s) {
	if (p.codes || !p.sorted || hnsw_server_lds_bytes(p) > kServerLdsLimit) return false;
	if (p.ef_cap <= 128u ? (!p.vis_lds || p.vis_lds_log2 > 14) : (p.vis_lds || p.vis_lds_log2 || !p.visited || p.vis_hash_log2 < 14)) return false;
	switch (p.dim) {
		case 128: launch_hnsw_server_dim<2>(metric, p, sv, slots, s); return true;
		case 512: launch_hnsw_server_dim<8>(metric, p, sv, slots, s); return true;
		case 768: launch_hnsw_server_dim<12>(metric, p, sv, slots, s); return true;
		default: return false;
	}
}

}  // namespace rxgpu
