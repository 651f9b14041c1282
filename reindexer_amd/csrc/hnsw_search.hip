// HNSW search on gfx950: one wavefront per query, traversal IDENTICAL to the reference's.
//
// Replaces (cpp_src/core/index/float_vector/hnswlib/hnswalg.h): getLayer0EntryPoint :799-827, initLayer0SearchState :829-858,
// layer0ShouldStopBeforePop :860-869, runLayer0Step :871-963 (non-streaming), searchBaseLayerST :966-975, SearchKnn :1988-2012.
//
// Why the result can be *equal*, not just "recall-close": every distance is bit-identical to the CPU engine's
// (knn_kernels.hip.h), the graph is the same flat graph, and the two working heaps (PriorityQueue + CompareByFirst,
// priority_queue.h:7-152 / hnswalg.h:581-585) are replayed with the reference's sift mechanics, so ties break the same way.
// The best-first search is sequential per query; parallelism comes from
//   * lanes: the <=2M neighbours of the popped node are fetched/visited-tested one per lane, their distances computed
//     4 rows at a time (16 lanes per row, 16-byte loads), and
//   * queries: thousands of independent wavefronts in flight — the kernel is bound by random 3 KB row gathers from HBM.
// Per query state: candidate heap + result heap in LDS (heap updates by lane 0 in neighbour order), visited bitset in HBM
// (one atomicOr per neighbour is both the test and the mark).  A query whose candidate heap outgrows LDS is flagged and
// re-run with the heap in a global scratch (still on the GPU) — never on the CPU.
// Graphs without deleted nodes, ef <= 256, start differently: both queues as ONE sorted list in registers (HnswSortedList below), and
// only a search in which equal distances could change what the reference's heaps do starts over on the heaps, inside the same kernel.


#include "hnsw_search_core.hip.h"

namespace rxgpu {

template <int kMetric, int NB, int kSorted, bool kDel, int kTeam>
__global__ __launch_bounds__(64 * kTeam) void hnsw_team_kernel(HnswParams p) {
	__shared__ HnswTeamBox box;
	const uint32_t slot = blockIdx.x;
	if (threadIdx.x < 64) {
		hnsw_search_one<kMetric, false, NB, true, false, kSorted, kDel, kTeam>(p, slot, p.only ? p.only[slot] : slot, &box);
		if (threadIdx.x == 0) box.cnt = -1;
		__syncthreads();
	} else {
		hnsw_team_serve<kMetric, NB, kTeam>(p, &box);
	}
}

template <int kMetric, bool kGlobalCand, int NB, bool kLatency, bool kSq8 = false, int kSorted = 0, bool kDel = false>
__global__ __launch_bounds__(64, (kSorted == 2 && !kDel && !kLatency && !kSq8 && NB == 12) ? 5 : 1) void hnsw_search_kernel(HnswParams p) {
	const uint32_t slot = blockIdx.x;
	hnsw_search_one<kMetric, kGlobalCand, NB, kLatency, kSq8, kSorted, kDel>(p, slot, p.only ? p.only[slot] : slot);
	if (p.helper_n) {   // a batch with helper workgroups: this search has ended, and what it queued for them is visible
		__threadfence();
		if (threadIdx.x == 0) atomicAdd(p.helper_n + 2, 1u);
	}
}

// Helper workgroups of a batch: workgroup w serves entries w, w + G, ... of the overflow queue until the stop word is set (a 4-byte memset
// behind the batch's last launch) and its next entry does not exist.  Every search runs on the heap kernel's code with the largest LDS heap
// and the workgroup's own bitset (zeroed here); a search that overflows again keeps kHnswOverflow for the host's global-heap tiers.  A
// wall-clock limit remains as the last resort; the end of the batch itself is a count of finished searches in device memory (HnswHelper::
// finished), which needs no memset or signal to get through a queue the helper may share.
template <int kMetric, int NB, bool kSq8>
__global__ __launch_bounds__(64) void hnsw_helper_kernel(HnswParams p, HnswHelper hq) {
	const uint32_t w = blockIdx.x, G = gridDim.x;
	const int lane = threadIdx.x;
	const unsigned long long t0 = wall_clock64();
	uint4* bits = reinterpret_cast<uint4*>(p.visited + size_t(w) * p.visited_words);
	for (uint32_t s = w; s < hq.cap; s += G) {
		uint32_t id1 = 0;
		for (;;) {
			id1 = __hip_atomic_load(&hq.ids[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
			if (id1) break;
			if (__hip_atomic_load(hq.stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) ||
				__hip_atomic_load(hq.finished, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= hq.expected) {   // the batch is over: the queue is final
				const uint32_t n = __hip_atomic_load(hq.n, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
				if (s >= n) return;
				id1 = __hip_atomic_load(&hq.ids[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
				if (id1) break;
			}
			if (wall_clock64() - t0 > hq.ticks_limit) return;
			__builtin_amdgcn_s_sleep(100);
		}
		id1 = uint32_t(__builtin_amdgcn_readfirstlane(int(id1)));
		for (uint64_t i = lane; i < p.visited_words / 4; i += 64) bits[i] = make_uint4(0u, 0u, 0u, 0u);   // (visited_words is padded to 4)
		__threadfence();
		__syncthreads();
		hnsw_search_one<kMetric, false, NB, (NB > 8 && !kSq8), kSq8, 0, false>(p, w, id1 - 1u);
		__syncthreads();
	}
}

template <bool kGlobalCand, int NB, int kSorted = 0, bool kDel = false>
static void launch_hnsw_nb(int metric, const HnswParams& p, uint32_t blocks, hipStream_t s) {
	size_t lds = (size_t(p.ef_cap) + (kGlobalCand ? 0 : p.lds_cand_cap)) * 8 + size_t(NB) * 256;   // heaps + the query fragment (NB*16 float4)
	constexpr bool kHasLatencyVariant = NB > 8;
	const bool latency = kHasLatencyVariant && blocks <= 3072;   // no more searches than the chip holds of this form (3 per SIMD): spend registers on fewer round trips
	// a handful of searches (the Map's single queries and small coalesced batches): four wavefronts per search (hnsw_team_kernel)
	if constexpr (!kGlobalCand && NB > 0 && kSorted > 0) {
		if (p.team > 1 && blocks <= p.team_max) {
			HnswParams pt = p;
			size_t lds_t = lds;
			// the visited set in LDS, one size up from what the caller allows where that stays within 64 KB (a handful of workgroups: each may take
			// a large part of its CU's 160 KB): at 10M x 768, ef = 128, 0.7 % of the searches mark more than the 4096 nodes a 32 KB set holds, and
			// every one of them costs a re-run launch with a bitset behind the batch
			if (p.vis_lds_log2) {
				uint32_t lg = p.vis_lds_log2;
				if (lg >= 12u && lg < 14u) lg += 1u;   // (smaller sizes are the tests' way of forcing the re-run tiers: left as asked)
				if ((size_t(4) << lg) <= (64u << 10)) {
					pt.vis_lds = 1;
					pt.vis_hash_log2 = lg;
					lds_t += size_t(4) << lg;
				}
			}
			if (p.nbl && !p.spec && p.maxM0 < 64u && lds_t + kHnswNblBytes <= (150u << 10)) {   // the link blocks of a hop's rows come along with the rows (hnsw_team_serve)
				pt.nbl_off = uint32_t(lds_t);
				lds_t += kHnswNblBytes;
			}
			if (p.spec && !kDel && p.maxM0 < 64u && lds_t + kHnswSpecBytes <= (150u << 10)) {   // distances of the next candidate's neighbours ride along (hnsw_search_core.hip.h)
				pt.spec_off = uint32_t(lds_t);
				lds_t += kHnswSpecBytes;
			}
#define RX_TEAM(M)                                                                                                                         \
	do {                                                                                                                                   \
		static std::atomic<uint64_t> raised{0};                                                                                            \
		if (lds_t > (size_t(60) << 10)) (void)raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&hnsw_team_kernel<M, NB, kSorted, kDel, 4>), size_t(150) << 10); \
		hipLaunchKernelGGL((hnsw_team_kernel<M, NB, kSorted, kDel, 4>), dim3(blocks), dim3(256), lds_t, s, pt);                             \
	} while (0)
			switch (metric) {
				case kL2: RX_TEAM(kL2); break;
				case kIP: RX_TEAM(kIP); break;
				default: RX_TEAM(kCos); break;
			}
#undef RX_TEAM
			return;
		}
	}
	// at most two searches per CU and a set of up to 32 KB: the visited set moves into LDS (dynamic LDS stays under the 64 KB a launch gets
	// without an attribute)
	HnswParams pl = p;
	if (latency && !kGlobalCand && p.vis_lds_log2 && blocks <= 512 && (size_t(4) << p.vis_lds_log2) <= (32u << 10) && lds + (size_t(4) << p.vis_lds_log2) <= (60u << 10)) {
		pl.vis_lds = 1;
		pl.vis_hash_log2 = p.vis_lds_log2;
		lds += size_t(4) << p.vis_lds_log2;
	}
#define RX_HNSW(M)                                                                                                                         \
	do {                                                                                                                                   \
		if (latency) {                                                                                                                     \
			hipLaunchKernelGGL((hnsw_search_kernel<M, kGlobalCand, NB, kHasLatencyVariant, false, kSorted, kDel>), dim3(blocks), dim3(64), lds, s, pl); \
		} else {                                                                                                                           \
			hipLaunchKernelGGL((hnsw_search_kernel<M, kGlobalCand, NB, false, false, kSorted, kDel>), dim3(blocks), dim3(64), lds, s, p);    \
		}                                                                                                                                  \
	} while (0)
	switch (metric) {
		case kL2: RX_HNSW(kL2); break;
		case kIP: RX_HNSW(kIP); break;
		default: RX_HNSW(kCos); break;
	}
#undef RX_HNSW
}

template <int kSorted, bool kDel>
static void launch_hnsw_sorted(int metric, const HnswParams& p, uint32_t blocks, hipStream_t s) {
	switch (p.dim) {
		case 128: launch_hnsw_nb<false, 2, kSorted, kDel>(metric, p, blocks, s); break;
		case 512: launch_hnsw_nb<false, 8, kSorted, kDel>(metric, p, blocks, s); break;
		case 768: launch_hnsw_nb<false, 12, kSorted, kDel>(metric, p, blocks, s); break;
		default: launch_hnsw_nb<false, 0, kSorted, kDel>(metric, p, blocks, s); break;
	}
}

template <bool kGlobalCand>
static void launch_hnsw_mode(int metric, const HnswParams& p, uint32_t blocks, hipStream_t s) {
	switch (p.dim) {
		case 128: launch_hnsw_nb<kGlobalCand, 2>(metric, p, blocks, s); break;
		case 512: launch_hnsw_nb<kGlobalCand, 8>(metric, p, blocks, s); break;
		case 768: launch_hnsw_nb<kGlobalCand, 12>(metric, p, blocks, s); break;
		default: launch_hnsw_nb<kGlobalCand, 0>(metric, p, blocks, s); break;
	}
}

template <bool kGlobalCand, int NB, int kSorted = 0, bool kDel = false>
static void launch_hnsw_sq8_nb(int metric, const HnswParams& p, uint32_t blocks, hipStream_t s) {
	const size_t lds = (size_t(p.ef_cap) + (kGlobalCand ? 0 : p.lds_cand_cap)) * 8;
	switch (metric) {
		case kL2: hipLaunchKernelGGL((hnsw_search_kernel<kL2, kGlobalCand, NB, false, true, kSorted, kDel>), dim3(blocks), dim3(64), lds, s, p); break;
		case kIP: hipLaunchKernelGGL((hnsw_search_kernel<kIP, kGlobalCand, NB, false, true, kSorted, kDel>), dim3(blocks), dim3(64), lds, s, p); break;
		default: hipLaunchKernelGGL((hnsw_search_kernel<kCos, kGlobalCand, NB, false, true, kSorted, kDel>), dim3(blocks), dim3(64), lds, s, p); break;
	}
}

template <int kSorted, bool kDel>
static void launch_hnsw_sq8_sorted(int metric, const HnswParams& p, uint32_t blocks, hipStream_t s) {
	if constexpr (kDel) {   // graphs with deleted nodes: two embedding sizes get the 16-byte-load form, the rest the generic one
		switch (p.dim) {
			case 128: launch_hnsw_sq8_nb<false, 2, kSorted, true>(metric, p, blocks, s); break;
			case 768: launch_hnsw_sq8_nb<false, 12, kSorted, true>(metric, p, blocks, s); break;
			default: launch_hnsw_sq8_nb<false, 0, kSorted, true>(metric, p, blocks, s); break;
		}
	} else {
		switch (p.dim) {
			case 128: launch_hnsw_sq8_nb<false, 2, kSorted>(metric, p, blocks, s); break;
			case 384: launch_hnsw_sq8_nb<false, 6, kSorted>(metric, p, blocks, s); break;
			case 512: launch_hnsw_sq8_nb<false, 8, kSorted>(metric, p, blocks, s); break;
			case 768: launch_hnsw_sq8_nb<false, 12, kSorted>(metric, p, blocks, s); break;
			case 1024: launch_hnsw_sq8_nb<false, 16, kSorted>(metric, p, blocks, s); break;
			case 1536: launch_hnsw_sq8_nb<false, 24, kSorted>(metric, p, blocks, s); break;
			default: launch_hnsw_sq8_nb<false, 0, kSorted>(metric, p, blocks, s); break;
		}
	}
}

template <bool kGlobalCand>
static void launch_hnsw_sq8(int metric, const HnswParams& p, uint32_t blocks, hipStream_t s) {
	switch (p.dim) {   // the embedding sizes in use get the 16-byte-load form; any other dim the generic one
		case 128: launch_hnsw_sq8_nb<kGlobalCand, 2>(metric, p, blocks, s); break;
		case 384: launch_hnsw_sq8_nb<kGlobalCand, 6>(metric, p, blocks, s); break;
		case 512: launch_hnsw_sq8_nb<kGlobalCand, 8>(metric, p, blocks, s); break;
		case 768: launch_hnsw_sq8_nb<kGlobalCand, 12>(metric, p, blocks, s); break;
		case 1024: launch_hnsw_sq8_nb<kGlobalCand, 16>(metric, p, blocks, s); break;
		case 1536: launch_hnsw_sq8_nb<kGlobalCand, 24>(metric, p, blocks, s); break;
		default: launch_hnsw_sq8_nb<kGlobalCand, 0>(metric, p, blocks, s); break;
	}
}

void launch_hnsw_search(int metric, const HnswParams& p, uint32_t blocks, bool global_cand, hipStream_t s) {
	if (p.sorted && !global_cand) {
		// both queues as one sorted list in registers.  Without deleted nodes: 2 entries a lane up to ef = 128, 4 up to 256; with deleted
		// nodes the list also holds the deleted candidates in reach: 2 entries a lane up to ef = 96, 3 up to 160, 4 up to 224 (kHnswSortedMaxEfDel)
		if (p.bare) {
			const bool four = p.ef > 128;
			if (p.codes) {
				four ? launch_hnsw_sq8_sorted<4, false>(metric, p, blocks, s) : launch_hnsw_sq8_sorted<2, false>(metric, p, blocks, s);
			} else {
				four ? launch_hnsw_sorted<4, false>(metric, p, blocks, s) : launch_hnsw_sorted<2, false>(metric, p, blocks, s);
			}
		} else {
			const int slots = p.ef <= 96 ? 2 : p.ef <= 160 ? 3 : 4;
			if (p.codes) {
				slots == 2 ? launch_hnsw_sq8_sorted<2, true>(metric, p, blocks, s)
						   : slots == 3 ? launch_hnsw_sq8_sorted<3, true>(metric, p, blocks, s) : launch_hnsw_sq8_sorted<4, true>(metric, p, blocks, s);
			} else {
				slots == 2 ? launch_hnsw_sorted<2, true>(metric, p, blocks, s)
						   : slots == 3 ? launch_hnsw_sorted<3, true>(metric, p, blocks, s) : launch_hnsw_sorted<4, true>(metric, p, blocks, s);
			}
		}
		return;
	}
	if (p.codes) {   // SQ8 graph
		if (global_cand) {
			launch_hnsw_sq8<true>(metric, p, blocks, s);
		} else {
			launch_hnsw_sq8<false>(metric, p, blocks, s);
		}
		return;
	}
	if (global_cand) {
		launch_hnsw_mode<true>(metric, p, blocks, s);
	} else {
		launch_hnsw_mode<false>(metric, p, blocks, s);
	}
}


template <int NB, bool kSq8>
static void launch_hnsw_helper_nb(int metric, const HnswParams& p, const HnswHelper& hq, uint32_t groups, hipStream_t s) {
	const size_t lds = (size_t(p.ef_cap) + p.lds_cand_cap) * 8 + (kSq8 ? 0 : size_t(NB) * 256);
	switch (metric) {
		case kL2: hipLaunchKernelGGL((hnsw_helper_kernel<kL2, NB, kSq8>), dim3(groups), dim3(64), lds, s, p, hq); break;
		case kIP: hipLaunchKernelGGL((hnsw_helper_kernel<kIP, NB, kSq8>), dim3(groups), dim3(64), lds, s, p, hq); break;
		default: hipLaunchKernelGGL((hnsw_helper_kernel<kCos, NB, kSq8>), dim3(groups), dim3(64), lds, s, p, hq); break;
	}
}

// the helper workgroups of a batch (hnsw_helper_kernel): p = the batch's parameters with the helpers' own visited bitsets, the largest LDS
// heap, no `only` list, no overflow queue of their own
void launch_hnsw_helper(int metric, const HnswParams& p, const HnswHelper& hq, uint32_t groups, hipStream_t s) {
	if (p.codes) {
		switch (p.dim) {
			case 128: launch_hnsw_helper_nb<2, true>(metric, p, hq, groups, s); break;
			case 768: launch_hnsw_helper_nb<12, true>(metric, p, hq, groups, s); break;
			default: launch_hnsw_helper_nb<0, true>(metric, p, hq, groups, s); break;
		}
		return;
	}
	switch (p.dim) {
		case 128: launch_hnsw_helper_nb<2, false>(metric, p, hq, groups, s); break;
		case 768: launch_hnsw_helper_nb<12, false>(metric, p, hq, groups, s); break;
		default: launch_hnsw_helper_nb<0, false>(metric, p, hq, groups, s); break;
	}
}

// SearchRange's second half on the device (hnswalg.h:2030-2064): the closure of the ef-search's hits under "neighbour on level 0, not
// deleted, dist < radius".  The result is a SET (the reference's radius_queue order does not change it), so the expansion runs level by
// level: one workgroup of 16 wavefronts, every wavefront takes frontier nodes — the node's list block in one round, test-and-mark on a
// fresh visited bitset (atomicOr; the reference starts a new visited list too: only the ef hits are marked), distances of the fresh
// neighbours 4 rows per step — and appends what lies inside the radius to the result and to the next frontier.
constexpr int kRangeThreads = 1024;
constexpr int kRangeWaves = kRangeThreads / 64;

template <int kMetric>
__global__ __launch_bounds__(kRangeThreads) void hnsw_range_kernel(HnswParams p, HnswRange r) {
	__shared__ uint32_t nb_id[kRangeWaves][kHnswMaxNeighbors];
	__shared__ float nb_d[kRangeWaves][kHnswMaxNeighbors];
	__shared__ uint32_t s_na, s_nb;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const bool sq8 = p.codes != nullptr;
	uint32_t* fa = r.frontier;
	uint32_t* fb = r.frontier + r.cap;
	if (tid == 0) {
		s_na = 0;
		s_nb = 0;
	}
	__syncthreads();
	// the ef hits: all marked visited, those inside the radius are results and the first frontier
	for (uint32_t i = tid; i < r.seed_n; i += kRangeThreads) {
		const uint32_t id = r.seed_row[i];
		const float d = r.seed_dist[i];
		atomicOr(&r.visited[id >> 5], 1u << (id & 31));
		if (d < r.radius) {
			const unsigned long long pos = atomicAdd(r.total, 1ull);
			if (pos < r.cap) {
				r.out_dist[pos] = d;
				r.out_row[pos] = id;
				fa[atomicAdd(&s_na, 1u)] = id;
			}
		}
	}
	__syncthreads();
	for (;;) {
		const uint32_t na = s_na;
		if (na == 0) break;
		for (uint32_t i = wave; i < na; i += kRangeWaves) {
			const uint32_t node = fa[i];
			const uint32_t* ll = p.links0 + size_t(node) * (1 + p.maxM0);
			int nfresh = 0, cnt = 0;
			for (int base = 0; base <= int(p.maxM0); base += 64) {
				const int w = base + lane;
				const uint32_t word = w <= int(p.maxM0) ? ll[w] : 0u;
				if (base == 0) cnt = int(__builtin_amdgcn_readfirstlane(word));
				if (base > cnt) break;   // uniform
				const int j = w - 1;
				bool fresh = false;
				if (j >= 0 && j < cnt && (p.bare || !p.deleted[word])) {   // IsMarkedDeleted first: a deleted neighbour is not even marked
					const uint32_t bit = 1u << (word & 31);
					fresh = !(atomicOr(&r.visited[word >> 5], bit) & bit);
				}
				const uint64_t fm = __ballot(fresh);
				if (fresh) nb_id[wave][nfresh + __popcll(fm & ((1ull << lane) - 1))] = word;
				nfresh += __popcll(fm);
			}
			__builtin_amdgcn_wave_barrier();
			if (sq8) {
				batch_distances_sq8<kMetric>(p, p.qcodes, p.qcorr[0], p.qnorm[0], nb_id[wave], nfresh, nb_d[wave], lane);
			} else {
				batch_distances<kMetric>(p, p.queries, nb_id[wave], nfresh, nb_d[wave], lane);
			}
			__builtin_amdgcn_wave_barrier();
			for (int j = lane; j < nfresh; j += 64) {
				const float d = nb_d[wave][j];
				if (d < r.radius) {
					const unsigned long long pos = atomicAdd(r.total, 1ull);
					if (pos < r.cap) {
						r.out_dist[pos] = d;
						r.out_row[pos] = nb_id[wave][j];
						fb[atomicAdd(&s_nb, 1u)] = nb_id[wave][j];   // at most cap entries ever enter a frontier: one per stored result
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
		__syncthreads();
		if (tid == 0) {
			s_na = s_nb;
			s_nb = 0;
		}
		uint32_t* t = fa;
		fa = fb;
		fb = t;
		__syncthreads();
	}
}

void launch_hnsw_range(int metric, const HnswParams& p, const HnswRange& r, hipStream_t s) {
	switch (metric) {
		case kL2: hipLaunchKernelGGL(hnsw_range_kernel<kL2>, dim3(1), dim3(kRangeThreads), 0, s, p, r); break;
		case kIP: hipLaunchKernelGGL(hnsw_range_kernel<kIP>, dim3(1), dim3(kRangeThreads), 0, s, p, r); break;
		default: hipLaunchKernelGGL(hnsw_range_kernel<kCos>, dim3(1), dim3(kRangeThreads), 0, s, p, r); break;
	}
}

// rxgpu_hnsw_patch_graph: a host insert (addPoint / updatePoint, hnswalg.h:1472-1852) rewrites the lists of the new element and of a few
// dozen neighbours; their staged copies are scattered into the resident graph instead of re-uploading all of it.
__global__ __launch_bounds__(64) void hnsw_patch_kernel(HnswPatch p) {
	const uint32_t j = blockIdx.x, lane = threadIdx.x;
	const uint32_t id = p.ids[j];
	const uint32_t stride0 = 1 + p.maxM0, stride = 1 + p.M;
	for (uint32_t w = lane; w < stride0; w += 64) p.links0[uint64_t(id) * stride0 + w] = p.src_links0[uint64_t(j) * stride0 + w];
	uint64_t at = p.upper_at[j];
	if (at == ~uint64_t(0)) {
		at = p.upper_off[id];
	} else if (lane == 0) {
		p.upper_off[id] = at;
	}
	const uint32_t words = uint32_t(p.levels[j]) * stride;
	for (uint32_t w = lane; w < words; w += 64) p.upper[at * stride + w] = p.src_upper[uint64_t(p.upper_src[j]) * stride + w];
	if (lane == 0) p.deleted[id] = p.src_deleted[j];
}

void launch_hnsw_patch(const HnswPatch& p, uint32_t n_dirty, hipStream_t s) {
	if (n_dirty) hipLaunchKernelGGL(hnsw_patch_kernel, dim3(n_dirty), dim3(64), 0, s, p);
}

}  // namespace rxgpu
