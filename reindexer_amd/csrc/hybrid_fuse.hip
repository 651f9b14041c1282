// Hybrid FT + KNN rank fusion ON THE DEVICE: the step behind both engines in `WHERE ft = '...' OR/AND KNN(...)
// ORDER BY RRF() / rank expression`.  Replaces, for one query,
//   RanksHolder::InitRRFPositions                 cpp_src/core/nsselecter/ranks_holder.h:61-76
//   RerankerRRF / RerankerLinear                  cpp_src/core/sorting/reranker.h:11-39
//   MergerRankedImpl::operator() + mergeRanked    cpp_src/core/nsselecter/selectiteratorcontainer.cc:1343-1423, 1454-1559
//   Merged<desc> = set<IdRank<desc>> ordering      selectiteratorcontainer.cc:1258-1283  (desc: rank descending, ties by DESCENDING id)
// and, when the FT side comes straight from the merge train, Merger::postProcessResults (merger.h:111-140: drop proc < minRank, scale to
// 0..255, uint8 truncation) — so that the FT result (HBM, ft_finish's output) and the KNN result (HBM, the scan's (dist, row) list) are
// fused where they lie and ONE list of (id, rank) leaves the device.
//
// The structure the kernel uses: an FT rank is a uint8 (normalizedProc), so the FT side has at most 256 rank classes.  The RRF position of
// a class is 1 + #documents in better classes (equal ranks share the position of their run's first element); a document found by the FT
// side only gets a fused rank that depends on its class alone; classes whose fused ranks are equal floats form one group; inside a group
// Merged<desc> orders by id.  Hence:  final order of the FT-only tail = stable split by group of the id-ordered FT list.  The (at most
// 1024) documents that came through the KNN list get their ranks individually, are ordered by counting, and the two ordered lists are
// merged by position arithmetic.  One workgroup of 1024 threads: the data is a few hundred KB and every step is a dependent pass — more
// workgroups would only add grid-wide synchronisation.
//
//   pass 1   FT: max proc, postProcess -> (id, class); class histogram; max id
//   pass 2   FT: LSD radix sort by id (8-bit digits, as many passes as the largest id has bytes; wave-private tiles, stable)
//   pass 3   KNN ids into an LDS hash table; FT probes it (marks itself "also in the KNN list", leaves its index + class)
//   pass 4   KNN entries: position of the equal-rank run, fused rank, order by counting, exact (rank, id) repeats dropped
//   pass 5   FT tail: one more stable radix pass by group (in-KNN / dropped documents fall off the end)
//   pass 6   both lists written to their final places
// Bound: latency (a chain of ~12 dependent passes over L2-resident data); algorithmic bytes ~ (n_ft + k) * 40 B.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "rxgpu_internal.h"
#include "knn_kernels.hip.h"

namespace rxgpu {

namespace {

constexpr int kFuseThreads = 1024;
constexpr int kFuseWaves = kFuseThreads / 64;
constexpr int kDigits = 257;             // 256 values + "falls off the end"
constexpr uint32_t kHashSlots = 4096;    // >= 4 x the largest KNN list
constexpr uint16_t kClsDropped = 0xFFFF;
constexpr uint16_t kClsInKnn = 0x0100;   // flag on a class: the document also came through the KNN list

__device__ __forceinline__ uint32_t sortable_bits(float v) {
	const float r = v + 0.0f;   // -0 -> +0: the reference compares floats, both zeros are one rank
	const uint32_t u = __float_as_uint(r);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct FuseShared {
	uint32_t hist[kFuseWaves * kDigits];   // radix passes: per-wave digit counters / running offsets
	uint32_t tot[kDigits + 7];             // digit totals -> exclusive prefix
	uint32_t cls_count[256];               // FT documents per rank class (after postProcess)
	uint32_t cls_pos[256];                 // RRF position of a class: 1 + #documents in better classes
	uint32_t cls_key[256];                 // fused rank of an FT-only document of the class, as an order key (smaller = earlier)
	float cls_rank[256];                   // ... and as the float that is returned
	uint32_t cls_group[256];               // group digit: #distinct better keys among the classes
	uint32_t cls_first[256];               // the class holds documents and no lower class has its key
	uint32_t grp_key[256];                 // key of group g (0xFFFFFFFF: empty)
	uint32_t grp_start[kDigits + 1];       // first tail index of group g
	int32_t hash_id[kHashSlots];           // KNN ids (open addressing), -1 = empty
	uint32_t hash_ft[kHashSlots];          // FT index | class << 24 found for that id, 0xFFFFFFFF = not in the FT list
	int32_t k_id[kMaxFuseKnn];
	float k_rank[kMaxFuseKnn];             // KNN rank as the caller sees it (L2: distance, IP / cosine: -distance)
	uint32_t k_key[kMaxFuseKnn];           // fused order key; 0xFFFFFFFF + k_keep = 0: not part of the result
	float k_fused[kMaxFuseKnn];
	uint8_t k_keep[kMaxFuseKnn];
	uint32_t h_key[kMaxFuseKnn];           // head in final order
	int32_t h_id[kMaxFuseKnn];
	float h_rank[kMaxFuseKnn];
	uint32_t red_u[kFuseWaves];
	float red_f[kFuseWaves];
	uint32_t n_ft, n_knn, n_head, n_tail, max_id, id_passes;
	float scale;
};

__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, uint32_t* red) {
	for (int off = 32; off > 0; off >>= 1) v = max(v, uint32_t(__shfl_xor(int(v), off, 64)));
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	uint32_t r = red[0];
	for (int w = 1; w < kFuseWaves; ++w) r = max(r, red[w]);
	__syncthreads();
	return r;
}
__device__ __forceinline__ float block_max_f32(float v, float* red) {
	for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	float r = red[0];
	for (int w = 1; w < kFuseWaves; ++w) r = fmaxf(r, red[w]);
	__syncthreads();
	return r;
}

// One stable LSD pass over n (key, class) pairs: element i moves to the slot of its digit.  Every wavefront owns a contiguous chunk and
// walks it in tiles of 64 in order, so "earlier in the input" == (earlier wave, earlier tile, lower lane) and the pass is stable.
template <typename DigitFn>
__device__ void radix_pass(FuseShared& s, const uint32_t* kin, const uint16_t* cin, uint32_t* kout, uint16_t* cout, uint32_t n, DigitFn digit) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t chunk = ((n + kFuseWaves - 1) / kFuseWaves + 63) & ~63u;
	const uint32_t begin = min(n, wave * chunk), end = min(n, begin + chunk);
	for (int i = threadIdx.x; i < kFuseWaves * kDigits; i += kFuseThreads) s.hist[i] = 0;
	__syncthreads();
	for (uint32_t i = begin + lane; i < end; i += 64) atomicAdd(&s.hist[wave * kDigits + digit(kin[i], cin[i])], 1u);
	__syncthreads();
	if (threadIdx.x < kDigits) {   // per digit: counts of the waves -> exclusive prefix over the waves, total aside
		uint32_t sum = 0;
		for (int w = 0; w < kFuseWaves; ++w) {
			const uint32_t v = s.hist[w * kDigits + threadIdx.x];
			s.hist[w * kDigits + threadIdx.x] = sum;
			sum += v;
		}
		s.tot[threadIdx.x] = sum;
	}
	__syncthreads();
	if (wave == 0) {   // exclusive prefix of the digit totals: 5 per lane + a wave scan
		uint32_t v[5], local = 0;
#pragma unroll
		for (int j = 0; j < 5; ++j) {
			const int d = lane * 5 + j;
			v[j] = d < kDigits ? s.tot[d] : 0;
			local += v[j];
		}
		uint32_t incl = local;
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t o = __shfl_up(incl, off, 64);
			if (lane >= off) incl += o;
		}
		uint32_t run = incl - local;
#pragma unroll
		for (int j = 0; j < 5; ++j) {
			const int d = lane * 5 + j;
			if (d < kDigits) s.tot[d] = run;
			run += v[j];
		}
	}
	__syncthreads();
	if (threadIdx.x < kDigits) {
		const uint32_t base = s.tot[threadIdx.x];
		for (int w = 0; w < kFuseWaves; ++w) s.hist[w * kDigits + threadIdx.x] += base;
	}
	__syncthreads();
	const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
	uint32_t* off_w = &s.hist[wave * kDigits];
	for (uint32_t t = begin; t < end; t += 64) {
		const uint32_t i = t + lane;
		const bool valid = i < end;
		const uint32_t key = valid ? kin[i] : 0;
		const uint16_t cls = valid ? cin[i] : uint16_t(0);
		const uint32_t d = valid ? digit(key, cls) : 0;
		uint64_t peers = __ballot(valid);
#pragma unroll
		for (int b = 0; b < 9; ++b) {
			const uint64_t m = __ballot((d >> b) & 1u);
			peers &= ((d >> b) & 1u) ? m : ~m;
		}
		if (valid) {
			const uint32_t rank = __popcll(peers & lt), cnt = __popcll(peers);
			const uint32_t off = off_w[d];
			if (rank + 1 == cnt) off_w[d] = off + cnt;   // the last lane of the digit's peers moves the wave's running offset
			kout[off + rank] = key;
			cout[off + rank] = cls;
		}
		__builtin_amdgcn_wave_barrier();
	}
	__syncthreads();
}

__device__ __forceinline__ uint32_t hash_slot(int32_t id) { return (uint32_t(id) * 2654435761u) >> 20; }   // 12 bits

}  // namespace

__global__ __launch_bounds__(kFuseThreads) void hybrid_fuse_kernel(HybridFuseArgs a) {
	extern __shared__ __align__(16) unsigned char fuse_lds[];
	FuseShared& s = *reinterpret_cast<FuseShared*>(fuse_lds);
	const int tid = threadIdx.x;
	const bool desc = a.desc != 0, rrf = a.kind == 0, is_union = a.is_union != 0;
	auto rank_key = [desc](float r) { return desc ? ~sortable_bits(r) : sortable_bits(r); };
	auto id_before = [desc](int32_t l, int32_t r) { return desc ? l > r : l < r; };
	uint32_t* keyA = a.scratch_key;
	uint32_t* keyB = a.scratch_key + a.ft_cap;
	uint16_t* clsA = a.scratch_cls;
	uint16_t* clsB = a.scratch_cls + a.ft_cap;

	if (tid == 0) {
		uint32_t n = a.ft_count_ptr ? *a.ft_count_ptr : a.ft_n;
		s.n_ft = min(n, a.ft_cap);
		uint32_t nk = a.knn_count_ptr ? min(*a.knn_count_ptr, a.knn_n) : a.knn_n;
		s.n_knn = min(min(nk, a.k), uint32_t(kMaxFuseKnn));
	}
	for (int i = tid; i < 256; i += kFuseThreads) s.cls_count[i] = 0;
	for (int i = tid; i < int(kHashSlots); i += kFuseThreads) {
		s.hash_id[i] = -1;
		s.hash_ft[i] = 0xFFFFFFFFu;
	}
	__syncthreads();
	const uint32_t n = s.n_ft, nk = s.n_knn;

	// ---- pass 1: postProcessResults (merger.h:111-140) when the FT side is the merge train's raw output; ids, classes, histogram, max id
	float scale = 1.0f;
	if (a.ft_proc) {
		float mx = 0.0f;
		for (uint32_t i = tid; i < n; i += kFuseThreads) mx = fmaxf(mx, a.ft_proc[i]);
		mx = block_max_f32(mx, s.red_f);
		scale = mx > 255.0f ? float(255.0 / double(mx)) : 1.0f;
	}
	uint32_t max_id = 0;
	for (uint32_t i = tid; i < n; i += kFuseThreads) {
		uint16_t cls;
		if (a.ft_proc) {
			const float proc = a.ft_proc[i];
			cls = proc < a.min_rank ? kClsDropped : uint16_t(uint8_t(proc * scale));
		} else {
			cls = a.ft_rank_u8[i];
		}
		const uint32_t doc = a.ft_doc[i];
		const uint32_t id = a.row_of_doc ? uint32_t(a.row_of_doc[doc]) : doc;
		keyA[i] = id;
		clsA[i] = cls;
		if (cls != kClsDropped) {
			atomicAdd(&s.cls_count[cls], 1u);
			max_id = max(max_id, id);
		}
	}
	max_id = block_max_u32(max_id, s.red_u);
	if (tid == 0) {
		s.max_id = max_id;
		s.id_passes = max_id ? (32 - __clz(max_id) + 7) / 8 : 1;
	}
	if (tid < 256) {   // RRF position of a class (InitRRFPositions: equal ranks share the 1-based position of their run's first element)
		uint32_t better = 0;
		for (int c = tid + 1; c < 256; ++c) better += s.cls_count[c];
		s.cls_pos[tid] = 1 + better;
	}
	__syncthreads();
	if (tid < 256) {   // fused rank of an FT-only document of class tid (reranker.h: CalculateSingle / CalculateJustFt)
		float f;
		if (rrf) {
			f = float(1.0 / (a.params[0] + double(s.cls_pos[tid])));
		} else {
			f = float(a.params[0] * a.params[1] + a.params[2] * double(float(tid)) + a.params[4]);
		}
		s.cls_rank[tid] = f;
		s.cls_key[tid] = rank_key(f);
	}
	__syncthreads();
	if (tid < 256) {   // classes with equal fused ranks interleave by id, so they share a group: one representative per distinct key
		bool first = s.cls_count[tid] != 0;
		for (int c2 = 0; c2 < tid && first; ++c2) first = !(s.cls_count[c2] && s.cls_key[c2] == s.cls_key[tid]);
		s.cls_first[tid] = first ? 1u : 0u;
		s.grp_key[tid] = 0xFFFFFFFFu;
	}
	__syncthreads();
	if (tid < 256) {   // group digit = #distinct better keys among the classes that hold documents
		uint32_t g = 0;
		const uint32_t mine = s.cls_key[tid];
		for (int c = 0; c < 256; ++c) g += (s.cls_first[c] && s.cls_key[c] < mine) ? 1u : 0u;
		s.cls_group[tid] = g;
	}
	__syncthreads();
	if (tid < 256 && s.cls_count[tid]) s.grp_key[s.cls_group[tid]] = s.cls_key[tid];   // equal values from every class of the group
	// ids ordered the way Merged<desc> lists equal ranks: descending for desc — sort by (max_id - id) ascending
	if (desc) {
		for (uint32_t i = tid; i < n; i += kFuseThreads) keyA[i] = clsA[i] != kClsDropped ? s.max_id - keyA[i] : keyA[i];
	}
	__syncthreads();

	// ---- pass 2: stable LSD radix sort of the FT documents by id key; dropped documents travel along (they fall off in pass 5)
	const uint32_t passes = s.id_passes;
	for (uint32_t p = 0; p < passes; ++p) {
		const uint32_t shift = p * 8;
		radix_pass(s, keyA, clsA, keyB, clsB, n, [shift](uint32_t key, uint16_t cls) { return cls == kClsDropped ? 256u : ((key >> shift) & 255u); });
		uint32_t* tk = keyA;
		keyA = keyB;
		keyB = tk;
		uint16_t* tc = clsA;
		clsA = clsB;
		clsB = tc;
	}

	// ---- pass 3: the KNN list — ranks as the planner sees them (hnsw_index.cc:205-229), ids into the hash table; the FT documents probe it
	if (tid < int(nk)) {
		const uint32_t row = a.knn_row[tid];
		const int32_t id = a.rowid_of_row ? a.rowid_of_row[row] : int32_t(row);
		const float d = a.knn_dist[tid];
		s.k_id[tid] = id;
		s.k_rank[tid] = a.knn_negate ? -d : d;
		uint32_t slot = hash_slot(id);
		for (;;) {
			const int32_t prev = atomicCAS(&s.hash_id[slot], -1, id);
			if (prev == -1 || prev == id) break;   // an id that is twice in the KNN list (two vectors of one array row) shares the slot
			slot = (slot + 1) & (kHashSlots - 1);
		}
	}
	__syncthreads();
	if (nk) {
		for (uint32_t i = tid; i < n; i += kFuseThreads) {
			const uint16_t cls = clsA[i];
			if (cls == kClsDropped) continue;
			const int32_t id = int32_t(desc ? s.max_id - keyA[i] : keyA[i]);
			uint32_t slot = hash_slot(id);
			for (;;) {
				const int32_t h = s.hash_id[slot];
				if (h == -1) break;
				if (h == id) {
					s.hash_ft[slot] = uint32_t(cls);   // ids are unique on the FT side: one writer
					clsA[i] = uint16_t(cls | kClsInKnn);
					break;
				}
				slot = (slot + 1) & (kHashSlots - 1);
			}
		}
	}
	__syncthreads();

	// ---- pass 4: the head — fused ranks of the documents that came through the KNN list (selectiteratorcontainer.cc:1343-1423)
	if (tid < int(nk)) {
		const float r = s.k_rank[tid];
		// position of the run of equal ranks (the list is best first: L2 ascending, IP / cosine descending)
		uint32_t better = 0;
		for (uint32_t j = 0; j < uint32_t(tid); ++j) better += (a.metric_l2 ? s.k_rank[j] < r : s.k_rank[j] > r) ? 1u : 0u;
		const uint32_t knn_pos = better + 1;
		const int32_t id = s.k_id[tid];
		uint32_t slot = hash_slot(id);
		while (s.hash_id[slot] != id) slot = (slot + 1) & (kHashSlots - 1);
		const uint32_t ft_cls = s.hash_ft[slot];
		float f = 0.0f;
		bool keep = true;
		if (ft_cls != 0xFFFFFFFFu) {
			f = rrf ? float(1.0 / (a.params[0] + double(knn_pos)) + 1.0 / (a.params[0] + double(s.cls_pos[ft_cls])))
					: float(a.params[0] * double(r) + a.params[2] * double(float(ft_cls)) + a.params[4]);
		} else if (is_union) {
			f = rrf ? float(1.0 / (a.params[0] + double(knn_pos))) : float(a.params[0] * double(r) + a.params[2] * a.params[3] + a.params[4]);
		} else {
			keep = false;
		}
		s.k_fused[tid] = f;
		s.k_key[tid] = rank_key(f);
		s.k_keep[tid] = keep ? 1 : 0;
	}
	__syncthreads();
	if (tid < int(nk) && s.k_keep[tid]) {   // an exact (rank, id) repeat is one element of the set: the first occurrence stays
		const uint32_t key = s.k_key[tid];
		const int32_t id = s.k_id[tid];
		bool dup = false;
		for (uint32_t j = 0; j < uint32_t(tid) && !dup; ++j) dup = s.k_keep[j] && s.k_key[j] == key && s.k_id[j] == id;
		if (dup) s.k_keep[tid] = 2;   // decided on the original flags of the earlier entries: 2 still counts as "kept" for later readers
	}
	__syncthreads();
	if (tid == 0) s.n_head = 0;
	__syncthreads();
	if (tid < int(nk) && s.k_keep[tid] == 1) {
		const uint32_t key = s.k_key[tid];
		const int32_t id = s.k_id[tid];
		uint32_t before = 0;
		for (uint32_t j = 0; j < nk; ++j) {
			if (s.k_keep[j] != 1 || j == uint32_t(tid)) continue;
			before += (s.k_key[j] < key || (s.k_key[j] == key && id_before(s.k_id[j], id))) ? 1u : 0u;
		}
		s.h_key[before] = key;
		s.h_id[before] = id;
		s.h_rank[before] = s.k_fused[tid];
		atomicAdd(&s.n_head, 1u);
	}
	__syncthreads();
	const uint32_t nh = s.n_head;

	// ---- pass 5: the tail — FT documents that did not come through the KNN list, split by group (stable: ids stay ordered inside a group)
	uint32_t nt = 0;
	if (is_union) {
		radix_pass(s, keyA, clsA, keyB, clsB, n,
				   [&s](uint32_t, uint16_t cls) { return (cls == kClsDropped || (cls & kClsInKnn)) ? 256u : s.cls_group[cls & 255]; });
		if (tid <= 256) s.grp_start[tid] = s.tot[tid];   // exclusive prefix of the group sizes; [256] = #documents in the tail
		__syncthreads();
		nt = s.grp_start[256];
	}

	// ---- pass 6: final places.  A head entry goes behind the tail documents that precede it, a tail document behind the head entries that do.
	if (tid < int(nh)) {
		const uint32_t key = s.h_key[tid];
		const int32_t id = s.h_id[tid];
		uint32_t before = 0;
		if (nt) {
			// groups are ordered by key: the first group whose key is not better than mine
			uint32_t g = 0, ghi = 256;   // grp_key ascends over the used groups, the unused ones behind them carry 0xFFFFFFFF
			while (g < ghi) {
				const uint32_t mid = (g + ghi) >> 1;
				if (s.grp_key[mid] < key) {
					g = mid + 1;
				} else {
					ghi = mid;
				}
			}
			before = g < 256 ? s.grp_start[g] : nt;
			if (g < 256 && s.grp_key[g] == key) {   // same fused rank: ordered by id inside the group
				const uint32_t want = desc ? s.max_id - uint32_t(id) : uint32_t(id);   // tail keys ascend; ids above max_id precede everything when desc
				uint32_t lo = s.grp_start[g], hi = s.grp_start[g + 1];
				if (desc && uint32_t(id) > s.max_id) {
					hi = lo;
				}
				while (lo < hi) {
					const uint32_t mid = (lo + hi) >> 1;
					if (keyB[mid] < want) {
						lo = mid + 1;
					} else {
						hi = mid;
					}
				}
				before = lo;
			}
		}
		a.out_ids[tid + before] = id;
		a.out_ranks[tid + before] = s.h_rank[tid];
	}
	for (uint32_t i = tid; i < nt; i += kFuseThreads) {
		const uint32_t cls = clsB[i] & 255u;
		const uint32_t key = s.cls_key[cls];
		const int32_t id = int32_t(desc ? s.max_id - keyB[i] : keyB[i]);
		uint32_t lo = 0, hi = nh;   // #head entries before (key, id)
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			const bool head_first = s.h_key[mid] < key || (s.h_key[mid] == key && id_before(s.h_id[mid], id));
			if (head_first) {
				lo = mid + 1;
			} else {
				hi = mid;
			}
		}
		a.out_ids[i + lo] = id;
		a.out_ranks[i + lo] = s.cls_rank[cls];
	}
	if (tid == 0) {
		a.out_header[0] = nh + nt;
		// a distance tie straddling the k-th place is decided by labels on the host (gpu_bruteforce_map.cc: replayTies): tell the caller
		uint32_t flags = 0;
		const uint32_t avail = a.knn_count_ptr ? min(*a.knn_count_ptr, a.knn_n) : a.knn_n;
		if (a.k >= 1 && avail > a.k && a.k <= uint32_t(kMaxFuseKnn) && a.knn_dist[a.k] == a.knn_dist[a.k - 1]) flags |= 1u;
		a.out_header[1] = flags;
		a.out_header[2] = nh;
		a.out_header[3] = nt;
	}
}

hipError_t launch_hybrid_fuse(const HybridFuseArgs& a, hipStream_t st) {
	static std::atomic<uint64_t> raised{0};
	if (hipError_t e = raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&hybrid_fuse_kernel), sizeof(FuseShared)); e != hipSuccess) return e;
	hipLaunchKernelGGL(hybrid_fuse_kernel, dim3(1), dim3(kFuseThreads), sizeof(FuseShared), st, a);
	return hipGetLastError();
}

}  // namespace rxgpu
