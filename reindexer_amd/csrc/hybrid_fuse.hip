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
// The structure used: an FT rank is a uint8 (normalizedProc), so the FT side has at most 256 rank classes.  The RRF position of a class is
// 1 + #documents in better classes (equal ranks share the position of their run's first element); a document found by the FT side only
// gets a fused rank that depends on its class alone; classes whose fused ranks are equal floats form one group; inside a group
// Merged<desc> orders by id.  Hence: order of the FT documents among themselves = stable split by group of the id-ordered FT list — and
// NONE of that depends on the KNN half.  So the work is cut in two kernels:
//
//   hybrid_prepare_kernel   FT only, enqueued right behind the merge train: it runs WHILE the KNN scan (ten times longer than the merge)
//                           is still streaming the corpus.  postProcess -> (id, class), class histogram -> positions, fused ranks, groups;
//                           LSD radix sort by id (8-bit digits, wave-private tiles, stable), one more stable pass by group (fed back to
//                           front for desc).  Leaves the FT documents in their final mutual order + the class / group tables in HBM.
//                           One workgroup of 512 threads at <= 128 VGPRs: it fits beside the scan's two workgroups on a CU.
//   hybrid_join_kernel      what is left once both halves are there (the only part on the query's critical path): the (<= 1024) KNN
//                           entries into an LDS hash table, one coalesced pass of the prepared FT list through it (the documents that also
//                           came through the KNN list leave the tail), the KNN entries' fused ranks and order by counting, and both
//                           lists written to their final places by position arithmetic — no sort, no scattered pass.
// Bound: latency (chains of dependent passes over L2-resident data; every pass issues a thread's loads before it uses the first).  The
// scattered stores of a radix pass (20 000 documents x 2 arrays through ONE CU's memory pipeline, ~30 us a pass) are why the sort lives in
// the overlapped kernel.  Algorithmic bytes ~ (n_ft + k) * 40 B.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "rxgpu_internal.h"
#include "knn_kernels.hip.h"

namespace rxgpu {

namespace {

constexpr int kPrepThreads = 512;
constexpr int kPrepWaves = kPrepThreads / 64;
constexpr int kJoinThreads = 1024;
constexpr int kDigits = 257;             // 256 values + "falls off the end"
constexpr uint32_t kHashSlots = 4096;    // >= 4 x the largest KNN list
constexpr uint16_t kClsDropped = 0xFFFF;
constexpr int kBatch = 8;                // elements a thread loads before it uses the first

__device__ __forceinline__ uint32_t sortable_bits(float v) {
	const float r = v + 0.0f;   // -0 -> +0: the reference compares floats, both zeros are one rank
	const uint32_t u = __float_as_uint(r);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint32_t rank_key(float r, bool desc) { return desc ? ~sortable_bits(r) : sortable_bits(r); }

#define FUSE_STAMP(dbg, k)                                        \
	do {                                                          \
		if ((dbg) && threadIdx.x == 0) (dbg)[k] = wall_clock64(); \
	} while (0)

// ------------------------------------------------------------------------------------------------------------------ prepare (FT only)
struct PrepShared {
	uint32_t hist[kPrepWaves * kDigits];   // radix passes: per-wave digit counters / running offsets
	uint32_t tot[kDigits + 7];             // digit totals -> exclusive prefix
	uint32_t cls_count[256];
	uint32_t cls_pos[256];
	uint32_t cls_key[256];
	float cls_rank[256];
	uint32_t cls_group[256];
	uint32_t cls_sorted[256];
	uint32_t grp_key[256];
	uint32_t cls_present;
	uint32_t red_u[kPrepWaves];
	float red_f[kPrepWaves];
	uint32_t n_ft, max_id, id_passes;
};

__device__ __forceinline__ uint32_t prep_max_u32(uint32_t v, uint32_t* red) {
	for (int off = 32; off > 0; off >>= 1) v = max(v, uint32_t(__shfl_xor(int(v), off, 64)));
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	uint32_t r = red[0];
	for (int w = 1; w < kPrepWaves; ++w) r = max(r, red[w]);
	__syncthreads();
	return r;
}
__device__ __forceinline__ float prep_max_f32(float v, float* red) {
	for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	float r = red[0];
	for (int w = 1; w < kPrepWaves; ++w) r = fmaxf(r, red[w]);
	__syncthreads();
	return r;
}

// lanes of a wavefront that hold the same digit (one ballot per digit bit): one LDS operation per distinct digit instead of one per
// element — FT ranks are a few dozen distinct values, 64 same-address atomics would be served one after the other
template <int BITS>
__device__ __forceinline__ uint64_t peers_of(uint32_t d, bool valid) {
	uint64_t peers = __ballot(valid);
#pragma unroll
	for (int b = 0; b < BITS; ++b) {
		const uint64_t m = __ballot((d >> b) & 1u);
		peers &= ((d >> b) & 1u) ? m : ~m;
	}
	return peers;
}

// One stable LSD pass over n (key, class) pairs: the element at logical position i (read from src(i)) moves to the slot of its digit.
// Every wavefront owns a contiguous chunk of logical positions and walks it in tiles of 64 in order, so "earlier in the input" ==
// (earlier wave, earlier tile, lower lane) and the pass is stable.  The first kRegTiles tiles of a wave are loaded ONCE, back to back,
// and stay in registers across the counting and the scattering phase; longer chunks continue tile by tile.
constexpr int kRegTiles = 12;

template <typename DigitFn, typename SrcFn>
__device__ __forceinline__ void radix_pass(PrepShared& s, const uint32_t* __restrict__ kin, const uint16_t* __restrict__ cin,
													 uint32_t* __restrict__ kout, uint16_t* __restrict__ cout, uint32_t n, DigitFn digit, SrcFn src) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t chunk = ((n + kPrepWaves - 1) / kPrepWaves + 63) & ~63u;
	const uint32_t begin = min(n, wave * chunk), end = min(n, begin + chunk);
	const uint32_t ntiles = (end - begin + 63) / 64;
	for (int i = threadIdx.x; i < kPrepWaves * kDigits; i += kPrepThreads) s.hist[i] = 0;
	uint32_t rk[kRegTiles], rc2[kRegTiles / 2];   // classes two to a register
	auto rc = [&rc2](int t) { return uint16_t(rc2[t >> 1] >> ((t & 1) * 16)); };
#pragma unroll
	for (int t = 0; t < kRegTiles; t += 2) {   // unconditional loads (a clamped index), all in flight together
		const uint32_t i0 = begin + uint32_t(t) * 64 + lane, i1 = i0 + 64;
		const uint32_t s0 = i0 < end ? src(i0) : 0u, s1 = i1 < end ? src(i1) : 0u;
		rk[t] = n ? kin[s0] : 0u;
		rk[t + 1] = n ? kin[s1] : 0u;
		const uint32_t c0 = n ? uint32_t(cin[s0]) : 0u, c1 = n ? uint32_t(cin[s1]) : 0u;
		rc2[t >> 1] = c0 | (c1 << 16);
	}
	__syncthreads();
	uint32_t* hist_w = &s.hist[wave * kDigits];
	const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
	for (int t = 0; t < kRegTiles; ++t) {
		if (uint32_t(t) < ntiles) {
			const bool valid = begin + uint32_t(t) * 64 + lane < end;
			const uint32_t d = valid ? digit(rk[t], rc(t)) : 0;
			const uint64_t peers = peers_of<9>(d, valid);
			if (valid && (peers & lt) == 0) atomicAdd(&hist_w[d], uint32_t(__popcll(peers)));   // the first lane of every digit
		}
		__builtin_amdgcn_sched_barrier(0);
	}
	for (uint32_t t = kRegTiles; t < ntiles; ++t) {
		const uint32_t i = begin + t * 64 + lane;
		const bool valid = i < end;
		const uint32_t si = valid ? src(i) : 0u;
		const uint32_t d = valid ? digit(kin[si], cin[si]) : 0;
		const uint64_t peers = peers_of<9>(d, valid);
		if (valid && (peers & lt) == 0) atomicAdd(&hist_w[d], uint32_t(__popcll(peers)));
	}
	__syncthreads();
	if (threadIdx.x < kDigits) {   // per digit: counts of the waves -> exclusive prefix over the waves, total aside
		uint32_t sum = 0;
#pragma unroll
		for (int w = 0; w < kPrepWaves; ++w) {
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
#pragma unroll
		for (int w = 0; w < kPrepWaves; ++w) s.hist[w * kDigits + threadIdx.x] += base;
	}
	__syncthreads();
	auto place = [&](uint32_t key, uint32_t cls, bool valid) {
		const uint32_t d = valid ? digit(key, uint16_t(cls)) : 0;
		const uint64_t peers = peers_of<9>(d, valid);
		if (valid) {
			const uint32_t rank = __popcll(peers & lt), cnt = __popcll(peers);
			const uint32_t off = hist_w[d];
			if (rank + 1 == cnt) hist_w[d] = off + cnt;   // the last lane of the digit's peers moves the wave's running offset
			kout[off + rank] = key;
			cout[off + rank] = uint16_t(cls);
		}
		__builtin_amdgcn_wave_barrier();
	};
#pragma unroll
	for (int t = 0; t < kRegTiles; ++t) {
		if (uint32_t(t) < ntiles) place(rk[t], rc(t), begin + uint32_t(t) * 64 + lane < end);   // wave-uniform condition
		__builtin_amdgcn_sched_barrier(0);   // tile after tile: interleaving them only lengthens the live ranges
	}
	for (uint32_t t = kRegTiles; t < ntiles; ++t) {
		const uint32_t i = begin + t * 64 + lane;
		const bool valid = i < end;
		const uint32_t si = valid ? src(i) : 0u;
		place(kin[si], cin[si], valid);
	}
	__syncthreads();
}

}  // namespace

__global__ __launch_bounds__(kPrepThreads, 4) void hybrid_prepare_kernel(HybridFuseArgs a) {
	__shared__ PrepShared s;
	const int tid = threadIdx.x;
	const bool desc = a.desc != 0, rrf = a.kind == 0;
	uint32_t* keyA = a.scratch_key;
	uint32_t* keyB = a.scratch_key + a.ft_cap;
	uint16_t* clsA = a.scratch_cls;
	uint16_t* clsB = a.scratch_cls + a.ft_cap;
	HybridFuseState* st = a.state;

	if (tid == 0) {
		const uint32_t n = a.ft_count_ptr ? *a.ft_count_ptr : a.ft_n;
		s.n_ft = min(n, a.ft_cap);
	}
	for (int i = tid; i < 256; i += kPrepThreads) s.cls_count[i] = 0;
	__syncthreads();
	const uint32_t n = s.n_ft;

	// ---- postProcessResults (merger.h:111-140) when the FT side is the merge train's raw output: the largest proc
	float scale = 1.0f;
	if (a.ft_proc) {
		float mx = 0.0f;
		for (uint32_t base = tid; base < n; base += kPrepThreads * kBatch) {
			float v[kBatch];
#pragma unroll
			for (int j = 0; j < kBatch; ++j) {
				const uint32_t i = base + uint32_t(j) * kPrepThreads;
				v[j] = a.ft_proc[i < n ? i : 0];
				if (i >= n || (a.ft_terms && a.ft_terms[i] == 0xFFFFu)) v[j] = 0.0f;
			}
#pragma unroll
			for (int j = 0; j < kBatch; ++j) mx = fmaxf(mx, v[j]);
		}
		mx = prep_max_f32(mx, s.red_f);
		scale = mx > 255.0f ? float(255.0 / double(mx)) : 1.0f;
	}
	// ---- ids, classes, class histogram, largest id
	const int lane_id = tid & 63;
	const uint64_t lt_mask = lane_id ? (~0ull >> (64 - lane_id)) : 0ull;
	uint32_t max_id = 0;
	for (uint32_t base = tid; base < n; base += kPrepThreads * kBatch) {
		uint32_t doc[kBatch], r8[kBatch], id[kBatch];
		float proc[kBatch];
		bool gone[kBatch];   // removed with its partial synonym: as if it had not been merged
#pragma unroll
		for (int j = 0; j < kBatch; ++j) {
			const uint32_t i = base + uint32_t(j) * kPrepThreads, ci = i < n ? i : 0;
			doc[j] = a.ft_doc[ci];
			proc[j] = a.ft_proc ? a.ft_proc[ci] : 0.0f;
			r8[j] = a.ft_rank_u8 ? uint32_t(a.ft_rank_u8[ci]) : 0u;
			gone[j] = a.ft_terms != nullptr && a.ft_terms[ci] == 0xFFFFu;
		}
#pragma unroll
		for (int j = 0; j < kBatch; ++j) id[j] = a.row_of_doc ? uint32_t(a.row_of_doc[doc[j]]) : doc[j];   // the gathers, together
#pragma unroll
		for (int j = 0; j < kBatch; ++j) {
			const uint32_t i = base + uint32_t(j) * kPrepThreads;
			const uint32_t cls = gone[j] ? uint32_t(kClsDropped)
									 : a.ft_proc ? (proc[j] < a.min_rank ? uint32_t(kClsDropped) : uint32_t(uint8_t(proc[j] * scale))) : r8[j];
			const bool valid = i < n && cls != kClsDropped;
			if (i < n) {
				keyA[i] = id[j];
				clsA[i] = uint16_t(cls);
			}
			if (valid) max_id = max(max_id, id[j]);
			const uint64_t peers = peers_of<8>(cls & 255u, valid);
			if (valid && (peers & lt_mask) == 0) atomicAdd(&s.cls_count[cls & 255u], uint32_t(__popcll(peers)));
		}
	}
	max_id = prep_max_u32(max_id, s.red_u);
	if (tid == 0) {
		s.max_id = max_id;
		s.id_passes = max_id ? (32 - __clz(max_id) + 7) / 8 : 1;
		s.cls_present = 0;
	}
	{
		// 2 threads per class (kPrepThreads / 256), each over half of the other classes, folded with a shuffle
		constexpr int kParts = kPrepThreads / 256;
		const int c = tid / kParts, part = tid % kParts;
		// RRF position of a class (InitRRFPositions: equal ranks share the 1-based position of their run's first element)
		uint32_t better = 0;
		for (int c2 = part; c2 < 256; c2 += kParts) better += c2 > c ? s.cls_count[c2] : 0u;
#pragma unroll
		for (int off = 1; off < kParts; off <<= 1) better += __shfl_xor(better, off, 64);
		float f;   // fused rank of an FT-only document of class c (reranker.h: CalculateSingle / CalculateJustFt)
		if (rrf) {
			f = float(1.0 / (a.params[0] + double(1 + better)));
		} else {
			f = float(a.params[0] * a.params[1] + a.params[2] * double(float(c)) + a.params[4]);
		}
		const uint32_t key = rank_key(f, desc);
		if (part == 0) {
			s.cls_pos[c] = 1 + better;
			s.cls_rank[c] = f;
			s.cls_key[c] = key;
			s.grp_key[c] = 0xFFFFFFFFu;
			s.cls_group[c] = 0;
		}
		__syncthreads();
		// order of the classes that hold documents by (key, class): rank by counting
		const bool present = s.cls_count[c] != 0;
		uint32_t before = 0;
		for (int c2 = part; c2 < 256; c2 += kParts) {
			const uint32_t k2 = s.cls_key[c2];
			before += (s.cls_count[c2] != 0 && (k2 < key || (k2 == key && c2 < c))) ? 1u : 0u;
		}
#pragma unroll
		for (int off = 1; off < kParts; off <<= 1) before += __shfl_xor(before, off, 64);
		if (part == 0 && present) {
			s.cls_sorted[before] = uint32_t(c);
			atomicAdd(&s.cls_present, 1u);
		}
		__syncthreads();
		// classes with equal fused ranks interleave by id, so they share a group: group = #distinct keys in front (a scan over <= 256 flags)
		if (tid < 64) {
			const uint32_t np = s.cls_present;
			uint32_t flag[4], local = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t i = uint32_t(tid) * 4 + j;
				flag[j] = (i < np && i > 0 && s.cls_key[s.cls_sorted[i]] != s.cls_key[s.cls_sorted[i - 1]]) ? 1u : 0u;
				local += flag[j];
			}
			uint32_t incl = local;
			for (int off = 1; off < 64; off <<= 1) {
				const uint32_t o = __shfl_up(incl, off, 64);
				if (tid >= off) incl += o;
			}
			uint32_t run = incl - local;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t i = uint32_t(tid) * 4 + j;
				run += flag[j];
				if (i < np) {
					const uint32_t cc = s.cls_sorted[i];
					s.cls_group[cc] = run;
					s.grp_key[run] = s.cls_key[cc];   // equal values from every class of the group
				}
			}
		}
		__syncthreads();
	}

	// ---- stable LSD radix sort of the FT documents by id (ascending); dropped documents travel at the end and fall off in the last pass
	const uint32_t passes = s.id_passes;
	auto same = [](uint32_t i) { return i; };
	for (uint32_t p = 0; p < passes; ++p) {
		const uint32_t shift = p * 8;
		radix_pass(s, keyA, clsA, keyB, clsB, n, [shift](uint32_t key, uint16_t cls) { return cls == kClsDropped ? 256u : ((key >> shift) & 255u); }, same);
		uint32_t* tk = keyA;
		keyA = keyB;
		keyB = tk;
		uint16_t* tc = clsA;
		clsA = clsB;
		clsB = tc;
	}
	const uint32_t n_valid = s.tot[256];   // the last pass's prefix: documents in front of the dropped ones
	// ---- one more stable pass by group, fed back to front for desc: inside a group the ids then descend, as Merged<desc> lists equal ranks
	auto order = [desc, n_valid](uint32_t i) { return (desc && i < n_valid) ? n_valid - 1 - i : i; };
	const uint32_t* groups = s.cls_group;
	radix_pass(s, keyA, clsA, keyB, clsB, n, [groups](uint32_t, uint16_t cls) { return cls == kClsDropped ? 256u : groups[cls & 255]; }, order);

	// ---- what the join needs, in HBM
	if (tid < 256) {
		st->cls_pos[tid] = s.cls_pos[tid];
		st->cls_key[tid] = s.cls_key[tid];
		st->cls_rank[tid] = s.cls_rank[tid];
		st->cls_group[tid] = s.cls_group[tid];
		st->grp_key[tid] = s.grp_key[tid];
	}
	if (tid <= 256) st->grp_start[tid] = s.tot[tid];   // exclusive prefix of the group sizes; [256] = #documents that passed postProcess
	if (tid == 0) {
		st->n_valid = s.tot[256];
		st->max_id = s.max_id;
		st->result_in_second = (keyB == a.scratch_key + a.ft_cap) ? 1u : 0u;
	}
}

// ------------------------------------------------------------------------------------------------------------------------------- join
namespace {
struct JoinShared {
	uint32_t cls_pos[256];
	uint32_t cls_key[256];
	float cls_rank[256];
	uint32_t cls_group[256];
	uint32_t grp_key[256];
	uint32_t grp_start[kDigits + 1];
	uint32_t head_lt[257], head_le[257];   // #head entries with a key better than / not worse than the key of group g
	int32_t hash_id[kHashSlots];           // KNN ids (open addressing), -1 = empty
	uint32_t hash_ft[kHashSlots];          // class of the FT document with that id, 0xFFFFFFFF = not in the FT list
	int32_t k_id[kMaxFuseKnn];
	float k_rank[kMaxFuseKnn];             // KNN rank as the caller sees it (L2: distance, IP / cosine: -distance)
	uint32_t k_key[kMaxFuseKnn];
	float k_fused[kMaxFuseKnn];
	uint8_t k_keep[kMaxFuseKnn];
	uint32_t h_key[kMaxFuseKnn];           // head in final order
	int32_t h_id[kMaxFuseKnn];
	float h_rank[kMaxFuseKnn];
	uint32_t rm[kMaxFuseKnn];              // positions (in the prepared order) of the FT documents that are in the KNN list ...
	uint32_t rm_sorted[kMaxFuseKnn];       // ... ascending
	uint32_t n_knn, n_head, n_rm;
};
__device__ __forceinline__ uint32_t hash_slot(int32_t id) { return (uint32_t(id) * 2654435761u) >> 20; }   // 12 bits
}  // namespace

__global__ __launch_bounds__(kJoinThreads) void hybrid_join_kernel(HybridFuseArgs a) {
	extern __shared__ __align__(16) unsigned char join_lds[];
	JoinShared& s = *reinterpret_cast<JoinShared*>(join_lds);
	const int tid = threadIdx.x;
	const bool desc = a.desc != 0, rrf = a.kind == 0, is_union = a.is_union != 0;
	auto id_before = [desc](int32_t l, int32_t r) { return desc ? l > r : l < r; };
	const HybridFuseState* st = a.state;
	FUSE_STAMP(a.dbg, 0);

	if (tid == 0) {
		const uint32_t nk = a.knn_count_ptr ? min(*a.knn_count_ptr, a.knn_n) : a.knn_n;
		s.n_knn = min(min(nk, a.k), uint32_t(kMaxFuseKnn));
		s.n_head = 0;
		s.n_rm = 0;
	}
	for (int i = tid; i < int(kHashSlots); i += kJoinThreads) {
		s.hash_id[i] = -1;
		s.hash_ft[i] = 0xFFFFFFFFu;
	}
	if (tid < 256) {
		s.cls_pos[tid] = st->cls_pos[tid];
		s.cls_key[tid] = st->cls_key[tid];
		s.cls_rank[tid] = st->cls_rank[tid];
		s.cls_group[tid] = st->cls_group[tid];
		s.grp_key[tid] = st->grp_key[tid];
	}
	if (tid <= 256) s.grp_start[tid] = st->grp_start[tid];
	const uint32_t n_valid = min(st->n_valid, a.ft_cap);
	const bool second = st->result_in_second != 0;
	const uint32_t* __restrict__ keyS = a.scratch_key + (second ? a.ft_cap : 0);
	const uint16_t* __restrict__ clsS = a.scratch_cls + (second ? a.ft_cap : 0);
	__syncthreads();
	const uint32_t nk = s.n_knn;

	// ---- the KNN list — ranks as the planner sees them (hnsw_index.cc:205-229), ids into the hash table
	if (tid < int(nk)) {
		const uint32_t row = a.knn_row[tid];
		const float d = a.knn_dist[tid];
		const int32_t id = a.rowid_of_row ? a.rowid_of_row[row] : int32_t(row);
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
	FUSE_STAMP(a.dbg, 1);
	// ---- one coalesced pass of the prepared FT list through the table: a document that is also in the KNN list leaves its class there
	// and its position in the removal list
	if (nk) {
		for (uint32_t base = tid; base < n_valid; base += kJoinThreads * kBatch) {
			uint32_t tk[kBatch], tc[kBatch];
#pragma unroll
			for (int j = 0; j < kBatch; ++j) {
				const uint32_t i = base + uint32_t(j) * kJoinThreads, ci = i < n_valid ? i : 0;
				tk[j] = keyS[ci];
				tc[j] = clsS[ci];
			}
#pragma unroll
			for (int j = 0; j < kBatch; ++j) {
				const uint32_t i = base + uint32_t(j) * kJoinThreads;
				if (i >= n_valid) continue;
				const int32_t id = int32_t(tk[j]);
				uint32_t slot = hash_slot(id);
				for (;;) {
					const int32_t h = s.hash_id[slot];
					if (h == -1) break;
					if (h == id) {
						s.hash_ft[slot] = tc[j] & 255u;   // ids are unique on the FT side: one writer
						s.rm[atomicAdd(&s.n_rm, 1u)] = i;
						break;
					}
					slot = (slot + 1) & (kHashSlots - 1);
				}
			}
		}
	}
	__syncthreads();
	FUSE_STAMP(a.dbg, 2);

	// ---- the head — fused ranks of the documents that came through the KNN list (selectiteratorcontainer.cc:1343-1423).
	// P threads per entry (as many as the workgroup affords), each over a P-th of the other entries, folded with shuffles.
	uint32_t P = 64;
	while (P > 1 && P * nk > uint32_t(kJoinThreads)) P >>= 1;
	const uint32_t per_round = kJoinThreads / P, part = uint32_t(tid) % P;
	const uint32_t rounds = nk ? (nk + per_round - 1) / per_round : 0;
	auto fold = [P](uint32_t v) {
		for (uint32_t off = 1; off < P; off <<= 1) v += __shfl_xor(v, int(off), 64);
		return v;
	};
	for (uint32_t r0 = 0; r0 < rounds; ++r0) {
		const uint32_t e = r0 * per_round + uint32_t(tid) / P;
		const bool live = e < nk;
		const float r = live ? s.k_rank[e] : 0.0f;
		// position of the run of equal ranks (the list is best first: L2 ascending, IP / cosine descending)
		uint32_t better = 0;
		for (uint32_t j = part; j < nk; j += P) better += (live && j < e && (a.metric_l2 ? s.k_rank[j] < r : s.k_rank[j] > r)) ? 1u : 0u;
		better = fold(better);
		if (live && part == 0) {
			const uint32_t knn_pos = better + 1;
			const int32_t id = s.k_id[e];
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
			s.k_fused[e] = f;
			s.k_key[e] = rank_key(f, desc);
			s.k_keep[e] = keep ? 1 : 0;
		}
	}
	// the removal list in ascending order (rank by counting, Q threads per entry)
	const uint32_t nrm = min(s.n_rm, uint32_t(kMaxFuseKnn));
	{
		uint32_t Q = 64;
		while (Q > 1 && Q * nrm > uint32_t(kJoinThreads)) Q >>= 1;
		const uint32_t per = kJoinThreads / Q, qpart = uint32_t(tid) % Q;
		const uint32_t qrounds = nrm ? (nrm + per - 1) / per : 0;
		for (uint32_t r0 = 0; r0 < qrounds; ++r0) {
			const uint32_t e = r0 * per + uint32_t(tid) / Q;
			const bool live = e < nrm;
			const uint32_t mine = live ? s.rm[e] : 0;
			uint32_t before = 0;
			for (uint32_t j = qpart; j < nrm; j += Q) before += (live && s.rm[j] < mine) ? 1u : 0u;   // positions are distinct
			for (uint32_t off = 1; off < Q; off <<= 1) before += __shfl_xor(before, int(off), 64);
			if (live && qpart == 0) s.rm_sorted[before] = mine;
		}
	}
	__syncthreads();
	for (uint32_t r0 = 0; r0 < rounds; ++r0) {   // an exact (rank, id) repeat is one element of the set: the first occurrence stays
		const uint32_t e = r0 * per_round + uint32_t(tid) / P;
		const bool live = e < nk && s.k_keep[e];
		const uint32_t key = live ? s.k_key[e] : 0;
		const int32_t id = live ? s.k_id[e] : 0;
		uint32_t dup = 0;
		for (uint32_t j = part; j < nk; j += P) dup += (live && j < e && s.k_keep[j] && s.k_key[j] == key && s.k_id[j] == id) ? 1u : 0u;
		dup = fold(dup);
		if (live && part == 0 && dup) s.k_keep[e] = 2;   // 2 is still "kept" for the entries behind it that compare against the original flags
	}
	__syncthreads();
	for (uint32_t r0 = 0; r0 < rounds; ++r0) {
		const uint32_t e = r0 * per_round + uint32_t(tid) / P;
		const bool live = e < nk && s.k_keep[e] == 1;
		const uint32_t key = live ? s.k_key[e] : 0;
		const int32_t id = live ? s.k_id[e] : 0;
		uint32_t before = 0;
		for (uint32_t j = part; j < nk; j += P) {
			const bool counts = live && j != e && s.k_keep[j] == 1 && (s.k_key[j] < key || (s.k_key[j] == key && id_before(s.k_id[j], id)));
			before += counts ? 1u : 0u;
		}
		before = fold(before);
		if (live && part == 0) {
			s.h_key[before] = key;
			s.h_id[before] = id;
			s.h_rank[before] = s.k_fused[e];
			atomicAdd(&s.n_head, 1u);
		}
	}
	__syncthreads();
	const uint32_t nh = s.n_head;
	const uint32_t nt = is_union ? n_valid - nrm : 0;   // the tail: FT documents that did not come through the KNN list
	FUSE_STAMP(a.dbg, 3);

	// #removed documents in front of prepared position p
	auto removed_before = [&s, nrm](uint32_t p) {
		uint32_t lo = 0, hi = nrm;
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if (s.rm_sorted[mid] < p) {
				lo = mid + 1;
			} else {
				hi = mid;
			}
		}
		return lo;
	};
	if (nt && tid < 257) {   // per group: the head entries in front of all its documents / in front of or level with them
		const uint32_t key = tid < 256 ? s.grp_key[tid] : 0xFFFFFFFFu;
		uint32_t lo = 0, hi = nh;
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if (s.h_key[mid] < key) {
				lo = mid + 1;
			} else {
				hi = mid;
			}
		}
		s.head_lt[tid] = lo;
		hi = nh;
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if (s.h_key[mid] <= key) {
				lo = mid + 1;
			} else {
				hi = mid;
			}
		}
		s.head_le[tid] = lo;
	}
	// ---- final places.  A head entry goes behind the tail documents that precede it, a tail document behind the head entries that do.
	if (tid < int(nh)) {
		const uint32_t key = s.h_key[tid];
		const int32_t id = s.h_id[tid];
		uint32_t before = 0;
		if (nt) {
			uint32_t g = 0, ghi = 256;   // grp_key ascends over the used groups, the unused ones behind them carry 0xFFFFFFFF
			while (g < ghi) {
				const uint32_t mid = (g + ghi) >> 1;
				if (s.grp_key[mid] < key) {
					g = mid + 1;
				} else {
					ghi = mid;
				}
			}
			uint32_t p = g < 256 ? s.grp_start[g] : n_valid;   // prepared position of the first document that does not precede the entry
			if (g < 256 && s.grp_key[g] == key) {   // same fused rank: its place among the group's ids
				uint32_t lo = s.grp_start[g], hi = s.grp_start[g + 1];
				while (lo < hi) {
					const uint32_t mid = (lo + hi) >> 1;
					if (id_before(int32_t(keyS[mid]), id)) {
						lo = mid + 1;
					} else {
						hi = mid;
					}
				}
				p = lo;
			}
			before = p - removed_before(p);
		}
		a.out_ids[tid + before] = id;
		a.out_ranks[tid + before] = s.h_rank[tid];
	}
	__syncthreads();
	if (nt) {
		for (uint32_t base = tid; base < n_valid; base += kJoinThreads * kBatch) {
			uint32_t tk[kBatch], tc[kBatch];
#pragma unroll
			for (int j = 0; j < kBatch; ++j) {
				const uint32_t i = base + uint32_t(j) * kJoinThreads, ci = i < n_valid ? i : 0;
				tk[j] = keyS[ci];
				tc[j] = clsS[ci];
			}
#pragma unroll
			for (int j = 0; j < kBatch; ++j) {
				const uint32_t i = base + uint32_t(j) * kJoinThreads;
				if (i >= n_valid) continue;
				const uint32_t rb = removed_before(i);
				if (rb < nrm && s.rm_sorted[rb] == i) continue;   // came through the KNN list: it is in the head
				const uint32_t cls = tc[j] & 255u, g = s.cls_group[cls];
				const int32_t id = int32_t(tk[j]);
				uint32_t lo = s.head_lt[g], hi = s.head_le[g];   // head entries of the same fused rank (rare): the document's place among their ids
				while (lo < hi) {
					const uint32_t mid = (lo + hi) >> 1;
					if (id_before(s.h_id[mid], id)) {
						lo = mid + 1;
					} else {
						hi = mid;
					}
				}
				const uint32_t dst = i - rb + lo;
				a.out_ids[dst] = id;
				a.out_ranks[dst] = s.cls_rank[cls];
			}
		}
	}
	__syncthreads();
	FUSE_STAMP(a.dbg, 4);
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

hipError_t launch_hybrid_prepare(const HybridFuseArgs& a, hipStream_t st) {
	hipLaunchKernelGGL(hybrid_prepare_kernel, dim3(1), dim3(kPrepThreads), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_hybrid_join(const HybridFuseArgs& a, hipStream_t st) {
	static std::atomic<uint64_t> raised{0};
	if (hipError_t e = raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&hybrid_join_kernel), sizeof(JoinShared)); e != hipSuccess) return e;
	hipLaunchKernelGGL(hybrid_join_kernel, dim3(1), dim3(kJoinThreads), sizeof(JoinShared), st, a);
	return hipGetLastError();
}

}  // namespace rxgpu
