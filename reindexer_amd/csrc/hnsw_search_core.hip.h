// Device code shared by the HNSW search kernels (hnsw_search.hip: the launches; hnsw_server.hip: the resident form): the reference's two
// heaps, the sorted list, distance batches, visited sets and ONE search by one wavefront / one team (hnsw_search_one).  See hnsw_search.hip
// for what is replaced and why the results are equal.
#pragma once
#include <type_traits>

#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {

// --- PriorityQueue<pair<float,tableint>, vector, CompareByFirst> restated on an array of (dist bits, id) pairs (executed by ONE lane) ---
// One 8-byte entry per element: a sift level is ONE dependent LDS round trip (both children, each one ds_read_b64, issued together)
// instead of three (two distances, then the winner's id) — the heap updates of lane 0 are the longest stretch of a hop.
__device__ __forceinline__ float hp_dist(uint2 e) { return __uint_as_float(e.x); }
__device__ __forceinline__ void hp_sift_up(uint2* h, int child) {
	const uint2 v = h[child];
	while (child > 0) {
		const int parent = (child - 1) / 2;
		const uint2 pe = h[parent];
		if (!(hp_dist(pe) < hp_dist(v))) break;
		h[child] = pe;
		child = parent;
	}
	h[child] = v;
}
__device__ __forceinline__ void hp_sift_down(uint2* h, int parent, int size) {
	const uint2 v = h[parent];
	for (;;) {
		const int left = parent * 2 + 1;
		if (left >= size) break;
		const int right = left + 1;
		const uint2 le = h[left];
		const uint2 re = h[right < size ? right : left];
		const bool take_right = right < size && hp_dist(le) < hp_dist(re);
		const uint2 be = take_right ? re : le;
		if (!(hp_dist(v) < hp_dist(be))) break;
		h[parent] = be;
		parent = take_right ? right : left;
	}
	h[parent] = v;
}
__device__ __forceinline__ void hp_emplace(uint2* h, int& n, float vd, uint32_t vi) {
	h[n] = make_uint2(__float_as_uint(vd), vi);
	++n;
	if (n >= 2) hp_sift_up(h, n - 1);
}
__device__ __forceinline__ void hp_pop(uint2* h, int& n) {
	if (n >= 2) {
		const uint2 t = h[0];
		h[0] = h[n - 1];
		h[n - 1] = t;
		if (n > 2) hp_sift_down(h, 0, n - 1);
	}
	--n;
}
__device__ __forceinline__ void hp_replace_top(uint2* h, int n, float vd, uint32_t vi) {
	h[0] = make_uint2(__float_as_uint(vd), vi);
	hp_sift_down(h, 0, n);
}

// --- The fast path of a search over a graph without deleted nodes: BOTH queues as one sorted list spread over the lanes ---
// In a bare-bone search (hnswalg.h:873-876, 932-960) every candidate is pushed into top_candidates together with its push into
// candidate_set, so the live part of candidate_set is "the members of top_candidates that were not expanded yet": what top_candidates
// evicted lies at or above lowerBound for good, and popping it ends the search like an empty candidate_set does (at: see below).  One list of <= ef
// entries sorted by distance plus one "expanded" bit per entry therefore carries the whole Layer0SearchState: entry 64 s + l lives in
// slot s of lane l, an insertion is one ballot + one lane shift (wave_shr DPP) instead of two binary-heap sifts by one lane, the pop is a
// scalar find-first-zero.
// Equal distances.  With CompareByFirst heaps the reference's choice among EQUAL keys is whatever libstdc++'s sift loops leave on top,
// which a sorted list cannot know.  Where that choice cannot matter the list goes on; where it can, the search is flagged and starts
// over on the heaps (the code below the list's branch in hnsw_search_kernel; kHnswTie and a launch of its own if the launcher gave the
// workgroup no heap area):
//   * the popped candidate's key d equals the next unexpanded one's (which of the two the reference expands first is its heap's
//     secret) AND lowerBound has come down to d by the time every candidate with a key <= d is expanded.  While lowerBound stays above
//     d the order is immaterial: each node of key <= d that gets evaluated is admitted (key < lowerBound) and expanded before anything
//     farther, so both orders expand the same closure — "evaluated nodes of key <= d, and what their lists reach" —, top_candidates
//     is the ef smallest keys of the same evaluated set, and lowerBound in the other order is never below its value at the end of this
//     one (fewer keys seen, larger ef-th smallest).  The check is made when the first key > d is popped, or when the list runs dry;
//   * the popped key, or lowerBound when the list runs dry, equals the key of the last node that went OUTSIDE the list with a key equal
//     to lowerBound — evicted next to an equal maximum, or refused admission at dist == lowerBound: an evicted entry is still in the
//     reference's candidate_set, alive (dist > lowerBound is false) and due for expansion, and in another order of equal pops a refused
//     node is one that got in.  (Keys outside the list never lie below lowerBound, and lowerBound only falls: the last such key is the
//     only one that can still equal it.)
//   * entries k - 1 and k of the final list are equal (SearchKnn's trim to k pops one of them);
//   * a key is not finite.
// Everything else depends on keys alone: top_candidates is the multiset of the ef smallest keys seen, an eviction among equal maxima
// changes neither lowerBound nor candidate_set, and an equal key met in the middle of the list has no consequence until one of the
// events above.  (Flagging every equal key met on insertion re-ran 31 % of the queries of a 1M x 768 corpus: with 128 keys in a narrow
// band of float32 values a search of ~400 insertions meets one more often than not.)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) {   // lane l <- lane l - 1 (lane 0 keeps its value)
	return uint32_t(__builtin_amdgcn_update_dpp(int(x), int(x), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_value(float v, int l) { return __uint_as_float(uint32_t(__builtin_amdgcn_readlane(int(__float_as_uint(v)), l))); }
// The per-slot state of a list as VECTOR values, not arrays: an array member keeps the whole struct in scratch memory as soon as a loop
// over it is not unrolled before LLVM's SROA pass runs (seen for S = 3 and 4: 100 bytes of scratch per lane, every access a memory op).
template <int S>
struct HnswVec {
	typedef float f __attribute__((ext_vector_type(S)));
	typedef uint32_t u __attribute__((ext_vector_type(S)));
	typedef uint64_t m __attribute__((ext_vector_type(S)));
};
// entries >= pos move up by one (the last one falls off the registers), (nd, nid) lands at pos
template <int S>
__device__ __forceinline__ void sorted_shift_in(typename HnswVec<S>::f& d, typename HnswVec<S>::u& id, int pos, float nd, uint32_t nid, bool dpp, int lane) {
#pragma unroll
	for (int s = S - 1; s >= 0; --s) {
		if (pos >= 64 * (s + 1)) continue;   // uniform: the slot lies below the insertion point
		const int g = 64 * s + lane;
		float up_d;
		uint32_t up_i;
		if (dpp) {
			up_d = __uint_as_float(wave_shr1(__float_as_uint(d[s])));
			up_i = wave_shr1(id[s]);
		} else {   // RXGPU_HNSW_SORTED=2: the same shift through the LDS crossbar (ds_bpermute)
			up_d = __shfl_up(d[s], 1, 64);
			up_i = __shfl_up(id[s], 1, 64);
		}
		if (s > 0) {
			const float carry_d = lane_value(d[s > 0 ? s - 1 : 0], 63);
			const uint32_t carry_i = uint32_t(__builtin_amdgcn_readlane(int(id[s > 0 ? s - 1 : 0]), 63));
			if (lane == 0) {
				up_d = carry_d;
				up_i = carry_i;
			}
		}
		d[s] = g < pos ? d[s] : (g == pos ? nd : up_d);
		id[s] = g < pos ? id[s] : (g == pos ? nid : up_i);
	}
}
// the same for a wave-uniform bit per entry
template <int S>
__device__ __forceinline__ void mask_shift_in(typename HnswVec<S>::m& m, int pos, bool bit) {
#pragma unroll
	for (int s = S - 1; s >= 0; --s) {
		if (pos >= 64 * (s + 1)) continue;
		if (pos < 64 * s) {
			m[s] = (m[s] << 1) | (m[s > 0 ? s - 1 : 0] >> 63);
		} else {
			const uint64_t below = (1ull << (pos - 64 * s)) - 1ull;
			m[s] = (m[s] & below) | ((m[s] & ~below) << 1) | (uint64_t(bit) << (pos - 64 * s));
		}
	}
}
template <int S>
struct HnswSortedList {
	typename HnswVec<S>::f d;
	typename HnswVec<S>::u id;
	typename HnswVec<S>::m done;   // wave-uniform: bit l of word s = entry 64 s + l is expanded (or empty)
	int n;
	float lower;        // lowerBound: the largest key while the list is filling, entry ef - 1 afterwards
	float outside;      // key of the last node evicted, or refused at dist == lowerBound (NaN before the first: equal to nothing)
	float pend;         // largest key at which a pop met an equal unexpanded key and the verdict is still open
	bool pending;
	bool tie;
	bool dpp;

	__device__ __forceinline__ void init(bool use_dpp) {
		dpp = use_dpp;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			d[s] = __builtin_inff();
			id[s] = 0u;
			done[s] = ~0ull;
		}
		n = 0;
		lower = 3.402823466e+38f;
		outside = __builtin_nanf("");
		pend = 0.f;
		pending = false;
		tie = false;
	}
	// every candidate of key <= pend is expanded now: the order among the equal ones was immaterial iff lowerBound is still above pend
	// (a list that is not full admits everything, whatever lowerBound says)
	__device__ __forceinline__ void settle(int ef) {
		if (pending && n == ef && !(lower > pend)) tie = true;
		pending = false;
	}
	// key of entry e (wave-uniform e)
	__device__ __forceinline__ float key_at(int e) const {
		float v = d[0];
#pragma unroll
		for (int s = 1; s < S; ++s) v = (e >> 6) == s ? d[s] : v;
		return lane_value(v, e & 63);
	}
	// (nd, nid) wave-uniform; the caller has checked n < ef || lower > nd
	__device__ __forceinline__ void insert(float nd, uint32_t nid, int ef, int lane) {
		int pos = 0;
#pragma unroll
		for (int s = 0; s < S; ++s) pos += __popcll(__ballot(d[s] < nd));
		tie = tie || !(nd < __builtin_inff());
		if (n == ef) outside = lower;   // the list is full: its last entry leaves
		sorted_shift_in<S>(d, id, pos, nd, nid, dpp, lane);
		mask_shift_in<S>(done, pos, false);   // bit pos: 0 = not expanded
		if (n < ef) {
			++n;
		} else {
#pragma unroll
			for (int s = 0; s < S; ++s) {   // the evicted maximum sits at index ef now (if the registers reach that far): an empty entry again
				const bool here = (ef >> 6) == s;
				d[s] = (here && lane == (ef & 63)) ? __builtin_inff() : d[s];
				done[s] |= here ? 1ull << (ef & 63) : 0ull;
			}
		}
		lower = key_at(n - 1);
	}
	// the interface the kernel shares with HnswSortedListDel: members of top_candidates held, insertion with a delete mark (none here)
	__device__ __forceinline__ int held() const { return n; }
	__device__ __forceinline__ void insert(float nd, uint32_t nid, bool, int ef, int lane) { insert(nd, nid, ef, lane); }
	// first entry that was not expanded yet, -1 if none (selects over static slot numbers, no early exit: the arrays must stay in registers)
	__device__ __forceinline__ int first_open() const {
		int e = -1;
#pragma unroll
		for (int s = S - 1; s >= 0; --s) {
			const uint64_t o = ~done[s];
			e = o ? 64 * s + __builtin_ctzll(o) : e;
		}
		return e;
	}
	// the first unexpanded entry behind entry `after`, -1 if none
	__device__ __forceinline__ int next_open(int after) const {
		int e = -1;
#pragma unroll
		for (int s = S - 1; s >= 0; --s) {
			uint64_t o = ~done[s];
			if ((after >> 6) == s) o = (after & 63) == 63 ? 0ull : (o & (~0ull << ((after & 63) + 1)));
			if ((after >> 6) > s) o = 0ull;
			e = o ? 64 * s + __builtin_ctzll(o) : e;
		}
		return e;
	}
	// candidate_set.top() + pop(): the nearest entry that was not expanded yet
	__device__ __forceinline__ bool pop(uint32_t& node, float& dist, int ef) {
		const int e = first_open();
		if (e < 0) {
			settle(ef);
			return false;
		}
		uint32_t iv = id[0];
#pragma unroll
		for (int s = 1; s < S; ++s) iv = (e >> 6) == s ? id[s] : iv;
		dist = key_at(e);
		node = uint32_t(__builtin_amdgcn_readlane(int(iv), e & 63));
		if (pending && dist > pend) settle(ef);
#pragma unroll
		for (int s = 0; s < S; ++s) done[s] |= (e >> 6) == s ? 1ull << (e & 63) : 0ull;
		const int next = first_open();
		tie = tie || dist == outside;
		if (next >= 0 && key_at(next) == dist) {
			pend = pending ? fmaxf(pend, dist) : dist;
			pending = true;
		}
		return true;
	}
};

// The same list for a graph WITH deleted nodes (hnswalg.h:882-893, 943-957): a deleted node is a candidate like any other but never a
// member of top_candidates, lowerBound follows the LIVE entries only, the search stops on a far candidate only once ef live entries are
// held, and a deleted entry point enters candidate_set at FLT_MAX (initLayer0SearchState :853-856).  The list holds live and deleted
// entries in one order; `del` marks the deleted ones, `live` counts the others.  Once ef live entries are held everything behind the
// last of them is dead — deleted entries above lowerBound, the live entry a new one evicts — and leaves the list; `outside` takes the
// smallest key that leaves.  The equal-key rules are the bare list's, with "full" meaning live == ef; a list that runs out of registers
// (many deleted nodes in reach) flags the search like an equal key does.  tests/test_hnsw_sorted_model.py holds the Python restatement
// of both lists and runs it against the two-heap oracle.
template <int S>
struct HnswSortedListDel {
	static constexpr int kCap = 64 * S;
	typename HnswVec<S>::f d;
	typename HnswVec<S>::u id;
	typename HnswVec<S>::m done;   // wave-uniform: expanded (or empty)
	typename HnswVec<S>::m del;    // wave-uniform: the entry is a deleted node (0 for empty slots)
	int n, live;
	float lower, outside, pend;
	bool pending, tie, dpp;

	__device__ __forceinline__ void init(bool use_dpp) {
		dpp = use_dpp;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			d[s] = __builtin_inff();
			id[s] = 0u;
			done[s] = ~0ull;
			del[s] = 0ull;
		}
		n = live = 0;
		lower = 3.402823466e+38f;
		outside = __builtin_nanf("");
		pend = 0.f;
		pending = false;
		tie = false;
	}
	__device__ __forceinline__ float key_at(int e) const {
		float v = d[0];
#pragma unroll
		for (int s = 1; s < S; ++s) v = (e >> 6) == s ? d[s] : v;
		return lane_value(v, e & 63);
	}
	// bits of word s that belong to entries < count
	static __device__ __forceinline__ uint64_t below_count(int s, int count) {
		const int c = count - 64 * s;
		return c <= 0 ? 0ull : (c >= 64 ? ~0ull : ((1ull << c) - 1ull));
	}
	__device__ __forceinline__ int last_live() const {
		int e = -1;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const uint64_t m = ~del[s] & below_count(s, n);
			e = m ? 64 * s + 63 - __builtin_clzll(m) : e;
		}
		return e;
	}
	__device__ __forceinline__ int held() const { return live; }
	__device__ __forceinline__ void settle(int ef) {
		if (pending && live == ef && !(lower > pend)) tie = true;
		pending = false;
	}
	// (nd, nid, isdel) wave-uniform; the caller has checked live < ef || lower > nd
	__device__ __forceinline__ void insert(float nd, uint32_t nid, bool isdel, int ef, int lane) {
		tie = tie || !(nd < __builtin_inff());
		const bool full = live == ef;
		if (n == kCap && !(full && !isdel)) {   // no register left for one more entry: the heaps take the search over
			tie = true;
			return;
		}
		int pos = 0;
#pragma unroll
		for (int s = 0; s < S; ++s) pos += __popcll(__ballot(d[s] < nd));
		const float old_lower = lower;
		sorted_shift_in<S>(d, id, pos, nd, nid, dpp, lane);
		mask_shift_in<S>(done, pos, false);
		mask_shift_in<S>(del, pos, isdel);
		const bool fell_off = n == kCap;   // only with a full top and a live newcomer: the entry that fell off was the largest live one
		if (!fell_off) ++n;
		if (!isdel) {
			if (!full) {
				++live;
			} else if (fell_off) {
				outside = old_lower;
			} else {   // the largest live entry leaves top_candidates: from here on it is one of the dead behind the last live entry
				const int gone = last_live();
				outside = key_at(gone);
#pragma unroll
				for (int s = 0; s < S; ++s) del[s] |= (gone >> 6) == s ? 1ull << (gone & 63) : 0ull;
			}
			if (live == ef) {   // everything behind the last live entry is dead now
				const int keep = last_live() + 1;
				if (keep < n) {
					outside = key_at(keep);   // the smallest key that leaves
#pragma unroll
					for (int s = 0; s < S; ++s) {
						const uint64_t stay = below_count(s, keep);
						d[s] = (64 * s + lane) >= keep ? __builtin_inff() : d[s];
						done[s] |= ~stay;
						del[s] &= stay;
					}
					n = keep;
				}
			}
		}
		if (live > 0) lower = key_at(last_live());
	}
	__device__ __forceinline__ int first_open() const {
		int e = -1;
#pragma unroll
		for (int s = S - 1; s >= 0; --s) {
			const uint64_t o = ~done[s];
			e = o ? 64 * s + __builtin_ctzll(o) : e;
		}
		return e;
	}
	__device__ __forceinline__ bool pop(uint32_t& node, float& dist, int ef) {
		const int e = first_open();
		if (e < 0) {
			settle(ef);
			return false;
		}
		uint32_t iv = id[0];
#pragma unroll
		for (int s = 1; s < S; ++s) iv = (e >> 6) == s ? id[s] : iv;
		dist = key_at(e);
		node = uint32_t(__builtin_amdgcn_readlane(int(iv), e & 63));
		if (pending && dist > pend) settle(ef);
#pragma unroll
		for (int s = 0; s < S; ++s) done[s] |= (e >> 6) == s ? 1ull << (e & 63) : 0ull;
		const int next = first_open();
		tie = tie || dist == outside;
		if (next >= 0 && key_at(next) == dist) {
			pend = pending ? fmaxf(pend, dist) : dist;
			pending = true;
		}
		return true;
	}
	// index of the r-th live entry (r < live)
	__device__ __forceinline__ int live_at(int r) const {
		int e = -1, seen = 0;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			uint64_t m = ~del[s] & below_count(s, n);
			const int c = __popcll(m);
			if (e < 0 && r < seen + c) {
				for (int i = seen; i < r; ++i) m &= m - 1;   // drop the r - seen lowest set bits (wave-uniform loop)
				e = 64 * s + __builtin_ctzll(m);
			}
			seen += c;
		}
		return e;
	}
};

// Distances of `cnt` rows (ids in LDS) to the query, 4 rows per step; every lane participates in every step.
template <int kMetric>
__device__ __forceinline__ void batch_distances(const HnswParams& p, const float* q, const uint32_t* ids, int cnt, float* dists, int lane) {
	const int m = lane & 15, g = lane >> 4;
	for (int base = 0; base < cnt; base += kRowsPerWave) {
		const int idx = base + g;
		const bool ok = idx < cnt;
		const uint64_t row = ids[ok ? idx : base];
		const float sum = group_distance_generic<kMetric>(p.rows + row * p.stride, q, p.dim, m);
		const float dist = 1.0f * metric_epilogue<kMetric>(sum, p.inv_norms, row);   // normCoef == 1 (hnswalg.h:1855-1863)
		if (ok && m == 0) dists[idx] = dist;
	}
}

// dim == 64*NB, SQ8: 16-byte loads.  Lane m of the row's 16-lane group reads bytes [256 i + 16 m, + 16) of load i: block 4 i + (m >> 2),
// bytes 16 (m & 3) .. + 16 of it, i.e. the element pairs of reference lanes j = 8 (m & 1) + p, p = 0 .. 7 (pair p = halfword p of the 16
// bytes) — in the low half of the block for (m & 3) < 2, in the high half above; both halves feed the same reference lane.  Eight integer
// accumulators a lane, over all blocks (the reference's lanes accumulate over all blocks too); a transpose-reduction over the eight lanes
// of equal parity (xor 2, 4, 8: 7 exchanges) leaves reference lane j's sum on lane (j >> 3) + 2 ((j >> 2) & 1) + 4 ((j >> 1) & 1) + 8 (j & 1),
// and the 16 sums are then added as floats in the reference's order j = 0 .. 15.  L2: (a - b)^2 = a^2 - 2ab + b^2 per pair, exact in
// uint32; the query's b^2 share (qq) is computed once a search.  The query words and qq live in registers for the whole search.
template <int kMetric, int NB>
__device__ __forceinline__ void batch_distances_sq8_fixed(const HnswParams& p, const uint4 (&qw)[(NB + 3) / 4], const uint32_t (&qq)[8], float qcorr,
														  float qnorm, const uint32_t* ids, int cnt, float* dists, int lane) {
	constexpr int L = (NB + 3) / 4;
	const int m = lane & 15, g = lane >> 4;
	const int group_base = lane & ~15;
	for (int base = 0; base < cnt; base += kRowsPerWave) {
		const int idx = base + g;
		const bool ok = idx < cnt;
		const uint64_t row = ids[ok ? idx : base];
		const uint4* r = reinterpret_cast<const uint4*>(p.codes + row * uint64_t(64 * NB));
		uint4 a[L];
#pragma unroll
		for (int i = 0; i < L; ++i) {
			if (256 * i + 256 <= 64 * NB || 256 * i + 16 * m < 64 * NB) {
				a[i] = r[16 * i + m];
			} else {
				a[i] = make_uint4(0u, 0u, 0u, 0u);
			}
		}
		uint32_t ab[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, aa[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
		for (int i = 0; i < L; ++i) {
			const uint32_t aw[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
			const uint32_t bw[4] = {qw[i].x, qw[i].y, qw[i].z, qw[i].w};
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const uint32_t lo = aw[c] & 0x0000FFFFu, hi = aw[c] & 0xFFFF0000u;
				ab[2 * c] = __builtin_amdgcn_udot4(lo, bw[c], ab[2 * c], false);
				ab[2 * c + 1] = __builtin_amdgcn_udot4(hi, bw[c], ab[2 * c + 1], false);
				if constexpr (kMetric == kL2) {
					aa[2 * c] = __builtin_amdgcn_udot4(lo, aw[c], aa[2 * c], false);
					aa[2 * c + 1] = __builtin_amdgcn_udot4(hi, aw[c], aa[2 * c + 1], false);
				}
			}
		}
		uint32_t s8[8];
#pragma unroll
		for (int q = 0; q < 8; ++q) s8[q] = kMetric == kL2 ? aa[q] + qq[q] - 2u * ab[q] : ab[q];
		const bool up1 = (m & 2) != 0, up2 = (m & 4) != 0, up3 = (m & 8) != 0;
		uint32_t t4[4], t2[2];
#pragma unroll
		for (int q = 0; q < 4; ++q) t4[q] = (up1 ? s8[q + 4] : s8[q]) + uint32_t(__shfl_xor(int(up1 ? s8[q] : s8[q + 4]), 2, 64));
#pragma unroll
		for (int q = 0; q < 2; ++q) t2[q] = (up2 ? t4[q + 2] : t4[q]) + uint32_t(__shfl_xor(int(up2 ? t4[q] : t4[q + 2]), 4, 64));
		const uint32_t mine = (up3 ? t2[1] : t2[0]) + uint32_t(__shfl_xor(int(up3 ? t2[0] : t2[1]), 8, 64));
		float result = 0.f;
#pragma unroll
		for (int j = 0; j < 16; ++j) {   // result += (float)lane[j], j = 0 .. 15
			const int holder = (j >> 3) + 2 * ((j >> 2) & 1) + 4 * ((j >> 1) & 1) + 8 * (j & 1);
			result += float(uint32_t(__shfl(int(mine), group_base + holder, 64)));
		}
		result = result + 0.f;   // + (float)tail, tail == 0: dim is a multiple of 64 (x + 0.f == x for every x the sum can take)
		float dist;
		if constexpr (kMetric == kL2) {
			dist = p.alpha2 * result + qcorr + p.corr[row];
		} else {
			dist = -(p.alpha2 * result + qcorr + p.corr[row]);
			if constexpr (kMetric == kCos) dist *= p.inv_norms[row];
		}
		dist = qnorm * dist;
		if (ok && m == 0) dists[idx] = dist;
	}
}

// dim == 64*NB: the query fragment lives in registers and the NB 16-byte loads of EIGHT rows (two per 16-lane group) are
// issued before the first reduction — one HBM round trip per 8 neighbours instead of three per 4.
// kQLds: the query fragment is re-read from LDS (ds_read_b128) instead of living in NB*4 VGPRs — 48 fewer registers at D = 768, which is
// what lets four of these wavefronts (instead of two) share a SIMD: the search is latency-bound, occupancy is throughput.
template <int kMetric, int NB, bool kQLds, bool kTwoSets>
__device__ __forceinline__ void batch_distances_fixed(const HnswParams& p, const float4 (&q)[kQLds ? 1 : NB], const float4* qlds, const uint32_t* ids,
													  int cnt, float* dists, int lane) {
	const int m = lane & 15, g = lane >> 4;
	auto Q = [&](int t) -> float4 {
		if constexpr (kQLds) {
			return qlds[16 * t + m];
		} else {
			return q[t];
		}
	};
	for (int base = 0; base < cnt; base += (kTwoSets ? 2 : 1) * kRowsPerWave) {
		const int ia = base + g, ib = base + kRowsPerWave + g;
		const bool oka = ia < cnt, okb = ib < cnt;
		const uint64_t ra = ids[oka ? ia : base], rb = ids[okb ? ib : base];
		const float4* pa = reinterpret_cast<const float4*>(p.rows + ra * p.stride) + m;
		const float4* pb = reinterpret_cast<const float4*>(p.rows + rb * p.stride) + m;
		float4 xa[NB], xb[NB];
#pragma unroll
		for (int t = 0; t < NB; ++t) xa[t] = pa[16 * t];
		const bool second = kTwoSets && base + kRowsPerWave < cnt;   // wave-uniform; one row set per trip keeps the D = 768 kernel at 4+ waves per SIMD
		if (second) {
#pragma unroll
			for (int t = 0; t < NB; ++t) xb[t] = pb[16 * t];
		}
		// cosine: 1 / |row| of the epilogue travels WITH the rows — read behind the scheduling barrier it was a dependent round trip of its
		// own at the end of every distance trip
		float inva = 1.0f, invb = 1.0f;
		if constexpr (kMetric == kCos) {
			inva = p.inv_norms[ra];
			if (second) invb = p.inv_norms[rb];
		}
		__builtin_amdgcn_sched_barrier(0);
		float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
		for (int t = 0; t < NB; ++t) chain_step<kMetric>(acc, Q(t), xa[t]);
		const float da = 1.0f * metric_epilogue<kMetric>(fold_chains<false>(acc, nullptr, nullptr, 0, m) + 0.0f, &inva, 0);
		if (oka && m == 0) dists[ia] = da;
		if (second) {
			acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
			for (int t = 0; t < NB; ++t) chain_step<kMetric>(acc, Q(t), xb[t]);
			const float db = 1.0f * metric_epilogue<kMetric>(fold_chains<false>(acc, nullptr, nullptr, 0, m) + 0.0f, &invb, 0);
			if (okb && m == 0) dists[ib] = db;
		}
	}
}

// The node of the nearest entry that is not expanded yet (what pop() would return next if nothing nearer is inserted before), or
// 0xFFFFFFFF: the search prefetches that node's link block while the current hop's distances are computed.
template <typename List, int S>
__device__ __forceinline__ uint32_t hnsw_peek_open(const List& list) {
	const int e = list.first_open();
	if (e < 0) return 0xFFFFFFFFu;
	uint32_t iv = list.id[0];
#pragma unroll
	for (int s = 1; s < S; ++s) iv = (e >> 6) == s ? list.id[s] : iv;
	return uint32_t(__builtin_amdgcn_readlane(int(iv), e & 63));
}

// the node of list entry e (wave-uniform e >= 0)
template <typename List, int S>
__device__ __forceinline__ uint32_t hnsw_node_at(const List& list, int e) {
	uint32_t iv = list.id[0];
#pragma unroll
	for (int s = 1; s < S; ++s) iv = (e >> 6) == s ? list.id[s] : iv;
	return uint32_t(__builtin_amdgcn_readlane(int(iv), e & 63));
}

typedef __attribute__((address_space(3))) void hnsw_lds_void;
// "Seen before?" of the reference's visited list (vl_type tags, hnswalg.h:904-931), as test-and-set.  Bitset: one atomicOr.  Hash set
// (HnswParams::vis_hash_log2): linear probing with compare-and-swap on node + 1; the neighbours a wavefront tests together are distinct
// nodes, two lanes can only meet on an EMPTY slot and the CAS settles that.  The table is at most half full (the callers check), so a probe
// sequence ends.  Its 32 KB stay in the L2 / Infinity Cache for the life of the search, where a bitset over 10M nodes (1.25 MB per
// search, 6 GB for the searches in flight) sends every test to HBM.
__device__ __forceinline__ bool hnsw_visit(uint32_t* visited, uint32_t hash_log2, uint32_t id) {
	if (hash_log2 == 0) {
		const uint32_t bit = 1u << (id & 31);
		return !(atomicOr(&visited[id >> 5], bit) & bit);
	}
	const uint32_t mask = (1u << hash_log2) - 1u, key = id + 1u;
	uint32_t h = (id * 2654435761u) >> (32u - hash_log2);
	for (;;) {
		const uint32_t old = atomicCAS(&visited[h], 0u, key);
		if (old == 0u) return true;
		if (old == key) return false;
		h = (h + 1u) & mask;
	}
}

// the latency form may keep the hash set in LDS (HnswParams::vis_lds): same probing, a ds_cmpst_rtn instead of a global atomic.  (Its own
// function, without the bitset branch: sharing hnsw_visit let the compiler merge the two bitset paths into ONE flat_atomic_or.)
__device__ __forceinline__ bool hnsw_visit_lds(uint32_t* table, uint32_t hash_log2, uint32_t id) {
	const uint32_t mask = (1u << hash_log2) - 1u, key = id + 1u;
	uint32_t h = (id * 2654435761u) >> (32u - hash_log2);
	for (;;) {
		const uint32_t old = atomicCAS(&table[h], 0u, key);
		if (old == 0u) return true;
		if (old == key) return false;
		h = (h + 1u) & mask;
	}
}
// "seen before?" WITHOUT the mark (the speculative distance batch of a team search looks ahead at a candidate's neighbours, but only the
// expansion itself may mark them)
__device__ __forceinline__ bool hnsw_seen(const uint32_t* visited, uint32_t hash_log2, uint32_t id) {
	if (hash_log2 == 0) return ((__hip_atomic_load(&visited[id >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (id & 31)) & 1u) != 0u;
	const uint32_t mask = (1u << hash_log2) - 1u, key = id + 1u;
	uint32_t h = (id * 2654435761u) >> (32u - hash_log2);
	for (;;) {
		const uint32_t v = __hip_atomic_load(&visited[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (v == key) return true;
		if (v == 0u) return false;
		h = (h + 1u) & mask;
	}
}
__device__ __forceinline__ bool hnsw_seen_lds(const uint32_t* table, uint32_t hash_log2, uint32_t id) {
	const uint32_t mask = (1u << hash_log2) - 1u, key = id + 1u;
	uint32_t h = (id * 2654435761u) >> (32u - hash_log2);
	for (;;) {
		const uint32_t v = table[h];
		if (v == key) return true;
		if (v == 0u) return false;
		h = (h + 1u) & mask;
	}
}
template <bool kLds>
__device__ __forceinline__ bool hnsw_visit_sel(uint32_t* visited, uint32_t* lds_vis, bool use_lds, uint32_t hash_log2, uint32_t id) {
	if constexpr (kLds) {
		if (use_lds) return hnsw_visit_lds(lds_vis, hash_log2, id);
	}
	return hnsw_visit(visited, hash_log2, id);
}

// kLatency: few queries in flight -> two row sets per distance trip (fewer dependent round trips per hop, 172 VGPRs at D = 768);
// otherwise one set (92 VGPRs: twice the resident searches).  D <= 512 always affords two.
// kSorted: 0 = the reference's two heaps, replayed by lane 0; S > 0 = HnswSortedList<S> (ef <= 64 S, no deleted nodes, heaps not in LDS at all)
// kDel (with kSorted): the graph has deleted nodes — HnswSortedListDel
// A search that leaves as kHnswOverflow also goes into the launch's overflow queue when there is one: a few helper workgroups launched
// beside the batch (hnsw_helper_kernel) pick it up and run it with the largest LDS heap WHILE the batch is still running — after the batch
// such a search is pure tail (one of 16 384 queries took 5 - 7 ms of a 32 ms batch at 10M x 768).
__device__ __forceinline__ void hnsw_enqueue_overflow(const HnswParams& p, uint32_t qi) {   // lane 0, behind the out_count store
	if (!p.helper_n) return;
	__threadfence();
	const uint32_t at = atomicAdd(p.helper_n, 1u);
	if (at < p.helper_cap) __hip_atomic_store(&p.helper_ids[at], p.q_base + qi + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// One search by one wavefront: slot = its scratch (visited set, global heap), qi = the query.  hnsw_search_kernel runs it once per
// workgroup, hnsw_helper_kernel in a loop over the overflow queue.
// A TEAM search (kTeam > 1 wavefronts per query, small launches): wavefront 0 runs the search below unchanged — list, visited set, link
// blocks — and the others only ever compute distances: at every distance batch the driver posts (ids, count) in the workgroup's box, all
// wavefronts meet at a barrier, each takes a slice of the rows (four rows a 16-lane group step, two sets a trip: 32 rows of 3 KB in flight
// per trip with four wavefronts instead of 8), and they meet again.  A hop's ~16 fresh neighbours are then ONE memory round trip instead of
// two or three — the search is a chain of ~140 dependent hops, and with a handful of queries on the chip nothing else hides them.  The
// driver's own LDS traffic is ordered wavefront-locally (HN_SYNC): the helpers do not take part in its barriers.
struct HnswTeamBox {
	const uint32_t* ids;
	float* dists;
	int cnt;     // < 0: the search is over
	int links;   // the ids are a hop's fresh neighbours: the other wavefronts also fetch their link blocks into the team's block area (p.nbl_off)
};
template <int kTeam>
__device__ __forceinline__ void hn_sync() {
	if constexpr (kTeam > 1) {
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__builtin_amdgcn_wave_barrier();
	} else {
		__syncthreads();
	}
}
// the rows wavefront `wave` of a team takes of a batch of cnt: [begin, begin + n), begin a multiple of 4
template <int kTeam>
__device__ __forceinline__ void team_slice(int cnt, int wave, int& begin, int& n) {
	const int per = (((cnt + kTeam - 1) / kTeam) + 3) & ~3;
	begin = wave * per;
	n = cnt - begin < per ? cnt - begin : per;
	if (n < 0) n = 0;
}
#define HN_SYNC() hn_sync<kTeam>()
template <int kMetric, bool kGlobalCand, int NB, bool kLatency, bool kSq8 = false, int kSorted = 0, bool kDel = false, int kTeam = 1>
__device__ __forceinline__ void hnsw_search_one(const HnswParams& p, const uint32_t slot, const uint32_t qi, HnswTeamBox* box = nullptr) {
	static_assert(kTeam == 1 || (NB > 0 && !kSq8 && !kGlobalCand), "the team form serves the fixed-dimension float search with its heaps in LDS");
	static_assert(kSorted == 0 || !kGlobalCand, "the sorted-list search starts in LDS; its re-runs with a global heap are heap-kernel launches");
	// dynamic LDS: [ef_cap] result heap (dist, id) then [lds_cand_cap] candidate heap (dist, id) — sized by the launcher so that
	// small-ef searches keep more wavefronts resident per CU
	extern __shared__ __attribute__((aligned(16))) unsigned char hnsw_lds[];
	uint2* top = reinterpret_cast<uint2*>(hnsw_lds);   // (dist bits, id) entries
	uint2* lcand = top + p.ef_cap;
	constexpr bool kQLds = NB > 0 && !kSq8;   // fixed dims: query fragment in LDS behind the heaps (16-byte aligned: every part is a multiple of 64 entries)
	float4* q_s = reinterpret_cast<float4*>(lcand + (kGlobalCand ? 0 : p.lds_cand_cap));
	uint32_t* lds_vis = reinterpret_cast<uint32_t*>(q_s + (kQLds ? NB * 16 : 0));   // [1 << vis_hash_log2] when p.vis_lds (latency form only)
	const bool vis_in_lds = kLatency && p.vis_lds != 0;
	__shared__ uint32_t nb_id[kHnswMaxNeighbors];
	__shared__ float nb_d[kHnswMaxNeighbors];
	__shared__ uint8_t nb_del[kHnswMaxNeighbors];
	__shared__ uint32_t s_cur;
	__shared__ int s_flag;
	__shared__ uint32_t s_pre[64];   // sorted-list search: the link block of the candidate next in line, fetched one hop ahead by LDS-DMA

	const int lane = threadIdx.x & 63;   // (a team's driver is wavefront 0)
	// the speculative team search (its hop below) keeps, at p.spec_off of the dynamic LDS: a direct-mapped table of distances computed ahead of
	// time (keys = node + 1, values), the link blocks of the two nearest open candidates, and the id / distance / destination arrays of a trip
	constexpr bool kSpecOk = kTeam > 1 && kSorted > 0 && !kDel && NB > 0 && !kSq8 && !kGlobalCand;
	constexpr uint32_t kSpecLog2 = kHnswSpecLog2;
	constexpr int kSpecRows = 8 * kTeam;   // rows of ONE distance trip of the team (two row sets of four rows a wavefront)
	uint32_t* spec_key = reinterpret_cast<uint32_t*>(hnsw_lds + p.spec_off);
	float* spec_val = reinterpret_cast<float*>(spec_key + (1u << kSpecLog2));
	uint32_t* spec_pre = spec_key + (2u << kSpecLog2);
	uint32_t* spec_id = spec_pre + 128;
	float* spec_d = reinterpret_cast<float*>(spec_id + 64);
	uint32_t* spec_dst = spec_id + 128;
	bool use_spec = false;
	if constexpr (kSpecOk) {
		use_spec = p.spec_off != 0u && p.maxM0 < 64u;
		if (use_spec) {
			for (uint32_t i = lane; i < (1u << kSpecLog2); i += 64) spec_key[i] = 0u;
		}
	}
	const float* q = p.queries + size_t(qi) * p.dim;
	uint32_t* visited = p.visited + size_t(slot) * p.visited_words;
	uint2* cand = kGlobalCand ? p.gcand + size_t(slot) * p.gcand_cap : lcand;
	const uint64_t cand_cap = kGlobalCand ? p.gcand_cap : uint64_t(p.lds_cand_cap);
	unsigned long long ndist = 0, hops = 0;
	float4 qreg[1];
	constexpr int kSqL = (kSq8 && NB > 0) ? (NB + 3) / 4 : 1;
	uint4 sq_q[kSqL];
	uint32_t sq_qq[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
	if constexpr (kSq8 && NB > 0) {
		const uint4* qp = reinterpret_cast<const uint4*>(p.qcodes + size_t(qi) * p.dim);
		const int m = lane & 15;
#pragma unroll
		for (int i = 0; i < kSqL; ++i) {
			sq_q[i] = (256 * i + 16 * m < 64 * NB) ? qp[16 * i + m] : make_uint4(0u, 0u, 0u, 0u);
			if constexpr (kMetric == kL2) {
				const uint32_t bw[4] = {sq_q[i].x, sq_q[i].y, sq_q[i].z, sq_q[i].w};
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					sq_qq[2 * c] = __builtin_amdgcn_udot4(bw[c] & 0x0000FFFFu, bw[c], sq_qq[2 * c], false);
					sq_qq[2 * c + 1] = __builtin_amdgcn_udot4(bw[c] & 0xFFFF0000u, bw[c], sq_qq[2 * c + 1], false);
				}
			}
		}
	}
	if constexpr (NB > 0 && !kSq8) {
		const float4* qp = reinterpret_cast<const float4*>(q);
		for (int i = lane; i < NB * 16; i += 64) q_s[i] = qp[i];
		HN_SYNC();
	}
	int team_links = 0;   // (team searches) 1 while the batches are a hop's fresh neighbours, see HnswTeamBox::links
	auto distances = [&](const uint32_t* ids, int cnt, float* dists) {
		if constexpr (kSq8 && NB > 0) {
			batch_distances_sq8_fixed<kMetric, NB>(p, sq_q, sq_qq, p.qcorr[qi], p.qnorm[qi], ids, cnt, dists, lane);
		} else if constexpr (kSq8) {
			batch_distances_sq8<kMetric>(p, p.qcodes + size_t(qi) * p.dim, p.qcorr[qi], p.qnorm[qi], ids, cnt, dists, lane);
		} else if constexpr (NB > 0 && kTeam > 1) {
			box->ids = ids;
			box->dists = dists;
			box->cnt = cnt;
			box->links = team_links;
			__syncthreads();   // the whole team: the batch is posted (and everything the driver wrote to LDS before it is visible)
			int b0, n0;
			team_slice<kTeam>(cnt, 0, b0, n0);
			batch_distances_fixed<kMetric, NB, kQLds, true>(p, qreg, q_s, ids + b0, n0, dists + b0, lane);
			__syncthreads();   // ... every slice is written
		} else if constexpr (NB > 0) {
			batch_distances_fixed<kMetric, NB, kQLds, (kLatency || NB <= 8)>(p, qreg, q_s, ids, cnt, dists, lane);
		} else {
			batch_distances<kMetric>(p, q, ids, cnt, dists, lane);
		}
	};

	const uint32_t vis_hash = p.vis_hash_log2;
	const unsigned long long vis_limit = vis_hash ? (1ull << (vis_hash - 1)) : ~0ull;   // entries the hash set may hold
	if (vis_hash) {   // the search zeroes its own (small) set: 16-byte stores, in flight during the descent through the upper levels
		if (vis_in_lds) {
			uint4* v4 = reinterpret_cast<uint4*>(lds_vis);
			for (uint32_t w = lane; w < (1u << vis_hash) / 4; w += 64) v4[w] = make_uint4(0u, 0u, 0u, 0u);
		} else {
			uint4* v4 = reinterpret_cast<uint4*>(visited);
			for (uint32_t w = lane; w < (1u << vis_hash) / 4; w += 64) v4[w] = make_uint4(0u, 0u, 0u, 0u);
		}
	}
	// ---- upper levels: greedy descent (getLayer0EntryPoint)
	uint32_t cur = p.entry;
	if (lane == 0) nb_id[0] = cur;
	HN_SYNC();
	distances(nb_id, 1, nb_d);
	HN_SYNC();
	float curdist = nb_d[0];
	ndist += 1;
	for (int level = p.maxlevel; level > 0; --level) {
		bool changed = true;
		while (changed) {
			HN_SYNC();
			const uint32_t* ll = p.upper + (p.upper_off[cur] + uint64_t(level - 1)) * (1 + p.M);
			const int cnt = int(ll[0]);
			for (int j = lane; j < cnt; j += 64) nb_id[j] = ll[1 + j];
			HN_SYNC();
			distances(nb_id, cnt, nb_d);
			HN_SYNC();
			ndist += cnt;
			changed = false;
			for (int i = 0; i < cnt; ++i) {   // uniform scalar-style scan (every lane computes the same thing)
				const float d = nb_d[i];
				if (d < curdist) {
					curdist = d;
					cur = nb_id[i];
					changed = true;
				}
			}
		}
	}

	const unsigned long long ndist_upper = ndist;
	if constexpr (kSorted > 0) {
		// ---- layer 0 on the sorted list
		using List = typename std::conditional<kDel, HnswSortedListDel<kSorted>, HnswSortedList<kSorted>>::type;
		List list;
		list.init(p.sorted == 1);
		const int ef = int(p.ef);
		if (!kDel || !p.deleted[cur]) {
			list.insert(curdist, cur, false, ef, lane);
			ndist += 1;   // the reference recomputes the entry distance here (same value)
		} else {
			list.insert(3.402823466e+38f, cur, true, ef, lane);   // a deleted entry point: candidate at FLT_MAX, lowerBound = FLT_MAX (:853-856)
		}
		if (vis_hash) {   // the zeroing stores have landed before the first test-and-set
			__threadfence();
			HN_SYNC();
		}
		if (lane == 0) (void)hnsw_visit_sel<kLatency>(visited, lds_vis, vis_in_lds, vis_hash, cur);
		// the link block of the candidate that is next in line, requested one hop ahead: it arrives while this hop's visited tests and row
		// gathers are in flight, and saves the next hop its first dependent round trip whenever no nearer candidate turned up meanwhile
		// (LDS-DMA: the block goes straight into s_pre, no register lives across the hop — the D = 768 kernel sits at its 96-VGPR budget)
		uint32_t pre_node = 0xFFFFFFFFu;
		uint32_t pre_tag0 = 0xFFFFFFFFu, pre_tag1 = 0xFFFFFFFFu;   // speculative team search: the nodes whose link blocks sit in spec_pre[0] / [1]
		unsigned long long spec_trips = 0;
		int nbl_rows = 0;                   // team searches: rows of the last hop whose link blocks sit in the team's block area
		unsigned long long nbl_hits = 0;
#ifdef RXGPU_HNSW_PHASES
		unsigned long long ph_a = 0, ph_b = 0, ph_c = 0, ph_t = __builtin_readcyclecounter(), ph_start = ph_t;
#define HN_PHASE(acc)                                              \
	do {                                                           \
		const unsigned long long now__ = __builtin_readcyclecounter(); \
		acc += now__ - ph_t;                                       \
		ph_t = now__;                                              \
	} while (0)
#else
#define HN_PHASE(acc) do { } while (0)
#endif
		for (;;) {
			if (ndist - ndist_upper + p.maxM0 > vis_limit) {   // the hash set would pass half full: this search goes to the bitset re-run
				if (lane == 0) {
					p.out_count[qi] = kHnswOverflow;
					hnsw_enqueue_overflow(p, qi);
				}
				return;
			}
			uint32_t node;
			float cdist;
			if (!list.pop(node, cdist, ef)) {   // candidate_set empty, or only dead entries left in it ...
				list.tie = list.tie || (list.held() == ef && list.lower == list.outside);   // ... of which the last one is still alive in the reference's
				break;
			}
			if (list.tie) break;                 // the rest of this search belongs to the heaps
			if (cdist > list.lower && (!kDel || list.held() == ef)) break;   // layer0ShouldStopBeforePop (never true for a member of a full list; kept for the form)
			hops += 1;
			int nfresh = 0;
			bool spec_hop = false;
			if constexpr (kSpecOk) spec_hop = use_spec;
			if (spec_hop) {
				if constexpr (kSpecOk) {
					// (RXGPU_HNSW_SPEC=1.  MEASURED, 1M x 768, ef = 128, one query at a time: 0.547 ms against 0.452 ms without it — only 23 % of the
					// hops run without a trip (the candidate expanded next is more often one this very hop inserted than the one in line before
					// it) and the bookkeeping below costs ~0.7 us a hop; profiles/rd6sp_single_1m.json.  Exact — fuzz and parity tests run it —
					// but off by default.)
					// ---- a hop of the SPECULATIVE team search.  The traversal is the reference's — pops, marks and insertions in its order, every
					// distance the same bits — but a distance need not be computed in the hop that uses it: d(query, row) does not change during a
					// search.  Each distance trip of the team (32 rows, four wavefronts) carries, next to the rows this hop must have, the unmarked
					// neighbours of the candidate NEXT in line, and their distances wait in a small direct-mapped table (spec_key / spec_val; a
					// collision only costs a re-evaluation).  When that candidate is popped — it is, unless this hop inserts something nearer, and
					// then it usually is a hop later — its rows are already there and the hop runs without a memory round trip.  The link blocks of
					// the two nearest open candidates are kept in LDS (two buffers, LDS-DMA), so the look-ahead has its list without a trip either.
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every link block requested so far has landed
					const int e1 = list.first_open();
					const uint32_t c1 = e1 >= 0 ? hnsw_node_at<List, kSorted>(list, e1) : 0xFFFFFFFFu;
					const int e2 = e1 >= 0 ? list.next_open(e1) : -1;
					const uint32_t c2 = e2 >= 0 ? hnsw_node_at<List, kSorted>(list, e2) : 0xFFFFFFFFu;
					const int nbuf = node == pre_tag0 ? 0 : (node == pre_tag1 ? 1 : -1);   // uniform
					int c1buf = c1 == 0xFFFFFFFFu ? -1 : (c1 == pre_tag0 ? 0 : (c1 == pre_tag1 ? 1 : -1));
					const uint32_t* ll = p.links0 + size_t(node) * (1 + p.maxM0);
					const uint32_t word = nbuf >= 0 ? spec_pre[64 * nbuf + lane] : (lane <= int(p.maxM0) ? ll[lane] : 0u);
					const uint32_t w1 = c1buf >= 0 ? spec_pre[64 * c1buf + lane] : 0u;
					const int cnt = int(__builtin_amdgcn_readfirstlane(word));
					const int cnt1 = int(__builtin_amdgcn_readfirstlane(w1));   // (both blocks are in registers: their buffers may be overwritten)
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					if (p.prefetch_links) {   // keep the blocks of the two nearest open candidates in LDS
						auto fetch = [&](int b, uint32_t nd) {
							const uint32_t* src = p.links0 + size_t(nd) * (1 + p.maxM0) + (lane <= int(p.maxM0) ? lane : int(p.maxM0));
							__builtin_amdgcn_global_load_lds(src, (hnsw_lds_void*)(spec_pre + 64 * b), 4, 0, 0);
							if (b == 0) {
								pre_tag0 = nd;
							} else {
								pre_tag1 = nd;
							}
						};
						int holds1 = c1buf;   // the buffer that holds (or is about to hold) c1's block
						if (c1 != 0xFFFFFFFFu && c1buf < 0) {
							holds1 = (c2 != 0xFFFFFFFFu && c2 == pre_tag0) ? 1 : 0;
							fetch(holds1, c1);
						}
						if (c2 != 0xFFFFFFFFu && holds1 >= 0 && c2 != pre_tag0 && c2 != pre_tag1) fetch(1 - holds1, c2);
					}
					// the popped node's neighbours: test-and-mark, in list order
					{
						const int j = lane - 1;
						bool fresh = false;
						if (j >= 0 && j < cnt) fresh = hnsw_visit_sel<kLatency>(visited, lds_vis, vis_in_lds, vis_hash, word);
						const uint64_t fm = __ballot(fresh);
						if (fresh) nb_id[__popcll(fm & ((1ull << lane) - 1))] = word;
						nfresh = __popcll(fm);
					}
					HN_SYNC();
					HN_PHASE(ph_a);
					// which of them were evaluated ahead of time
					int nmiss = 0;
					{
						const bool mine = lane < nfresh;
						const uint32_t id = mine ? nb_id[lane] : 0u;
						const uint32_t sl = (id * 2654435761u) >> (32u - kSpecLog2);
						const bool have = mine && spec_key[sl] == id + 1u;
						if (have) nb_d[lane] = spec_val[sl];
						const uint64_t mm = __ballot(mine && !have);
						if (mine && !have) {
							const int r = __popcll(mm & ((1ull << lane) - 1));
							spec_id[r] = id;
							spec_dst[r] = uint32_t(lane);
						}
						nmiss = __popcll(mm);
					}
					// ... and the look-ahead: neighbours of the next candidate that are neither marked nor evaluated yet, as far as the trip has room
					int total = nmiss;
					if (c1buf >= 0 && nmiss < kSpecRows) {
						const int j = lane - 1;
						bool want = false;
						if (j >= 0 && j < cnt1) {
							const bool seen = vis_in_lds ? hnsw_seen_lds(lds_vis, vis_hash, w1) : hnsw_seen(visited, vis_hash, w1);
							want = !seen && spec_key[(w1 * 2654435761u) >> (32u - kSpecLog2)] != w1 + 1u;
						}
						const uint64_t sm = __ballot(want);
						const int r = nmiss + __popcll(sm & ((1ull << lane) - 1));
						if (want && r < kSpecRows) {
							spec_id[r] = w1;
							spec_dst[r] = 0xFFFFFFFFu;
						}
						total = nmiss + __popcll(sm);
						total = total < kSpecRows ? total : kSpecRows;
					}
					HN_SYNC();
					if (total > 0) {
						distances(spec_id, total, spec_d);
						HN_SYNC();
						{
							const bool mine = lane < total;
							const float dv = mine ? spec_d[lane] : 0.f;
							const uint32_t dst = mine ? spec_dst[lane] : 0u;
							const bool ahead = mine && dst == 0xFFFFFFFFu;
							const uint32_t id = ahead ? spec_id[lane] : 0u;
							const uint32_t sl = (id * 2654435761u) >> (32u - kSpecLog2);
							if (mine && !ahead) nb_d[dst] = dv;
							// two look-ahead rows of one trip may share a table slot: the key decides which of them the slot keeps, and only that
							// lane writes the value (a key next to another row's value would be a wrong distance)
							if (ahead) spec_key[sl] = id + 1u;
							HN_SYNC();
							if (ahead && spec_key[sl] == id + 1u) spec_val[sl] = dv;
						}
						spec_trips += 1;
					}
					ndist += nfresh;
					HN_SYNC();
					HN_PHASE(ph_b);
				}
			} else {
				const uint32_t* ll = p.links0 + size_t(node) * (1 + p.maxM0);
				uint32_t first_word;
				uint64_t came_along = 0ull;   // (team searches) the popped node was one of the last hop's rows: its block came with them
				if constexpr (kTeam > 1) {
					if (nbl_rows > 0 && node != pre_node) came_along = __ballot(lane < nbl_rows && nb_id[lane] == node);   // nb_id still holds the last hop's rows
				}
				if (node == pre_node) {   // uniform: the block was requested a hop ago and has landed (every load issued since has been waited for)
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					first_word = s_pre[lane];
				} else if (came_along) {
					first_word = reinterpret_cast<const uint32_t*>(hnsw_lds + p.nbl_off)[64 * __builtin_ctzll(came_along) + lane];
					nbl_hits += 1;
				} else {
					first_word = lane <= int(p.maxM0) ? ll[lane] : 0u;
				}
				int cnt = 0;
				for (int base = 0; base <= int(p.maxM0); base += 64) {
					const int w = base + lane;
					const uint32_t word = base == 0 ? first_word : (w <= int(p.maxM0) ? ll[w] : 0u);
					if (base == 0) {
						cnt = int(__builtin_amdgcn_readfirstlane(word));   // (s_pre has been read: the next request may overwrite it)
						pre_node = p.prefetch_links ? hnsw_peek_open<List, kSorted>(list) : 0xFFFFFFFFu;
						if (pre_node != 0xFFFFFFFFu) {
							const uint32_t* src = p.links0 + size_t(pre_node) * (1 + p.maxM0) + (lane <= int(p.maxM0) ? lane : int(p.maxM0));
							__builtin_amdgcn_global_load_lds(src, (hnsw_lds_void*)s_pre, 4, 0, 0);
						}
					}
					if (base > cnt) break;   // uniform
					const int j = w - 1;
					bool fresh = false;
					if (j >= 0 && j < cnt) fresh = hnsw_visit_sel<kLatency>(visited, lds_vis, vis_in_lds, vis_hash, word);
					const uint64_t fm = __ballot(fresh);
					if (fresh) nb_id[nfresh + __popcll(fm & ((1ull << lane) - 1))] = word;
					nfresh += __popcll(fm);
				}
				HN_SYNC();
				HN_PHASE(ph_a);
				if constexpr (kDel) {   // the delete marks travel while the distances are computed
					for (int j = lane; j < nfresh; j += 64) nb_del[j] = p.deleted[nb_id[j]];
				}
				if constexpr (kTeam > 1) {
					team_links = (p.nbl_off != 0u && p.maxM0 < 64u) ? 1 : 0;
					nbl_rows = (team_links && nfresh > 0) ? (nfresh < kHnswNblRows ? nfresh : kHnswNblRows) : 0;
				}
				distances(nb_id, nfresh, nb_d);
				team_links = 0;
				ndist += nfresh;
				HN_SYNC();
				HN_PHASE(ph_b);
			}
			for (int base = 0; base < nfresh && !list.tie; base += 64) {   // runLayer0Step :932-960 in neighbour order; lane j carries neighbour base + j
				const int j = base + lane;
				const float dj = j < nfresh ? nb_d[j] : __builtin_inff();
				const uint32_t idj = j < nfresh ? nb_id[j] : 0u;
				const uint64_t delm = kDel ? __ballot(j < nfresh && nb_del[j] != 0) : 0ull;
				// lowerBound only falls while the list is full: what fails the test now fails it later in the loop as well
				uint64_t m = __ballot(j < nfresh && (list.held() < ef || list.lower > dj));
				if (list.held() == ef && __ballot(j < nfresh && dj == list.lower)) list.outside = list.lower;   // refused at dist == lowerBound
				while (m && !list.tie) {
					const int b = __builtin_ctzll(m);
					m &= m - 1;
					const float nd = lane_value(dj, b);
					if (list.held() < ef || list.lower > nd) {
						list.insert(nd, uint32_t(__builtin_amdgcn_readlane(int(idj), b)), ((delm >> b) & 1ull) != 0, ef, lane);
					} else if (nd == list.lower) {
						list.outside = nd;
					}
				}
			}
			HN_SYNC();
			HN_PHASE(ph_c);
		}
#ifdef RXGPU_HNSW_PHASES
		if (lane == 0 && p.stats) {
			atomicAdd(&p.stats[4], ph_a);
			atomicAdd(&p.stats[5], ph_b);
			atomicAdd(&p.stats[6], ph_c);
			atomicAdd(&p.stats[7], __builtin_readcyclecounter() - ph_start);
		}
#endif
		const int total = list.held();
		const int keep = total < int(p.k) ? total : int(p.k);   // SearchKnn :1998-2000: the k nearest of top_candidates
		if constexpr (!kDel) {
			if (!list.tie && total > keep && list.key_at(keep - 1) == list.key_at(keep)) list.tie = true;   // the trim pops one of two equal keys
		} else {
			if (!list.tie && total > keep && keep > 0 && list.key_at(list.live_at(keep - 1)) == list.key_at(list.live_at(keep))) list.tie = true;
		}
		if (!list.tie) {
			if constexpr (!kDel) {
#pragma unroll
				for (int s = 0; s < kSorted; ++s) {
					const int g = 64 * s + lane;
					if (g < keep) {
						p.out_dist[size_t(qi) * p.k + g] = list.d[s];
						p.out_row[size_t(qi) * p.k + g] = list.id[s];
					}
				}
			} else {
				int before = 0;   // live entries in the slots below
#pragma unroll
				for (int s = 0; s < kSorted; ++s) {
					const uint64_t lm = ~list.del[s] & List::below_count(s, list.n);
					const int r = before + __popcll(lm & ((1ull << lane) - 1ull));
					if (((lm >> lane) & 1ull) && r < keep) {
						p.out_dist[size_t(qi) * p.k + r] = list.d[s];
						p.out_row[size_t(qi) * p.k + r] = list.id[s];
					}
					before += __popcll(lm);
				}
			}
			if (lane == 0) {
				p.out_count[qi] = uint32_t(keep);
				if (p.stats) {
					atomicAdd(&p.stats[0], ndist);
					atomicAdd(&p.stats[1], hops);
					if (nbl_hits) atomicAdd(&p.stats[3], nbl_hits << 32);   // (high half: hops whose link block had come along with the previous hop's rows)
					if (spec_trips) atomicAdd(&p.stats[3], spec_trips);   // distance trips of the speculative team search (< hops: the rest ran on distances computed ahead)
				}
			}
			return;
		}
		// Equal keys that matter: this search starts over on the reference's heaps, here and now — while the rest of the batch keeps the
		// chip busy — instead of in a launch of its own after the batch (whose whole duration is the tail of ONE heap search).  The
		// launcher gives every workgroup a small heap area for that (lds_cand_cap entries; a search that outgrows it is flagged
		// kHnswOverflow and re-run with the global heap like any other); without one the query goes back to the host as kHnswTie.
		if (p.lds_cand_cap == 0) {
			if (lane == 0) p.out_count[qi] = kHnswTie;
			return;
		}
		if (vis_in_lds) {
			for (uint32_t w = lane; w < (1u << vis_hash); w += 64) lds_vis[w] = 0u;
		} else {
			for (uint64_t w = lane; w < p.visited_words; w += 64) visited[w] = 0u;
		}
		__threadfence();
		HN_SYNC();
		if (lane == 0 && p.stats) atomicAdd(&p.stats[2], 1ull);
		ndist = ndist_upper;
		hops = 0;
	}

	// ---- layer 0: initLayer0SearchState
	int top_n = 0, cand_n = 0;
	float lower;
	bool overflow = false;
	if (vis_hash) {   // the zeroing stores (kernel start, or the restart above) have landed before the first test-and-set
		__threadfence();
		HN_SYNC();
	}
	{
		const bool ep_ok = p.bare || !p.deleted[cur];
		if (lane == 0) {
			if (ep_ok) {
				hp_emplace(top, top_n, curdist, cur);
				hp_emplace(cand, cand_n, -curdist, cur);
			} else {
				hp_emplace(cand, cand_n, -3.402823466e+38f, cur);
			}
			(void)hnsw_visit_sel<kLatency>(visited, lds_vis, vis_in_lds, vis_hash, cur);
		}
		lower = ep_ok ? curdist : 3.402823466e+38f;
		if (ep_ok) ndist += 1;   // the reference recomputes the entry distance here (same value)
	}

	for (;;) {
		// layer0ShouldStopBeforePop + pop (lane 0), broadcast through LDS
		const bool vis_full = ndist - ndist_upper + p.maxM0 > vis_limit;   // uniform: the hash set would pass half full
		if (lane == 0) {
			int flag = 0;
			if (vis_full) overflow = true;
			if (cand_n == 0 || overflow) {
				flag = 1;
			} else {
				const uint2 best = cand[0];
				const float cdist = -hp_dist(best);
				if (p.bare ? (cdist > lower) : (cdist > lower && top_n >= int(p.ef))) {
					flag = 1;
				} else {
					s_cur = best.y;
					hp_pop(cand, cand_n);
				}
			}
			s_flag = flag;
		}
		HN_SYNC();
		if (s_flag) break;
		const uint32_t node = s_cur;
		hops += 1;
		// neighbours: the whole list block (count word + up to 2M ids) is fetched in ONE round — lane L reads word L of the block, so the
		// count arrives together with the ids instead of in front of them; atomicOr = visited test + mark
		const uint32_t* ll = p.links0 + size_t(node) * (1 + p.maxM0);
		int nfresh = 0;
		int cnt = 0;
		for (int base = 0; base <= int(p.maxM0); base += 64) {
			const int w = base + lane;   // word of the block; neighbour index j = w - 1
			const uint32_t word = w <= int(p.maxM0) ? ll[w] : 0u;
			if (base == 0) cnt = int(__builtin_amdgcn_readfirstlane(word));
			if (base > cnt) break;   // uniform
			const int j = w - 1;
			bool fresh = false;
			if (j >= 0 && j < cnt) fresh = hnsw_visit_sel<kLatency>(visited, lds_vis, vis_in_lds, vis_hash, word);
			const uint64_t fm = __ballot(fresh);
			if (fresh) nb_id[nfresh + __popcll(fm & ((1ull << lane) - 1))] = word;
			nfresh += __popcll(fm);
		}
		HN_SYNC();
		distances(nb_id, nfresh, nb_d);
		if (!p.bare) {
			for (int j = lane; j < nfresh; j += 64) nb_del[j] = p.deleted[nb_id[j]];
		}
		ndist += nfresh;
		HN_SYNC();
		if (lane == 0) {   // sequential heap updates in neighbour order (runLayer0Step :932-960)
			for (int i = 0; i < nfresh; ++i) {
				const float d = nb_d[i];
				const uint32_t id = nb_id[i];
				if (top_n < int(p.ef) || lower > d) {
					if (uint64_t(cand_n) >= cand_cap) {
						overflow = true;
						break;
					}
					hp_emplace(cand, cand_n, -d, id);
					if (p.bare || !nb_del[i]) {
						if (top_n < int(p.ef)) {
							hp_emplace(top, top_n, d, id);
						} else {
							hp_replace_top(top, top_n, d, id);
						}
					}
					if (top_n) lower = hp_dist(top[0]);
				}
			}
		}
		HN_SYNC();
	}

	if (lane == 0) {
		if (overflow) {
			p.out_count[qi] = kHnswOverflow;
			if constexpr (!kGlobalCand) hnsw_enqueue_overflow(p, qi);
		} else {
			while (top_n > int(p.k)) hp_pop(top, top_n);   // SearchKnn :1998-2000
			for (int i = 0; i < top_n; ++i) {
				p.out_dist[size_t(qi) * p.k + i] = hp_dist(top[i]);
				p.out_row[size_t(qi) * p.k + i] = top[i].y;
			}
			p.out_count[qi] = uint32_t(top_n);
		}
		if (p.stats) {
			atomicAdd(&p.stats[0], ndist);
			atomicAdd(&p.stats[1], hops);
		}
	}
}

#undef HN_SYNC

// The other wavefronts of a team: distance batches until the driver says the search is over.
template <int kMetric, int NB, int kTeam>
__device__ __forceinline__ void hnsw_team_serve(const HnswParams& p, const HnswTeamBox* box) {
	extern __shared__ __attribute__((aligned(16))) unsigned char hnsw_lds[];
	const float4* q_s = reinterpret_cast<const float4*>(reinterpret_cast<uint2*>(hnsw_lds) + p.ef_cap + p.lds_cand_cap);   // as hnsw_search_one lays it out
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	float4 qreg[1];
	for (;;) {
		__syncthreads();
		const int cnt = box->cnt;
		if (cnt < 0) return;
		int b0, n0;
		team_slice<kTeam>(cnt, wave, b0, n0);
		// (RXGPU_HNSW_NBL=1, an experiment that is OFF by default.)  The candidate expanded next is often one of the rows of THIS batch (a
		// neighbour that turns out nearer than everything in line), which the driver's one-hop-ahead prefetch cannot know.  So the link blocks of
		// the batch's rows come along with the rows and the next hop finds its block in LDS whichever row it pops.  MEASURED: 15 % of the hops
		// at 1M x 768 find their block this way, but the 16 extra 132-byte gathers a hop delay the row gathers behind them: one query 0.452 ->
		// 0.517 ms at 1M, 0.67 -> 1.05 ms at 10M rows, where every gather starts with a page walk and the walks are what a hop waits for
		// (profiles/rd6d_*_linkblocks.json).
		if (box->links && p.nbl_off) {
			uint32_t* nbl = reinterpret_cast<uint32_t*>(hnsw_lds + p.nbl_off);
			const int rows = cnt < kHnswNblRows ? cnt : kHnswNblRows;
			for (int j = wave - 1; j < rows; j += kTeam - 1) {
				const uint32_t* src = p.links0 + size_t(box->ids[j]) * (1 + p.maxM0) + (lane <= int(p.maxM0) ? lane : int(p.maxM0));
				__builtin_amdgcn_global_load_lds(src, (hnsw_lds_void*)(nbl + 64 * j), 4, 0, 0);
			}
		}
		if (n0 > 0) batch_distances_fixed<kMetric, NB, true, true>(p, qreg, q_s, box->ids + b0, n0, box->dists + b0, lane);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a wavefront without rows of its own still has blocks in flight)
		__syncthreads();
	}
}
}  // namespace rxgpu
