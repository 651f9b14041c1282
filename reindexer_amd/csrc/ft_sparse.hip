// ft_fast merge on gfx950, the train for SPARSELY hit document ranges (Merger::Merge, cpp_src/core/ft/ft_fast/mergerimpl.h:466-566, for the
// queries ft_sparse_eligible() in rxgpu_ft_capi.hip admits: plain terms whose every field has the same positive boost, at most kFtSparseSubs
// sub-terms, postings on a fraction of the documents).
//
// The dense train (ft_merge.hip) gives every document range of 8192 documents a 256-thread workgroup and keeps per-DOCUMENT arrays in HBM
// between its kernels (16-bit pre-scores, the restricting mask, 16-byte records of every ranked posting).  A two-term query over 5M
// documents touches 8 % of them: 650 postings per range, for which a workgroup walks ~20 us of dependent phases and the arrays move 4 x the
// bytes of the postings (profiles/rd5_bm25_train64_sparse_rocprof.json).  Here the unit of work is ONE WAVEFRONT per (query, range), four to a
// workgroup, no workgroup barrier and no shared descriptor copy in a unit, and nothing per document or per posting is written to HBM except
// one small task per MERGED document:
//
//  * every unit pulls the document ids of its range's postings (all sub-terms, one flat index space, eight independent loads per lane in
//    flight) and sets one bit per (sub-term, document) in LDS: kS bitmaps of 8192 bits.  Every per-document fact of the merge is then a
//    word-parallel expression over the bitmaps, lane l owning words l, l + 64, l + 128, l + 192 of each:
//      restrictingMask_ (buildRestrictingBitmask :326-384)   valid & ~excluded & AND-terms' unions & ~NOT sub-terms
//      calcTermScores (:289-324)                              per term the FIRST sub-term holding the document adds its proc16 (claim words:
//                                                             W[si] & ~seen-in-this-term), saturating sum = min(sum, 65535)
//      addDoc order (merger.h:161-180)                        a document is added by the first non-NOT sub-term holding it (every rank is
//                                                             positive: checked on the host), so "first met in row r" = W[r] & ~seen-so-far;
//                                                             its merge slot = documents first met in earlier rows + same row, smaller id
//      posting index of a document in a sub-term              segment start + set bits in front of it (popcount prefix along the bitmap)
//  * the global facts order the kernels: the pre-score histogram -> threshold (ft_sp_scan | ft_sp_threshold), the ties kept at the threshold
//    in document order (ft_sp_select's ordered count over the units) and the table of documents first met per (row, range) -> slot bases
//    (ft_slot_bases).  Each unit kernel rebuilds its bitmaps from the postings (4 B per posting, L2-warm the second time).
//  * calcTermRank + the per-document replay (mergeTerm :107-192 / mergeSimple :194-250) run LAST and document-parallel, in ft_sp_replay: one
//    thread per merged document reads its task (document, slot or (row, rank in the range), posting index per sub-term row) — a
//    whole-corpus single-term merge ranks its 20 000 merged documents, not its 600 000 postings, and no range is a straggler.
//
// Launch train: ft_sp_scan, [ft_sp_threshold, ft_sp_select — queries that may preselect: they come first in the batch], ft_slot_bases
// (ft_merge.hip), [ft_sp_place — the others], ft_sp_replay.
// Results are the dense train's to the bit (same float operations per document, same slots): tests/test_gpu_ft_sparse.py runs both.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "rxgpu_internal.h"
#include "knn_kernels.hip.h"
#include "ft_rank.hip.h"
#include "ft_scan.hip.h"
#include "ft_replay.hip.h"

namespace rxgpu {
namespace {

constexpr uint32_t kSpWords = kFtRangeDocs / 32;   // mask words of one range
constexpr uint32_t kSpUnits = 4;                   // units (wavefronts) per workgroup
constexpr uint32_t kSpLoads = 8;                   // posting loads a lane keeps in flight while the bitmaps are built
static_assert(kSpWords == 4 * 64, "a lane owns four words of every bitmap");

// sub-term attribute word (FtPlan::SpSub::attr, built on the host)
constexpr uint32_t kSpFirst = 1u << 16, kSpAnd = 1u << 17, kSpNot = 1u << 18;

// LDS traffic between the lanes of ONE wavefront: the hardware executes a wavefront's LDS instructions in order, the fence keeps the compiler
// from moving them across
__device__ __forceinline__ void sp_fence() {
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t sp_readlane(uint32_t v, int lane) { return uint32_t(__builtin_amdgcn_readlane(int(v), lane)); }
// Inclusive prefix sum over the wavefront on the DPP path: four row_shr steps inside the rows of 16 lanes, then lane 15 of rows 0 / 2 into
// rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31) — twelve VALU instructions, no LDS crossbar (the unit kernels
// are bound by instruction issue, and a ds_bpermute scan costs three times that).
__device__ __forceinline__ uint32_t sp_scan(uint32_t v) {
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
	v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
	return v;
}

// What a unit knows about its query (uniform over the wavefront) and its range
template <int kS>
struct SpCtx {
	uint32_t* bits;             // LDS [kS][kSpWords]
	uint32_t n_subs;
	uint32_t attr;              // lane si: attribute word of sub-term si
	uint32_t lo;                // lane si: first posting of the range in sub-term si
	unsigned long long first_m, and_m, not_m;
	uint32_t p16[kS];           // proc16 of sub-term si (uniform)
	bool empty_and;
	uint32_t range, d_begin, docs_here;
	uint32_t rm[4], ex[4];      // removed / excluded bits of the lane's four words
};

// The unit's bitmaps: one bit per (sub-term, document of the range).  Lane si reads sub-term si's entry of the plan and the range's segment
// of its list; the postings of all sub-terms form one flat index space, a lane keeps kSpLoads document loads in flight.
template <int kS>
__device__ __forceinline__ void sp_open(FtPlanK& p, uint32_t* unit_lds, uint32_t range, int lane, SpCtx<kS>& c) {
	c.bits = unit_lds;
	c.n_subs = p.n_subs;
	c.range = range;
	c.d_begin = range << kFtRangeShift;
	const uint64_t left = p.total_docs - uint64_t(c.d_begin);
	c.docs_here = uint32_t(left < kFtRangeDocs ? left : kFtRangeDocs);
	c.empty_and = p.sp_empty_and != 0;
	const uint32_t* docp = nullptr;
	uint32_t lo = 0, hi = 0;
	c.attr = 0;
	if (uint32_t(lane) < c.n_subs) {
		const auto& s = p.sp_sub[lane];
		docp = s.doc;
		c.attr = s.attr;
		lo = range < s.n_ranges ? s.range_off[range] : s.n;
		hi = range + 1 < s.n_ranges ? s.range_off[range + 1] : s.n;
	}
	c.lo = lo;
#pragma unroll
	for (int j = 0; j < 4; ++j) {   // requested now, consumed behind the bitmaps
		const uint64_t gw = uint64_t(c.d_begin) / 32 + uint32_t(64 * j + lane);
		const bool in = gw < p.nwords;
		c.rm[j] = (p.removed_bits && in) ? p.removed_bits[gw] : 0u;
		c.ex[j] = (p.excluded_bits && in) ? p.excluded_bits[gw] : 0u;
	}
	c.first_m = __ballot((c.attr & kSpFirst) != 0);
	c.and_m = __ballot((c.attr & kSpAnd) != 0);
	c.not_m = __ballot((c.attr & kSpNot) != 0);
#pragma unroll
	for (int si = 0; si < kS; ++si) c.p16[si] = sp_readlane(c.attr, si) & 0xFFFFu;
	uint4* b4 = reinterpret_cast<uint4*>(c.bits);
	for (uint32_t k = uint32_t(lane); k < kS * kSpWords / 4; k += 64) b4[k] = make_uint4(0u, 0u, 0u, 0u);
	const uint32_t len = hi - lo;
	const uint32_t incl = sp_scan(len);
	const uint32_t cum = incl - len;                     // lane si: postings of the range in front of sub-term si
	const uint32_t total = sp_readlane(incl, 63);
	uint32_t cum_u[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) cum_u[si] = sp_readlane(cum, si);   // (lanes past the last sub-term: total)
	const uint64_t doc_at = reinterpret_cast<uint64_t>(docp) + uint64_t(lo) * 4;   // lane si: address of the segment's first document id
	const uint32_t doc_lo = uint32_t(doc_at), doc_hi = uint32_t(doc_at >> 32);
	sp_fence();
	for (uint32_t base = 0; base < total; base += 64 * kSpLoads) {
		uint32_t d[kSpLoads], sub[kSpLoads];
		bool ok[kSpLoads];
#pragma unroll
		for (uint32_t k = 0; k < kSpLoads; ++k) {
			const uint32_t f = base + k * 64 + uint32_t(lane);
			ok[k] = f < total;
			uint32_t s = 0, cs = 0;
#pragma unroll
			for (int si = 1; si < kS; ++si) {   // the last sub-term that starts at or in front of f (empty ones in between share its start)
				const bool ge = f >= cum_u[si];
				s = ge ? uint32_t(si) : s;
				cs = ge ? cum_u[si] : cs;
			}
			sub[k] = s;
			const uint64_t a = (uint64_t(uint32_t(__shfl(int(doc_hi), int(s), 64))) << 32) | uint32_t(__shfl(int(doc_lo), int(s), 64));
			d[k] = 0;
			if (ok[k]) d[k] = reinterpret_cast<const uint32_t*>(a)[f - cs];
		}
#pragma unroll
		for (uint32_t k = 0; k < kSpLoads; ++k) {
			if (!ok[k]) continue;
			const uint32_t local = d[k] - c.d_begin;
			__hip_atomic_fetch_or(&c.bits[sub[k] * kSpWords + (local >> 5)], 1u << (local & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	sp_fence();
}

// The lane's word 64 j + lane of every bitmap and what follows from them
template <int kS>
struct SpWord {
	uint32_t W[kS];
	uint32_t mask;   // restrictingMask_
	uint32_t cand;   // ... & not removed & held by a sub-term that is merged: the documents the merge can add
};
template <int kS>
__device__ __forceinline__ void sp_word(const SpCtx<kS>& c, int j, int lane, SpWord<kS>& o) {
	const uint32_t w = uint32_t(64 * j + lane);
#pragma unroll
	for (int si = 0; si < kS; ++si) o.W[si] = uint32_t(si) < c.n_subs ? c.bits[uint32_t(si) * kSpWords + w] : 0u;
	const uint32_t d0 = w * 32;
	uint32_t valid = 0;
	if (d0 < c.docs_here) valid = c.docs_here - d0 >= 32 ? 0xFFFFFFFFu : ((1u << (c.docs_here - d0)) - 1u);
	uint32_t mask = valid & ~c.ex[j], tm = 0, notw = 0, any = 0;
	bool in_and = false;
#pragma unroll
	for (int si = 0; si < kS; ++si) {
		if (uint32_t(si) >= c.n_subs) continue;
		if ((c.first_m >> si) & 1ull) {   // calcTermBitmask of the term that just ended (mergerimpl.h:252-274): every occurrence is relevant
			if (in_and) mask &= tm;
			tm = 0;
			in_and = ((c.and_m >> si) & 1ull) != 0;
		}
		tm |= o.W[si];
		if ((c.not_m >> si) & 1ull) {
			notw |= o.W[si];   // excludeTermFromBitmask (:276-287)
		} else {
			any |= o.W[si];
		}
	}
	if (in_and) mask &= tm;
	mask &= ~notw;
	if (c.empty_and) mask = 0;
	o.mask = mask;
	o.cand = mask & ~c.rm[j] & any;
}
// calcTermScores: per term the first sub-term (SortSubterms order) holding the document scores it
template <int kS>
__device__ __forceinline__ void sp_term_claims(const SpCtx<kS>& c, const SpWord<kS>& w, uint32_t (&claim)[kS]) {
	uint32_t seen = 0;
#pragma unroll
	for (int si = 0; si < kS; ++si) {
		claim[si] = 0;
		if (uint32_t(si) >= c.n_subs) continue;
		if ((c.first_m >> si) & 1ull) seen = 0;
		if ((c.not_m >> si) & 1ull) continue;
		claim[si] = w.W[si] & ~seen;
		seen |= w.W[si];
	}
}
template <int kS>
__device__ __forceinline__ uint32_t sp_score(const SpCtx<kS>& c, const uint32_t (&claim)[kS], uint32_t b) {
	uint32_t sc = 0;
#pragma unroll
	for (int si = 0; si < kS; ++si) sc += ((claim[si] >> b) & 1u) * c.p16[si];
	return sc < 65535u ? sc : 65535u;   // each term adds min(proc16, 65535 - score so far): the saturating sum
}
// addDoc: the first merged sub-term (row order) holding the document
template <int kS>
__device__ __forceinline__ void sp_first_met(const SpCtx<kS>& c, const SpWord<kS>& w, uint32_t kept, uint32_t (&fm)[kS]) {
	uint32_t seen = 0;
#pragma unroll
	for (int si = 0; si < kS; ++si) {
		fm[si] = 0;
		if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
		fm[si] = w.W[si] & ~seen & kept;
		seen |= w.W[si];
	}
}

// A lane's own pre-score counts: four (score, count) pairs in registers, compared without a cross-lane operation per document (a query has
// a handful of distinct scores: sums of its sub-terms' proc16); a fifth distinct score in one lane goes straight to HBM
struct SpLaneKeys {
	uint32_t k[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0};
	template <typename Overflow>
	__device__ __forceinline__ void add(uint32_t sc, Overflow&& overflow) {   // sc != 0
		bool done = false;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const bool hit = !done && (k[i] == sc || c[i] == 0);
			k[i] = hit ? sc : k[i];
			c[i] += hit ? 1u : 0u;
			done = done || hit;
		}
		if (!done) overflow(sc, 1u);
	}
};
// The distinct pre-scores of a unit with their document counts, one (score, count) per lane; more than 64 distinct scores go straight to HBM
struct SpKeys {
	uint32_t key = 0, cnt = 0;
	uint32_t n = 0;   // uniform
};
template <typename Overflow>
__device__ __forceinline__ void sp_keys_add(SpKeys& t, bool have, uint32_t sc, uint32_t weight, int lane, Overflow&& overflow) {
	unsigned long long pending = __ballot(have);
	while (pending) {
		const int leader = __ffsll((long long)pending) - 1;
		const uint32_t v = uint32_t(__shfl(int(sc), leader, 64));
		const bool mine = have && sc == v;
		const unsigned long long same = __ballot(mine);
		const uint32_t c = wave_sum(mine ? weight : 0u);
		const unsigned long long found = __ballot(uint32_t(lane) < t.n && t.key == v);
		if (found) {
			if (lane == __ffsll((long long)found) - 1) t.cnt += c;
		} else if (t.n < 64) {
			if (uint32_t(lane) == t.n) {
				t.key = v;
				t.cnt = c;
			}
			++t.n;
		} else if (lane == 0) {
			overflow(v, c);
		}
		pending &= ~same;
	}
}

// the threshold of preselectMostRelevantDocs as ft_sp_threshold left it
struct SpThreshold {
	bool on, all_ties;
	uint32_t score, docs;
};
__device__ __forceinline__ SpThreshold sp_threshold(FtPlanK& p) {
	SpThreshold t{};
	if (!p.prescore) return t;
	const uint32_t flags = p.sync[kFtSyncThrFlags];
	t.on = (flags & 1u) != 0;
	t.all_ties = (flags & 2u) != 0;
	t.score = p.sync[kFtSyncThrScore];
	t.docs = p.sync[kFtSyncThrDocs];
	return t;
}
__device__ __forceinline__ uint32_t sp_lowest_bits(uint32_t x, uint32_t n) {   // the n lowest set bits of x
	uint32_t out = 0;
	while (n && x) {
		const uint32_t low = x & (0u - x);
		out |= low;
		x ^= low;
		--n;
	}
	return out;
}

// One task per merged document (FtPlan::t_doc / t_pos / t_idx): its posting index in every merged sub-term, from the lane's own words —
// segment start + bits of the sub-term's bitmap in front of the document (pre[si]: in front of this word).
template <int kS>
__device__ __forceinline__ void sp_emit(FtPlanK& p, const SpCtx<kS>& c, const SpWord<kS>& w, const uint32_t (&pre)[kS], const uint32_t (&lo_u)[kS], uint32_t t,
										uint32_t word, uint32_t b, uint32_t pos) {
	p.t_doc[t] = c.d_begin + word * 32 + b;
	p.t_pos[t] = pos;
	uint32_t* row = p.t_idx + size_t(t) * p.n_rows;
	const uint32_t below = (1u << b) - 1u;
#pragma unroll
	for (int si = 0; si < kS; ++si) {
		if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
		const uint32_t r = sp_readlane(c.attr, si) >> 20;
		row[r] = ((w.W[si] >> b) & 1u) ? lo_u[si] + pre[si] + uint32_t(__popc(w.W[si] & below)) + 1u : 0u;
	}
}

// the table of documents first met per (row, range): this unit's column, zeros included (the table is the merge's scratch)
// (kPerLane: rows[] holds every lane's own count; otherwise the unit's totals, the same in every lane)
template <int kS, bool kPerLane>
__device__ __forceinline__ void sp_store_rows(FtPlanK& p, const SpCtx<kS>& c, const uint32_t (&rows)[kS], int lane) {
#pragma unroll
	for (int si = 0; si < kS; ++si) {
		if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
		const uint32_t tot = kPerLane ? wave_sum(rows[si]) : rows[si];
		if (lane == 0) p.adders[uint64_t(sp_readlane(c.attr, si) >> 20) * p.n_ranges + c.range] = tot;
	}
}

// ---------------------------------------------------------------------------------------------- ft_sp_scan
// prescore: the restricting mask's popcount and the pre-score histogram of the unit's documents (the input of the 2-phase gate and of the
// threshold).  Otherwise (Simple() queries, queries below mergeLimit): the documents first met per (row, range) — ft_adders' table — at once.
template <int kS>
__device__ __forceinline__ void sp_scan_unit(FtPlanK& p, uint32_t* unit_lds, uint32_t range, int lane) {
	SpCtx<kS> c;
	sp_open<kS>(p, unit_lds, range, lane, c);
	uint32_t pop = 0;
	uint32_t rows[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) rows[si] = 0;
	// (the sparse train counts into histogram copy 0 alone: a unit adds each of its few distinct scores once, and ft_sp_threshold, which
	// clears what it has read, then has one copy to clear)
	uint32_t* hist_copy = p.hist;
	auto overflow = [&](uint32_t v, uint32_t n) {
		atomicAdd(&hist_copy[v], n);
		atomicAdd(&hist_copy[65536 + (v >> 6)], n);
	};
	SpLaneKeys mine;
	for (int j = 0; j < 4; ++j) {
		SpWord<kS> w;
		sp_word<kS>(c, j, lane, w);
		if (p.prescore) {
			pop += uint32_t(__popc(w.mask));
			uint32_t claim[kS];
			sp_term_claims(c, w, claim);
			uint32_t rem = w.cand;
			while (rem) {
				const uint32_t b = uint32_t(__ffs(int(rem)) - 1);
				rem &= rem - 1;
				const uint32_t sc = sp_score(c, claim, b);
				if (sc) mine.add(sc, overflow);   // (score 0 is not counted: mergerimpl.h:433 walks scores >= 1)
			}
		} else {
			uint32_t fm[kS];
			sp_first_met(c, w, w.cand, fm);
#pragma unroll
			for (int si = 0; si < kS; ++si) rows[si] += uint32_t(__popc(fm[si]));
		}
	}
	if (p.prescore) {
		pop = wave_sum(pop);
		if (lane == 0 && pop) atomicAdd(&p.sync[kFtSyncPop], pop);
		SpKeys keys;   // the lanes' tables folded into one per unit: one atomic per distinct score and unit
#pragma unroll
		for (int i = 0; i < 4; ++i) sp_keys_add(keys, mine.c[i] != 0, mine.k[i], mine.c[i], lane, overflow);
		if (uint32_t(lane) < keys.n) overflow(keys.key, keys.cnt);
	} else {
		sp_store_rows<kS, true>(p, c, rows, lane);
	}
}
// A batch is launched for its widest query (kMax bitmaps per unit in LDS); a unit of a narrower query runs the narrower code: the kernels
// are bound by instruction issue, and most of their loops run over the sub-terms.
#define SP_DISPATCH(kMax, n_subs, call4, call8, callmax) \
	do {                                                 \
		if (kMax > 4 && (n_subs) <= 4) {                 \
			call4;                                       \
		} else if (kMax > 8 && (n_subs) <= 8) {          \
			call8;                                       \
		} else {                                         \
			callmax;                                     \
		}                                                \
	} while (0)
template <int kMax>
__global__ __launch_bounds__(256) void ft_sp_scan(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
	const int lane = threadIdx.x & 63;
	const uint32_t unit = threadIdx.x >> 6, range = blockIdx.x * kSpUnits + unit;
	if (range >= p.n_ranges) return;
	uint32_t* unit_lds = sp_lds + unit * (kMax * kSpWords);
	SP_DISPATCH(kMax, p.n_subs, sp_scan_unit<4>(p, unit_lds, range, lane), sp_scan_unit<8>(p, unit_lds, range, lane), sp_scan_unit<kMax>(p, unit_lds, range, lane));
}

// ---------------------------------------------------------------------------------------------- ft_sp_threshold
// One workgroup per query: the device half of the 2-phase gate (mergerimpl.h:486-490) and preselectMostRelevantDocs' threshold (:433-446),
// once, for every unit of the kernels behind it.
__global__ __launch_bounds__(256) void ft_sp_threshold(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (!p.prescore) return;
	const bool on = ft_preselect_on(p);
	uint32_t score = 65535u, docs = 0;
	if (on) ft_pick_threshold(p, &score, &docs);
	if (threadIdx.x == 0) {
		uint32_t flags = on ? 1u : 0u;
		if (on) {
			const uint32_t ties = p.hist[score];
			if (ties <= docs) flags |= 2u;   // every document at the threshold score is kept: no order to respect
		}
		p.sync[kFtSyncThrScore] = score;
		p.sync[kFtSyncThrDocs] = docs;
		p.sync[kFtSyncThrFlags] = flags;
	}
	__syncthreads();   // the histogram has been read: it goes back zeroed (one copy, 266 KB, 65 16-byte stores per thread)
	uint4* h4 = reinterpret_cast<uint4*>(p.hist);
	for (uint32_t i = threadIdx.x; i < kFtHistStride / 4; i += 256) h4[i] = make_uint4(0u, 0u, 0u, 0u);
}

// exclusive prefix of `mine` over the units in front of `unit` (decoupled look-back, one word per unit; units start in ticket order, so
// every predecessor is resident)
__device__ inline uint32_t sp_lookback(uint32_t mine, uint32_t unit, unsigned long long* lb, uint32_t* error_flag, int lane) {
	if (lane == 0) __hip_atomic_store(&lb[unit], (unit == 0 ? kLbPrefix : kLbAggregate) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	uint32_t excl = 0;
	long long j = (long long)unit - 1;
	while (j >= 0) {
		const long long idx = j - lane;
		unsigned long long st = 0;
		if (idx >= 0) {
			uint32_t spins = 0;
			do {
				st = __hip_atomic_load(&lb[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (st) break;
				__builtin_amdgcn_s_sleep(1);
				if ((++spins & 1023u) == 0 && (spins > (1u << 24) || __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
					__hip_atomic_store(error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // never hang the GPU: the host reports it
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
	if (lane == 0 && unit != 0) __hip_atomic_store(&lb[unit], kLbPrefix | (unsigned long long)(excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return excl;
}

// The documents of the unit in (word round j, lane) = document order, a sub-term at a time: per merged sub-term the exclusive prefix of its
// bitmap's popcounts (posting indices) and of the documents first met in it (ranks inside (row, range)); `visit` gets the lane's first-met
// documents of every (round, sub-term) that has any: (j, si, words, popcount prefixes, the lane's first-met bits, the rank of its first
// one inside (row, range), documents of the round in front of the lane's, documents of the round).
template <int kS, typename KeptOf, typename Visit>
__device__ __forceinline__ void sp_walk(const SpCtx<kS>& c, int lane, uint32_t (&rows)[kS], KeptOf&& kept_of, Visit&& visit) {
	uint32_t run[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) run[si] = rows[si] = 0;
	for (int j = 0; j < 4; ++j) {
		SpWord<kS> w;
		sp_word<kS>(c, j, lane, w);
		const uint32_t kept = kept_of(j, w);
		uint32_t fm[kS], pre[kS];
		sp_first_met(c, w, kept, fm);
#pragma unroll
		for (int si = 0; si < kS; ++si) {
			pre[si] = 0;
			if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
			const uint32_t cnt_w = uint32_t(__popc(w.W[si]));
			pre[si] = cnt_w | (uint32_t(__popc(fm[si])) << 16);   // two counts (<= 32 a lane, <= 2048 a round) share one scan
		}
		uint32_t first_rank[kS], lanes_front[kS], totals[kS];
#pragma unroll
		for (int si = 0; si < kS; ++si) {
			first_rank[si] = lanes_front[si] = totals[si] = 0;
			if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
			const uint32_t both = pre[si], incl = sp_scan(both), last = sp_readlane(incl, 63);
			pre[si] = run[si] + (incl & 0xFFFFu) - (both & 0xFFFFu);
			run[si] += last & 0xFFFFu;
			lanes_front[si] = (incl >> 16) - (both >> 16);
			first_rank[si] = rows[si] + lanes_front[si];
			totals[si] = last >> 16;
		}
#pragma unroll
		for (int si = 0; si < kS; ++si) {
			if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
			if (totals[si]) visit(j, si, w, pre, fm[si], first_rank[si], lanes_front[si], totals[si]);
			rows[si] += totals[si];   // (uniform: documents of the unit first met in sub-term si so far)
		}
	}
}

// ---------------------------------------------------------------------------------------------- ft_sp_select
// Queries whose 2-phase gate held on the host: which documents preselectMostRelevantDocs keeps (mergerimpl.h:448-462; the ties at the
// threshold score in document order up to minScoreDocs: an ordered count over the units), the table of documents first met per (row,
// range) over those, and one task per kept document.
template <int kS>
__device__ __forceinline__ void sp_select_unit(FtPlanK& p, uint32_t* unit_lds, uint32_t* keptw, uint32_t range, int lane) {
	const SpThreshold thr = sp_threshold(p);
	SpCtx<kS> c;
	sp_open<kS>(p, unit_lds, range, lane, c);
	// pass 1: the unit's documents above / at the threshold score
	uint32_t gt[4], tie[4], ties = 0;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		SpWord<kS> w;
		sp_word<kS>(c, j, lane, w);
		uint32_t g = w.cand, t = 0;   // no preselect: every candidate stays
		if (thr.on) {
			uint32_t claim[kS];
			sp_term_claims(c, w, claim);
			uint32_t rem = w.cand;
			g = 0;
			while (rem) {
				const uint32_t b = uint32_t(__ffs(int(rem)) - 1);
				rem &= rem - 1;
				const uint32_t sc = sp_score(c, claim, b);
				g |= uint32_t(sc > thr.score) << b;
				t |= uint32_t(sc == thr.score) << b;
			}
			if (thr.all_ties) {
				g |= t;
				t = 0;
			}
		}
		gt[j] = g;
		tie[j] = t;
		ties += uint32_t(__popc(t));
	}
	{   // the ties this unit keeps: minScoreDocs minus those of the units in front, in document order
		uint32_t allowed = 0;
		if (thr.on && !thr.all_ties) {
			const uint32_t before = sp_lookback(wave_sum(ties), range, p.lb_units, p.sync + kFtSyncError, lane);
			allowed = thr.docs > before ? thr.docs - before : 0u;
		}
		uint32_t ties_before = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t cnt = uint32_t(__popc(tie[j]));
			const uint32_t incl = sp_scan(cnt);
			const uint32_t front = ties_before + incl - cnt;
			ties_before += sp_readlane(incl, 63);
			keptw[64 * j + lane] = gt[j] | sp_lowest_bits(tie[j], allowed > front ? allowed - front : 0u);
		}
	}
	sp_fence();
	uint32_t kept_total = 0;
#pragma unroll
	for (int j = 0; j < 4; ++j) kept_total += uint32_t(__popc(keptw[64 * j + lane]));
	kept_total = wave_sum(kept_total);
	uint32_t task0 = 0;
	if (lane == 0 && kept_total) task0 = atomicAdd(&p.sync[kFtSyncTasks], kept_total);
	task0 = sp_readlane(task0, 0);
	uint32_t rows[kS];
	if (task0 + kept_total > p.max_merged) {   // (cannot happen: preselectMostRelevantDocs keeps at most maxMergedDocs documents)
		if (lane == 0) p.sync[kFtSyncError] = 1;
#pragma unroll
		for (int si = 0; si < kS; ++si) rows[si] = 0;
		sp_store_rows<kS, false>(p, c, rows, lane);
		return;
	}
	// pass 2: first met per (row, range) over the kept documents, one task each
	uint32_t lo_u[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) lo_u[si] = sp_readlane(c.lo, si);
	uint32_t emitted = 0;
	sp_walk<kS>(
		c, lane, rows, [&](int j, const SpWord<kS>&) { return keptw[64 * j + lane]; },
		[&](int j, int si, const SpWord<kS>& w, const uint32_t (&pre)[kS], uint32_t fm, uint32_t rank0, uint32_t lanes_front, uint32_t total) {
			const uint32_t row = sp_readlane(c.attr, si) >> 20;
			const uint32_t t = task0 + emitted + lanes_front;
			emitted += total;
			uint32_t k = 0;
			while (fm) {
				const uint32_t b = uint32_t(__ffs(int(fm)) - 1);
				fm &= fm - 1;
				sp_emit<kS>(p, c, w, pre, lo_u, t + k, uint32_t(64 * j + lane), b, (row << 24) | (rank0 + k));
				++k;
			}
		});
	sp_store_rows<kS, false>(p, c, rows, lane);
}
template <int kMax>
__global__ __launch_bounds__(256) void ft_sp_select(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (!p.prescore) return;
	extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
	const uint32_t ticket = grab_ticket(p.sync + kFtSyncSpTicket);
	const int lane = threadIdx.x & 63;
	const uint32_t unit = threadIdx.x >> 6, range = ticket * kSpUnits + unit;
	if (range >= p.n_ranges) return;
	uint32_t* unit_lds = sp_lds + unit * ((kMax + 1) * kSpWords);
	uint32_t* keptw = unit_lds + kMax * kSpWords;   // [kSpWords] the kept documents of the unit, word by word
	SP_DISPATCH(kMax, p.n_subs, sp_select_unit<4>(p, unit_lds, keptw, range, lane), sp_select_unit<8>(p, unit_lds, keptw, range, lane),
				sp_select_unit<kMax>(p, unit_lds, keptw, range, lane));
}

// ---------------------------------------------------------------------------------------------- ft_sp_place
// Queries without a preselect (Simple() ones, queries the host found below mergeLimit): the slots are known (ft_slot_bases ran over
// ft_sp_scan's table), so only the documents below maxMergedDocs become tasks, at their slot.
template <int kS>
__device__ __forceinline__ void sp_place_unit(FtPlanK& p, uint32_t* unit_lds, uint32_t range, int lane) {
	// slot of the first document of (row, range): ft_slot_bases' prefix of the table, lane si holds sub-term si's
	uint32_t base_v = 0xFFFFFFFFu;
	if (uint32_t(lane) < p.n_subs) {
		const uint32_t a = p.sp_sub[lane].attr;
		if (!(a & kSpNot)) base_v = p.adders[uint64_t(a >> 20) * p.n_ranges + range];
	}
	uint32_t low = base_v;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const uint32_t o = uint32_t(__shfl_xor(int(low), off, 64));
		low = o < low ? o : low;
	}
	if (low >= p.max_merged) return;   // slots ascend with (row, range): a range whose every row starts at or beyond the limit merges nothing
	SpCtx<kS> c;
	sp_open<kS>(p, unit_lds, range, lane, c);
	uint32_t lo_u[kS], base_u[kS], rows[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) {
		lo_u[si] = sp_readlane(c.lo, si);
		base_u[si] = sp_readlane(base_v, si);
	}
	sp_walk<kS>(
		c, lane, rows, [&](int, const SpWord<kS>& w) { return w.cand; },
		[&](int j, int si, const SpWord<kS>& w, const uint32_t (&pre)[kS], uint32_t fm, uint32_t rank0, uint32_t, uint32_t) {
			const uint32_t slot0 = base_u[si] + rank0;
			uint32_t todo = sp_lowest_bits(fm, slot0 < p.max_merged ? p.max_merged - slot0 : 0u), k = 0;
			while (todo) {
				const uint32_t b = uint32_t(__ffs(int(todo)) - 1);
				todo &= todo - 1;
				sp_emit<kS>(p, c, w, pre, lo_u, slot0 + k, uint32_t(64 * j + lane), b, (slot0 + k) | 0x80000000u);
				++k;
			}
		});
}
template <int kMax>
__global__ __launch_bounds__(256) void ft_sp_place(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (p.prescore) return;
	extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
	const int lane = threadIdx.x & 63;
	const uint32_t unit = threadIdx.x >> 6, range = blockIdx.x * kSpUnits + unit;
	if (range >= p.n_ranges) return;
	uint32_t* unit_lds = sp_lds + unit * (kMax * kSpWords);
	SP_DISPATCH(kMax, p.n_subs, sp_place_unit<4>(p, unit_lds, range, lane), sp_place_unit<8>(p, unit_lds, range, lane), sp_place_unit<kMax>(p, unit_lds, range, lane));
}

// ---------------------------------------------------------------------------------------------- ft_sp_replay
// One thread per merged document: its postings in sub-term order = the order mergeTerm / mergeSimple met them (calcTermRank + the replay of
// ft_replay.hip.h), written at its slot.  The last workgroup writes the result header and hands the synchronisation words back zeroed.
template <int kS>
__global__ __launch_bounds__(256) void ft_sp_replay(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
	__shared__ uint32_t s_last;
	const uint32_t tid = threadIdx.x;
	const uint32_t n_docs = p.sync[kFtSyncNumDocs];   // ft_slot_bases: min(documents first met anywhere, maxMergedDocs)
	if (blockIdx.x * 256 < n_docs) {
		// the query's sub-term (by merge row) and term descriptors: calcTermRank reads a dozen of their fields per posting
		const FtPosSubterm* subs = reinterpret_cast<const FtPosSubterm*>(sp_lds);
		const FtTermCfg* terms = reinterpret_cast<const FtTermCfg*>(sp_lds + kS * (sizeof(FtPosSubterm) / 4));
		{
			const uint32_t sw = uint32_t(sizeof(FtPosSubterm) / 4), tw = uint32_t(sizeof(FtTermCfg) / 4);
			for (uint32_t w = tid; w < p.n_rows * sw; w += 256) {
				const uint32_t r = w / sw;
				sp_lds[w] = reinterpret_cast<const uint32_t*>(p.subs + p.merge_grid[r].sub)[w - r * sw];
			}
			uint32_t* dt = sp_lds + kS * sw;
			const uint32_t* src_t = reinterpret_cast<const uint32_t*>(p.terms);
			for (uint32_t w = tid; w < p.nterms * tw; w += 256) dt[w] = src_t[w];
		}
		__syncthreads();
		const uint32_t t = blockIdx.x * 256 + tid;
		if (t < n_docs) {
			const uint32_t doc = p.t_doc[t], pos = p.t_pos[t];
			uint32_t slot;
			if (pos >> 31) {
				slot = pos & 0x7FFFFFFFu;
			} else {   // ft_sp_select's task: the slot base of its (row, range) is known by now
				slot = p.adders[uint64_t(pos >> 24) * p.n_ranges + (doc >> kFtRangeShift)] + (pos & 0xFFFFFFu);
			}
			uint32_t idx[kS];
			const uint32_t* row = p.t_idx + size_t(t) * p.n_rows;
#pragma unroll
			for (int r = 0; r < kS; ++r) idx[r] = uint32_t(r) < p.n_rows ? row[r] : 0u;
			if (slot < p.max_merged) {
				FtReplayState st;
#pragma unroll
				for (int r = 0; r < kS; ++r) {
					if (!idx[r]) continue;
					const uint32_t i = idx[r] - 1;
					const FtPosSubterm& s = subs[r];
					const FtTermCfg& tc = terms[s.term];
					uint8_t field = 0;
					const float rank = ft_term_rank(tc, s, s.ent_off[i], s.ent_off[i + 1], doc, &field);
					FtPosList pl;
					if (!p.simple) {
						const uint32_t po0 = s.pos_off[i], po1 = s.pos_off[i + 1];
						pl.ptr = s.fpos + po0;
						pl.n = po1 - po0;
					}
					ft_replay_apply(p, st, rank, field, ft_row_qpw(s), pl);
				}
				p.out_doc[slot] = doc;
				ft_replay_finish(p, st, slot, doc);
			}
		}
	}
	// the look-back words of ft_sp_select go back zeroed (every unit of that kernel is over)
	if (p.prescore) {
		for (uint32_t r = blockIdx.x * 256 + tid; r < p.n_ranges; r += gridDim.x * 256) p.lb_units[r] = 0;
	}
	__syncthreads();
	if (tid == 0) s_last = atomicAdd(&p.sync[kFtSyncDoneFinish], 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	if (!s_last) return;
	if (tid == 0) {
		p.out_header[0] = n_docs;
		p.out_header[1] = p.sync[kFtSyncError];
		p.out_header[2] = p.prescore ? (p.sync[kFtSyncThrFlags] & 1u) : 0u;
		p.out_header[3] = 0;
	}
	__syncthreads();
	if (tid < kFtSyncWords) p.sync[tid] = 0;
}

template <int kS>
hipError_t sp_launch(const FtPlan* plans, uint32_t nq, uint32_t n_pre, uint32_t n_ranges, uint32_t t_max, uint32_t m_max, hipStream_t st) {
	constexpr uint32_t kTermsMax = 32;   // ft_sparse_eligible's bound on the query's terms
	if (t_max > kTermsMax) return hipErrorInvalidValue;
	const size_t lds_scan = size_t(kSpUnits) * kS * kSpWords * 4, lds_sel = size_t(kSpUnits) * (kS + 1) * kSpWords * 4;
	const size_t lds_replay = size_t(kS) * sizeof(FtPosSubterm) + size_t(kTermsMax) * sizeof(FtTermCfg);
	static std::atomic<uint64_t> raised_a{0}, raised_b{0}, raised_c{0}, raised_d{0};
	if (hipError_t e = raise_dynamic_lds_once(raised_a, reinterpret_cast<const void*>(&ft_sp_scan<kS>), lds_scan); e != hipSuccess) return e;
	if (hipError_t e = raise_dynamic_lds_once(raised_b, reinterpret_cast<const void*>(&ft_sp_select<kS>), lds_sel); e != hipSuccess) return e;
	if (hipError_t e = raise_dynamic_lds_once(raised_c, reinterpret_cast<const void*>(&ft_sp_place<kS>), lds_scan); e != hipSuccess) return e;
	if (hipError_t e = raise_dynamic_lds_once(raised_d, reinterpret_cast<const void*>(&ft_sp_replay<kS>), lds_replay); e != hipSuccess) return e;
	const uint32_t gx = (n_ranges + kSpUnits - 1) / kSpUnits;
	hipLaunchKernelGGL(ft_sp_scan<kS>, dim3(gx, nq), dim3(256), lds_scan, st, plans);
	if (n_pre) {
		hipLaunchKernelGGL(ft_sp_threshold, dim3(1, n_pre), dim3(256), 0, st, plans);
			hipLaunchKernelGGL(ft_sp_select<kS>, dim3(gx, n_pre), dim3(256), lds_sel, st, plans);
	}
	launch_ft_slot_bases(plans, nq, st);
	if (nq > n_pre) hipLaunchKernelGGL(ft_sp_place<kS>, dim3(gx, nq - n_pre), dim3(256), lds_scan, st, plans + n_pre);
	hipLaunchKernelGGL(ft_sp_replay<kS>, dim3((m_max + 255) / 256, nq), dim3(256), lds_replay, st, plans);
	return hipGetLastError();
}

}  // namespace

// plans: nq plans with sparse = 1 over ONE index, in HBM, those with prescore = 1 first; host_plans: their host copies
hipError_t launch_ft_merge_sparse(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, hipStream_t st) {
	if (!nq) return hipSuccess;
	uint32_t s_max = 1, t_max = 1, m_max = 1, n_pre = 0;
	for (uint32_t q = 0; q < nq; ++q) {
		s_max = std::max(s_max, host_plans[q].n_subs);
		t_max = std::max(t_max, host_plans[q].nterms);
		m_max = std::max(m_max, host_plans[q].max_merged);
		if (host_plans[q].prescore) {
			if (n_pre != q) return hipErrorInvalidValue;   // (the caller sorts them to the front)
			++n_pre;
		}
	}
	const uint32_t n_ranges = host_plans[0].n_ranges;
	if (s_max <= 4) return sp_launch<4>(plans, nq, n_pre, n_ranges, t_max, m_max, st);
	if (s_max <= 8) return sp_launch<8>(plans, nq, n_pre, n_ranges, t_max, m_max, st);
	if (s_max <= kFtSparseSubs) return sp_launch<16>(plans, nq, n_pre, n_ranges, t_max, m_max, st);
	return hipErrorInvalidValue;
}

}  // namespace rxgpu
