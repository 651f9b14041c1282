// ft_fast merge on gfx950, the train for SPARSELY hit document ranges (Merger::Merge, cpp_src/core/ft/ft_fast/mergerimpl.h:466-566, for the
// queries ft_sparse_eligible() in rxgpu_ft_capi.hip admits: plain terms whose every field has the same positive boost, at most kFtSparseSubs
// sub-terms, postings on a fraction of the documents).
//
// The dense train (ft_merge.hip) gives every document range of 8192 documents a 256-thread workgroup and keeps per-DOCUMENT arrays in HBM
// between its kernels (16-bit pre-scores, the restricting mask, 16-byte records of every ranked posting).  A two-term query over 5M
// documents touches 8 % of them: 650 postings per range, for which a workgroup walks ~20 us of dependent phases and the arrays move 4 x the
// bytes of the postings (profiles/rd5_bm25_train64_sparse_rocprof.json).  Here the unit of work is ONE WAVEFRONT per (query, range), four to a
// workgroup, no workgroup barrier inside a unit, and NOTHING per document or per posting is written to HBM:
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
//  * three global facts order the kernels: the pre-score histogram -> threshold (ft_sp_scan | ft_sp_threshold), the ties kept at the threshold
//    in document order and the table of documents first met per (row, range) (ft_sp_select | ft_slot_bases), the slots (ft_sp_finish).
//    Each unit kernel rebuilds its bitmaps from the postings (4 B per posting from L2 / HBM) instead of reading back what another wrote.
//  * calcTermRank + the per-document replay (mergeTerm :107-192 / mergeSimple :194-250) run LAST, in ft_sp_finish, only for documents whose
//    slot lies below maxMergedDocs: a whole-corpus single-term merge ranks its 20 000 merged documents, not its 600 000 postings.
//
// Launch train: ft_sp_scan, [ft_sp_threshold, ft_sp_select — queries that may preselect], ft_slot_bases (ft_merge.hip), ft_sp_finish.
// Results are the dense train's to the bit (same float operations per document, same slots): tests/test_gpu_ft_*.py run both.
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
constexpr uint32_t kSpRing = 128;                  // documents waiting for a replay lane (ft_sp_finish)
constexpr uint32_t kSpLoads = 8;                   // posting loads a lane keeps in flight while the bitmaps are built
static_assert(kSpWords == 4 * 64, "a lane owns four words of every bitmap");

// LDS of one workgroup (32-bit words): the query's sub-term and term descriptors (all four units serve the same query), one attribute
// word per sub-term, then per unit the bitmaps, the segment starts and — ft_sp_finish — the popcount prefixes and the replay ring.
struct SpLayout {
	uint32_t subs, terms, attr, meta, unit0, unit_stride;   // word offsets
	uint32_t u_bits, u_lo, u_prefix, u_ring;                // inside a unit
	uint32_t total_words;
};
__host__ __device__ inline SpLayout sp_layout(uint32_t ks, uint32_t t_max, bool finish) {
	SpLayout l{};
	uint32_t o = 0;
	l.subs = o;
	o += ks * uint32_t(sizeof(FtPosSubterm) / 4);
	l.terms = o;
	o += t_max * uint32_t(sizeof(FtTermCfg) / 4);
	l.attr = o;
	o += kFtSparseSubs;
	l.meta = o;
	o += 4;
	o = (o + 3u) & ~3u;
	l.unit0 = o;
	uint32_t u = 0;
	l.u_bits = u;
	u += ks * kSpWords;
	l.u_lo = u;
	u += kFtSparseSubs;
	if (finish) {
		l.u_prefix = u;
		u += ks * kSpWords / 2;   // 16 bits per word
		l.u_ring = u;
		u += 2 * kSpRing;
	}
	u = (u + 3u) & ~3u;
	l.unit_stride = u;
	l.total_words = o + kSpUnits * u;
	return l;
}
static_assert(sizeof(FtPosSubterm) % 8 == 0 && sizeof(FtTermCfg) % 8 == 0, "descriptors are copied word by word and hold 8-byte members");

// sub-term attribute word: proc16 | first sub-term of its term << 16 | AND term << 17 | NOT term << 18 | merge row << 20
constexpr uint32_t kSpFirst = 1u << 16, kSpAnd = 1u << 17, kSpNot = 1u << 18;

// LDS traffic between the lanes of ONE wavefront: the hardware executes a wavefront's LDS instructions in order, the fence keeps the compiler
// from moving them across
__device__ __forceinline__ void sp_fence() {
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t sp_readlane(uint32_t v, int lane) { return uint32_t(__builtin_amdgcn_readlane(int(v), lane)); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const uint32_t o = uint32_t(__shfl_xor(int(v), off, 64));
		v = o < v ? o : v;
	}
	return v;
}
__device__ __forceinline__ uint32_t sp_lanes_below(unsigned long long m, int lane) { return uint32_t(__popcll(m & ((1ull << lane) - 1ull))); }

// What every unit kernel knows about its query (uniform over the wavefront) and its range
template <int kS>
struct SpCtx {
	const FtPosSubterm* subs;   // LDS copies
	const FtTermCfg* terms;
	uint32_t* bits;             // LDS [kS][kSpWords]
	uint32_t* lo;               // LDS [kFtSparseSubs]: first posting of the range in sub-term si
	uint32_t n_subs;
	uint32_t attr;              // lane si: attribute word of sub-term si
	unsigned long long first_m, and_m, not_m;
	uint32_t p16[kS];           // proc16 of sub-term si (uniform)
	bool empty_and;             // an AND term without postings: no document passes (buildRestrictingBitmask)
	uint32_t range, d_begin, docs_here;
	uint32_t rm[4], ex[4];      // removed / excluded bits of the lane's four words
};

// The workgroup's share: descriptors into LDS, attribute words.  Ends with a workgroup barrier (the only one in front of the units).
template <int kS>
__device__ __forceinline__ void sp_setup(FtPlanK& p, uint32_t* lds, const SpLayout& L) {
	const uint32_t tid = threadIdx.x;
	uint32_t* d_subs = lds + L.subs;
	uint32_t* d_terms = lds + L.terms;
	const uint32_t ns_words = p.n_subs * uint32_t(sizeof(FtPosSubterm) / 4), nt_words = p.nterms * uint32_t(sizeof(FtTermCfg) / 4);
	const uint32_t* src_s = reinterpret_cast<const uint32_t*>(p.subs);
	const uint32_t* src_t = reinterpret_cast<const uint32_t*>(p.terms);
	for (uint32_t w = tid; w < ns_words; w += 256) d_subs[w] = src_s[w];
	for (uint32_t w = tid; w < nt_words; w += 256) d_terms[w] = src_t[w];
	if (tid < 4) lds[L.meta + tid] = 0;
	__syncthreads();
	const FtPosSubterm* subs = reinterpret_cast<const FtPosSubterm*>(d_subs);
	const FtTermCfg* terms = reinterpret_cast<const FtTermCfg*>(d_terms);
	if (tid < p.n_subs) {
		const FtPosSubterm& s = subs[tid];
		const FtTermCfg& t = terms[s.term];
		// calcTermScores (mergerimpl.h:312-315): every field has the same boost, so maxBoostFromFields is field 0's
		const float proc = s.proc * t.field_boost[0] * t.opts_boost;
		uint32_t p16 = uint32_t(int32_t(proc)) & 0xFFFFu;   // static_cast<uint16_t>(float) as x86 evaluates it
		p16 = p16 < 65535u / 4 ? p16 : 65535u / 4;
		uint32_t a = p16 | (s.row << 20);
		if (tid == t.sub_begin) a |= kSpFirst;
		if (t.op == 2) a |= kSpAnd;
		if (t.op == 3) a |= kSpNot;
		lds[L.attr + tid] = a;
	}
	if (tid < p.nterms && !p.simple) {
		const FtTermCfg& t = terms[tid];
		if (t.op == 2 && t.sub_begin == t.sub_end) lds[L.meta] = 1;
	}
	__syncthreads();
}

template <int kS>
__device__ __forceinline__ void sp_unit_ctx(FtPlanK& p, uint32_t* lds, const SpLayout& L, uint32_t unit_in_wg, uint32_t range, int lane, SpCtx<kS>& c) {
	uint32_t* ub = lds + L.unit0 + unit_in_wg * L.unit_stride;
	c.subs = reinterpret_cast<const FtPosSubterm*>(lds + L.subs);
	c.terms = reinterpret_cast<const FtTermCfg*>(lds + L.terms);
	c.bits = ub + L.u_bits;
	c.lo = ub + L.u_lo;
	c.n_subs = p.n_subs;
	c.attr = uint32_t(lane) < p.n_subs ? lds[L.attr + lane] : 0u;
	c.first_m = __ballot((c.attr & kSpFirst) != 0);
	c.and_m = __ballot((c.attr & kSpAnd) != 0);
	c.not_m = __ballot((c.attr & kSpNot) != 0);
#pragma unroll
	for (int si = 0; si < kS; ++si) c.p16[si] = sp_readlane(c.attr, si) & 0xFFFFu;
	c.empty_and = lds[L.meta] != 0;
	c.range = range;
	c.d_begin = range << kFtRangeShift;
	const uint64_t left = p.total_docs - uint64_t(c.d_begin);
	c.docs_here = uint32_t(left < kFtRangeDocs ? left : kFtRangeDocs);
#pragma unroll
	for (int j = 0; j < 4; ++j) {   // requested now, consumed behind the bitmaps
		const uint64_t gw = uint64_t(c.d_begin) / 32 + uint32_t(64 * j + lane);
		const bool in = gw < p.nwords;
		c.rm[j] = (p.removed_bits && in) ? p.removed_bits[gw] : 0u;
		c.ex[j] = (p.excluded_bits && in) ? p.excluded_bits[gw] : 0u;
	}
}

// One bit per (sub-term, document of the range).  The postings of all sub-terms form one flat index space; a lane keeps kSpLoads document
// loads in flight.
template <int kS>
__device__ __forceinline__ void sp_build(const SpCtx<kS>& c, int lane) {
	uint32_t lo = 0, hi = 0;
	if (uint32_t(lane) < c.n_subs) {
		const FtPosSubterm& s = c.subs[lane];
		lo = c.range < s.n_ranges ? s.range_off[c.range] : uint32_t(s.n);
		hi = c.range + 1 < s.n_ranges ? s.range_off[c.range + 1] : uint32_t(s.n);
		c.lo[lane] = lo;
	}
	uint4* b4 = reinterpret_cast<uint4*>(c.bits);
	for (uint32_t k = uint32_t(lane); k < kS * kSpWords / 4; k += 64) b4[k] = make_uint4(0u, 0u, 0u, 0u);
	const uint32_t len = hi - lo;
	const uint32_t incl = wave_inclusive_scan(len, lane);
	const uint32_t cum = incl - len;                     // lane si: postings of the range in front of sub-term si
	const uint32_t total = sp_readlane(incl, 63);
	uint32_t cum_u[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) cum_u[si] = sp_readlane(cum, si);   // (lanes past the last sub-term: total)
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
			d[k] = 0;
			if (ok[k]) d[k] = c.subs[s].doc[c.lo[s] + (f - cs)];
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

// The distinct pre-scores of a unit with their document counts, one (score, count) per lane; more than 64 distinct scores go straight to HBM
struct SpKeys {
	uint32_t key = 0, cnt = 0;
	uint32_t n = 0;   // uniform
};
template <typename Overflow>
__device__ __forceinline__ void sp_keys_add(SpKeys& t, bool have, uint32_t sc, int lane, Overflow&& overflow) {
	unsigned long long pending = __ballot(have);
	while (pending) {
		const int leader = __ffsll((long long)pending) - 1;
		const uint32_t v = uint32_t(__shfl(int(sc), leader, 64));
		const unsigned long long same = __ballot(have && sc == v);
		const uint32_t c = uint32_t(__popcll(same));
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

// the documents of the lane's word above / at the threshold score
template <int kS>
__device__ __forceinline__ void sp_gt_tie(const SpCtx<kS>& c, const SpWord<kS>& w, uint32_t thr, uint32_t& gt, uint32_t& tie) {
	uint32_t claim[kS];
	sp_term_claims(c, w, claim);
	gt = tie = 0;
	uint32_t rem = w.cand;
	while (rem) {
		const uint32_t b = uint32_t(__ffs(int(rem)) - 1);
		rem &= rem - 1;
		const uint32_t sc = sp_score(c, claim, b);
		gt |= uint32_t(sc > thr) << b;
		tie |= uint32_t(sc == thr) << b;
	}
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
// the documents of word j the merge may add: everything (no preselect), or above the threshold + the first `allowed` ties of the unit
template <int kS>
__device__ __forceinline__ uint32_t sp_kept_word(const SpCtx<kS>& c, const SpWord<kS>& w, const SpThreshold& thr, uint32_t allowed, uint32_t& ties_before, int lane) {
	if (!thr.on) return w.cand;
	uint32_t gt, tie;
	sp_gt_tie(c, w, thr.score, gt, tie);
	if (thr.all_ties) return gt | tie;
	const uint32_t cnt = uint32_t(__popc(tie));
	const uint32_t incl = wave_inclusive_scan(cnt, lane);
	const uint32_t before = ties_before + incl - cnt;   // ties of the unit in front of this word, in document order
	ties_before += sp_readlane(incl, 63);
	const uint32_t room = allowed > before ? allowed - before : 0u;
	return gt | sp_lowest_bits(tie, room);
}

// ---------------------------------------------------------------------------------------------- ft_sp_scan
// prescore: the restricting mask's popcount and the pre-score histogram of the unit's documents (the input of the 2-phase gate and of the
// threshold).  Otherwise (Simple() queries, queries below mergeLimit): the documents first met per (row, range) — ft_adders' table — at once.
template <int kS>
__global__ __launch_bounds__(256) void ft_sp_scan(const FtPlan* plans, uint32_t t_max) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
	const SpLayout L = sp_layout(kS, t_max, false);
	sp_setup<kS>(p, sp_lds, L);
	const int lane = threadIdx.x & 63;
	const uint32_t unit = threadIdx.x >> 6, range = blockIdx.x * kSpUnits + unit;
	if (range >= p.n_ranges) return;
	SpCtx<kS> c;
	sp_unit_ctx<kS>(p, sp_lds, L, unit, range, lane, c);
	sp_build<kS>(c, lane);
	uint32_t pop = 0;
	uint32_t rows[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) rows[si] = 0;
	SpKeys keys;
	uint32_t* hist_copy = p.prescore ? p.hist + size_t(range % kFtHistCopies) * kFtHistStride : nullptr;
	auto overflow = [&](uint32_t v, uint32_t n) {
		atomicAdd(&hist_copy[v], n);
		atomicAdd(&hist_copy[65536 + (v >> 6)], n);
	};
	for (int j = 0; j < 4; ++j) {
		SpWord<kS> w;
		sp_word<kS>(c, j, lane, w);
		if (p.prescore) {
			pop += uint32_t(__popc(w.mask));
			uint32_t claim[kS];
			sp_term_claims(c, w, claim);
			uint32_t rem = w.cand;
			while (__ballot(rem != 0)) {
				const bool have = rem != 0;
				uint32_t sc = 0;
				if (have) {
					const uint32_t b = uint32_t(__ffs(int(rem)) - 1);
					rem &= rem - 1;
					sc = sp_score(c, claim, b);
				}
				sp_keys_add(keys, have && sc != 0, sc, lane, overflow);   // (score 0 is not counted: mergerimpl.h:433 walks scores >= 1)
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
		if (uint32_t(lane) < keys.n) overflow(keys.key, keys.cnt);
	} else {
#pragma unroll
		for (int si = 0; si < kS; ++si) {
			if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
			const uint32_t tot = wave_sum(rows[si]);
			if (lane == 0) p.adders[uint64_t(sp_readlane(c.attr, si) >> 20) * p.n_ranges + range] = tot;
		}
	}
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
			uint32_t ties = 0;
			for (uint32_t k = 0; k < kFtHistCopies; ++k) ties += p.hist[size_t(k) * kFtHistStride + score];
			if (ties <= docs) flags |= 2u;   // every document at the threshold score is kept: no order to respect
		}
		p.sync[kFtSyncThrScore] = score;
		p.sync[kFtSyncThrDocs] = docs;
		p.sync[kFtSyncThrFlags] = flags;
	}
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

// ---------------------------------------------------------------------------------------------- ft_sp_select
// Queries whose 2-phase gate held on the host: which documents preselectMostRelevantDocs keeps (mergerimpl.h:448-462; the ties at the
// threshold score in document order up to minScoreDocs: an ordered count over the units) and, over those, the table of documents first met
// per (row, range).  Also hands the pre-score histogram back zeroed: every unit clears the counters of its own scores.
template <int kS>
__global__ __launch_bounds__(256) void ft_sp_select(const FtPlan* plans, uint32_t t_max) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (!p.prescore) return;
	extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
	const SpLayout L = sp_layout(kS, t_max, false);
	const uint32_t ticket = grab_ticket(p.sync + kFtSyncSpTicket);
	sp_setup<kS>(p, sp_lds, L);
	const int lane = threadIdx.x & 63;
	const uint32_t unit = threadIdx.x >> 6, range = ticket * kSpUnits + unit;
	if (range >= p.n_ranges) return;
	const SpThreshold thr = sp_threshold(p);
	SpCtx<kS> c;
	sp_unit_ctx<kS>(p, sp_lds, L, unit, range, lane, c);
	sp_build<kS>(c, lane);
	// pass 1: the unit's ties at the threshold; the distinct scores of its documents (their histogram counters are cleared)
	uint32_t* hist_copy = p.hist + size_t(range % kFtHistCopies) * kFtHistStride;
	auto clear = [&](uint32_t v, uint32_t) {
		hist_copy[v] = 0;
		hist_copy[65536 + (v >> 6)] = 0;
	};
	SpKeys keys;
	uint32_t ties = 0;
	for (int j = 0; j < 4; ++j) {
		SpWord<kS> w;
		sp_word<kS>(c, j, lane, w);
		uint32_t claim[kS];
		sp_term_claims(c, w, claim);
		uint32_t rem = w.cand;
		while (__ballot(rem != 0)) {
			const bool have = rem != 0;
			uint32_t sc = 0;
			if (have) {
				const uint32_t b = uint32_t(__ffs(int(rem)) - 1);
				rem &= rem - 1;
				sc = sp_score(c, claim, b);
			}
			ties += (have && thr.on && sc == thr.score) ? 1u : 0u;
			sp_keys_add(keys, have && sc != 0, sc, lane, clear);
		}
	}
	if (uint32_t(lane) < keys.n) clear(keys.key, keys.cnt);
	uint32_t allowed = 0xFFFFFFFFu;
	if (thr.on && !thr.all_ties) {
		ties = wave_sum(ties);
		const uint32_t before = sp_lookback(ties, range, p.lb_units, p.sync + kFtSyncError, lane);
		allowed = thr.docs > before ? thr.docs - before : 0u;
		if (lane == 0) p.unit_allow[range] = allowed;
	}
	// pass 2: first met per (row, range) over the kept documents
	uint32_t rows[kS];
#pragma unroll
	for (int si = 0; si < kS; ++si) rows[si] = 0;
	uint32_t ties_before = 0;
	for (int j = 0; j < 4; ++j) {
		SpWord<kS> w;
		sp_word<kS>(c, j, lane, w);
		const uint32_t kept = sp_kept_word(c, w, thr, allowed, ties_before, lane);
		uint32_t fm[kS];
		sp_first_met(c, w, kept, fm);
#pragma unroll
		for (int si = 0; si < kS; ++si) rows[si] += uint32_t(__popc(fm[si]));
	}
#pragma unroll
	for (int si = 0; si < kS; ++si) {
		if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
		const uint32_t tot = wave_sum(rows[si]);
		if (lane == 0) p.adders[uint64_t(sp_readlane(c.attr, si) >> 20) * p.n_ranges + range] = tot;
	}
}

// ---------------------------------------------------------------------------------------------- ft_sp_finish
// One document per lane: its postings in sub-term order = the order mergeTerm / mergeSimple met them (calcTermRank + replay), written at its slot
template <int kS>
__device__ __forceinline__ void sp_replay_chunk(FtPlanK& p, const SpCtx<kS>& c, const uint16_t* prefix, bool have, uint32_t dl, uint32_t slot) {
	uint32_t hits = 0;   // bit si: merged sub-term si holds the document
	const uint32_t word = dl >> 5, bit = dl & 31u;
	if (have) {
#pragma unroll
		for (int si = 0; si < kS; ++si) {
			if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
			hits |= ((c.bits[uint32_t(si) * kSpWords + word] >> bit) & 1u) << si;
		}
	}
	const uint32_t doc = c.d_begin + dl;
	FtReplayState st;
	while (hits) {
		const uint32_t si = uint32_t(__ffs(int(hits)) - 1);
		hits &= hits - 1;
		const FtPosSubterm& s = c.subs[si];
		const FtTermCfg& t = c.terms[s.term];
		const uint32_t wbits = c.bits[si * kSpWords + word];
		const uint32_t i = c.lo[si] + prefix[si * kSpWords + word] + uint32_t(__popc(wbits & ((1u << bit) - 1u)));
		uint8_t field = 0;
		const float rank = ft_term_rank(t, s, s.ent_off[i], s.ent_off[i + 1], doc, &field);
		FtPosList pos;
		if (!p.simple) {
			const uint32_t po0 = s.pos_off[i], po1 = s.pos_off[i + 1];
			pos.ptr = s.fpos + po0;
			pos.n = po1 - po0;
		}
		ft_replay_apply(p, st, rank, field, ft_row_qpw(s), pos);
	}
	if (have) {
		p.out_doc[slot] = doc;
		ft_replay_finish(p, st, slot, doc);
	}
}

template <int kS>
__global__ __launch_bounds__(256) void ft_sp_finish(const FtPlan* plans, uint32_t t_max) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
	__shared__ uint32_t s_last;
	const SpLayout L = sp_layout(kS, t_max, true);
	sp_setup<kS>(p, sp_lds, L);
	const int lane = threadIdx.x & 63;
	const uint32_t unit = threadIdx.x >> 6, range = blockIdx.x * kSpUnits + unit;
	if (range < p.n_ranges) {
		const SpThreshold thr = sp_threshold(p);
		SpCtx<kS> c;
		sp_unit_ctx<kS>(p, sp_lds, L, unit, range, lane, c);
		// slot of the first document of (row, range): ft_slot_bases' prefix of the table, lane si holds sub-term si's
		uint32_t base_v = 0xFFFFFFFFu;
		if (uint32_t(lane) < c.n_subs && !((c.not_m >> lane) & 1ull)) base_v = p.adders[uint64_t(c.attr >> 20) * p.n_ranges + range];
		uint32_t allowed = 0xFFFFFFFFu;
		if (thr.on && !thr.all_ties) allowed = p.unit_allow[range];
		if (lane == 0 && p.lb_units) p.lb_units[range] = 0;   // (ft_sp_select is over: the look-back word goes back zeroed)
		const uint32_t low = wave_min_u32(base_v);
		if (low < p.max_merged) {   // slots ascend with (row, range): a range whose every row starts at or beyond the limit merges nothing
			sp_build<kS>(c, lane);
			uint32_t* ub = sp_lds + L.unit0 + unit * L.unit_stride;
			uint16_t* prefix = reinterpret_cast<uint16_t*>(ub + L.u_prefix);   // [kS][kSpWords]: postings of the range in front of a word's
			uint32_t* ring = ub + L.u_ring;                                     // [kSpRing] document in the range, [kSpRing] slot
			{   // popcount prefix along every merged sub-term's bitmap: the posting index of a document = segment start + bits in front
				uint32_t run[kS];
#pragma unroll
				for (int si = 0; si < kS; ++si) run[si] = 0;
				for (int j = 0; j < 4; ++j) {
#pragma unroll
					for (int si = 0; si < kS; ++si) {
						if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
						const uint32_t cnt = uint32_t(__popc(c.bits[uint32_t(si) * kSpWords + uint32_t(64 * j + lane)]));
						const uint32_t incl = wave_inclusive_scan(cnt, lane);
						prefix[uint32_t(si) * kSpWords + uint32_t(64 * j + lane)] = uint16_t(run[si] + incl - cnt);
						run[si] += sp_readlane(incl, 63);
					}
				}
			}
			sp_fence();
			uint32_t base_u[kS], met[kS];
#pragma unroll
			for (int si = 0; si < kS; ++si) {
				base_u[si] = sp_readlane(base_v, si);
				met[si] = 0;   // documents of the unit first met in row si so far
			}
			uint32_t head = 0, count = 0, ties_before = 0;
			auto drain = [&](bool all) {
				while (count >= 64 || (all && count)) {
					const bool have = uint32_t(lane) < count;
					const uint32_t at = (head + uint32_t(lane)) & (kSpRing - 1);
					sp_replay_chunk<kS>(p, c, prefix, have, have ? ring[at] : 0u, have ? ring[kSpRing + at] : 0u);
					const uint32_t took = count < 64 ? count : 64;
					head = (head + took) & (kSpRing - 1);
					count -= took;
				}
			};
			for (int j = 0; j < 4; ++j) {
				SpWord<kS> w;
				sp_word<kS>(c, j, lane, w);
				const uint32_t kept = sp_kept_word(c, w, thr, allowed, ties_before, lane);
				uint32_t fm[kS];
				sp_first_met(c, w, kept, fm);
#pragma unroll
				for (int si = 0; si < kS; ++si) {
					if (uint32_t(si) >= c.n_subs || ((c.not_m >> si) & 1ull)) continue;
					if (base_u[si] + met[si] >= p.max_merged) continue;   // the rest of this row lies beyond the limit
					const uint32_t cnt = uint32_t(__popc(fm[si]));
					const uint32_t incl = wave_inclusive_scan(cnt, lane);
					const uint32_t first_slot = base_u[si] + met[si] + incl - cnt;
					met[si] += sp_readlane(incl, 63);
					uint32_t room = first_slot < p.max_merged ? p.max_merged - first_slot : 0u;
					uint32_t todo = sp_lowest_bits(fm[si], room), k = 0;
					while (__ballot(todo != 0)) {   // one document per lane and pass into the ring
						const bool have = todo != 0;
						const unsigned long long m = __ballot(have);
						if (have) {
							const uint32_t b = uint32_t(__ffs(int(todo)) - 1);
							todo &= todo - 1;
							const uint32_t at = (head + count + sp_lanes_below(m, lane)) & (kSpRing - 1);
							ring[at] = uint32_t(64 * j + lane) * 32 + b;
							ring[kSpRing + at] = first_slot + k;
							++k;
						}
						count += uint32_t(__popcll(m));
						sp_fence();
						drain(false);
					}
				}
			}
			sp_fence();
			drain(true);
		}
	}
	// ---- the last workgroup writes the result header and hands the synchronisation words back zeroed
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(&p.sync[kFtSyncDoneFinish], 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	if (!s_last) return;
	if (threadIdx.x == 0) {
		p.out_header[0] = p.sync[kFtSyncNumDocs];
		p.out_header[1] = p.sync[kFtSyncError];
		p.out_header[2] = p.prescore ? (p.sync[kFtSyncThrFlags] & 1u) : 0u;
		p.out_header[3] = 0;
	}
	__syncthreads();
	if (threadIdx.x < kFtSyncWords) p.sync[threadIdx.x] = 0;
}

template <int kS>
hipError_t sp_launch(const FtPlan* plans, uint32_t nq, uint32_t n_ranges, uint32_t t_max, bool any_pre, hipStream_t st) {
	const size_t lds_scan = size_t(sp_layout(kS, t_max, false).total_words) * 4, lds_fin = size_t(sp_layout(kS, t_max, true).total_words) * 4;
	static std::atomic<uint64_t> raised_a{0}, raised_b{0}, raised_c{0};
	constexpr uint32_t kTermsMax = 32;   // ft_sparse_eligible's bound on the query's terms
	if (t_max > kTermsMax) return hipErrorInvalidValue;
	const size_t max_scan = size_t(sp_layout(kS, kTermsMax, false).total_words) * 4, max_fin = size_t(sp_layout(kS, kTermsMax, true).total_words) * 4;
	if (hipError_t e = raise_dynamic_lds_once(raised_a, reinterpret_cast<const void*>(&ft_sp_scan<kS>), max_scan); e != hipSuccess) return e;
	if (hipError_t e = raise_dynamic_lds_once(raised_b, reinterpret_cast<const void*>(&ft_sp_select<kS>), max_scan); e != hipSuccess) return e;
	if (hipError_t e = raise_dynamic_lds_once(raised_c, reinterpret_cast<const void*>(&ft_sp_finish<kS>), max_fin); e != hipSuccess) return e;
	const dim3 grid((n_ranges + kSpUnits - 1) / kSpUnits, nq);
	hipLaunchKernelGGL(ft_sp_scan<kS>, grid, dim3(256), lds_scan, st, plans, t_max);
	if (any_pre) {
		hipLaunchKernelGGL(ft_sp_threshold, dim3(1, nq), dim3(256), 0, st, plans);
		hipLaunchKernelGGL(ft_sp_select<kS>, grid, dim3(256), lds_scan, st, plans, t_max);
	}
	launch_ft_slot_bases(plans, nq, st);
	hipLaunchKernelGGL(ft_sp_finish<kS>, grid, dim3(256), lds_fin, st, plans, t_max);
	return hipGetLastError();
}

}  // namespace

// plans: nq plans with sparse = 1 over ONE index, in HBM; host_plans: their host copies
hipError_t launch_ft_merge_sparse(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, hipStream_t st) {
	if (!nq) return hipSuccess;
	uint32_t s_max = 1, t_max = 1;
	bool any_pre = false;
	for (uint32_t q = 0; q < nq; ++q) {
		s_max = std::max(s_max, host_plans[q].n_subs);
		t_max = std::max(t_max, host_plans[q].nterms);
		any_pre = any_pre || host_plans[q].prescore;
	}
	const uint32_t n_ranges = host_plans[0].n_ranges;
	if (s_max <= 4) return sp_launch<4>(plans, nq, n_ranges, t_max, any_pre, st);
	if (s_max <= 8) return sp_launch<8>(plans, nq, n_ranges, t_max, any_pre, st);
	if (s_max <= kFtSparseSubs) return sp_launch<16>(plans, nq, n_ranges, t_max, any_pre, st);
	return hipErrorInvalidValue;
}

}  // namespace rxgpu
