// ft_fast merge on gfx950: Merger::Merge (cpp_src/core/ft/ft_fast/mergerimpl.h:466-566) for queries made of terms — mergeSimple (:194-250)
// for a Simple() query, buildRestrictingBitmask (:326-384) + the 2-phase gate / preselectMostRelevantDocs (:386-464, 486-490) + mergeTerm
// (:107-192) otherwise.  Phrases and multi-word synonyms stay on the CPU merger.
//
// The reference walks (term, sub-term, posting) sequentially and is order dependent in three places.  Round 1 reproduced that order with
// one launch per sub-term (9 + 9 launches and as many fills for a 3 x 3 query: launch-latency bound, 5-11 % of HBM).  Here the same
// RESULT is derived order-free, so a whole query is a fixed number of launches, each over ALL postings of the query:
//
//  * restricting bitmask and pre-score (buildRestrictingBitmask :326-384, calcTermScores :289-324): both are per-DOCUMENT facts (is the
//    document in every AND term / in no NOT term; per term the FIRST sub-term holding it contributes its proc16, saturating sum over the
//    terms).  Posting lists are sorted by document, so a workgroup that owns a RANGE of 8192 documents finds its segment of every list in
//    a per-word range index and resolves everything in LDS — bit arrays for the masks, a 16-bit score per document, sub-terms in order
//    behind workgroup barriers — without one global atomic  ->  ft_ranges.  (The first cut scattered postings into per-term arrays with
//    device-scope atomics: 140 us for a 3 x 3 query; atomics on random addresses resolve past the per-XCD L2s at ~28 G/s.)
//  * admission (addDoc until maxMergedDocs, merger.h:161-180): a document is added by its first posting (in global posting order) that is
//    eligible and has a non-zero rank; it gets the next merge slot if fewer than maxMergedDocs documents were added before it, and once the
//    limit is hit nothing is added any more.  Global posting order is (sub-term row, document) — every list ascends by document — so the
//    slot of a document is  #documents first met in an earlier row  +  #documents first met in the same row with a smaller id.
//    ft_rank_all ranks every eligible posting (calcTermRank) and drops the survivors (rank != 0; ~1 % after a preselect) into per-document-
//    range buckets; ft_adders, one workgroup per range of 8192 documents, finds every document's first row (16-bit minimum in LDS) and
//    counts them per (row, range); the last workgroup turns that small table into its exclusive prefix in row-major order = the slot bases.
//    (The first cuts ran this posting-side: an atomicMin table over all documents plus three passes over ALL postings — count, ordered
//    prefix, scatter — 45 us of a 128 us merge, each pass bound by its dependent gathers, not by bytes.)
//  * per-document state (`proc -= rank; proc += finalRank` on every strict improvement, switchToNextWord between terms, termsCounter):
//    a document meets at most one posting per sub-term and all its postings sit in ONE bucket, so ft_finish — again one workgroup per
//    range — sorts the range's first postings by (row, document) in LDS, adds the slot bases, drops the range's survivors into a per-slot
//    row indexed by sub-term and, behind a workgroup barrier, replays each of its documents in sub-term order with the reference's float
//    operations  ->  same bits.
//  * preselect ties at the threshold score are kept in document order: ordered prefix (ft_preselect_apply).
//
// Launch train of a multi-term query: ft_ranges, [ft_preselect_pick, ft_preselect_apply], ft_rank_all, ft_adders, ft_finish; the 2-phase
// gate's popcount test is evaluated ON THE DEVICE (no host round trip), the result leaves in one packed buffer.  A Simple() query:
// ft_ranges (mask only), ft_rank_all, ft_adders, ft_finish.  No fill kernel: the tables a merge reads before it writes (histogram,
// bucket counters, look-back and synchronisation words) are handed back ZEROED by the merge that used them (ft_adders / ft_finish).
//
// Bound: HBM gathers: per posting 4 B doc + 8 B entry offsets + 9 B per (field, tf, firstPos) entry streamed, 4 B
// words-in-field + the mask word gathered; 16 B per surviving posting written and read back.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "rxgpu_internal.h"
#include "knn_kernels.hip.h"
#include "ft_rank.hip.h"

namespace rxgpu {

namespace {

constexpr unsigned long long kLbPrefix = 1ull << 63;
constexpr unsigned long long kLbAggregate = 1ull << 62;
constexpr int kFtApplyWords = 4;   // mask words per thread in ft_preselect_apply

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
__device__ uint32_t ordered_prefix(uint32_t count, uint32_t ticket, unsigned long long* lookback, uint32_t* error_flag, uint32_t* grand_incl) {
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

// The doc ids of a thread's kFtPassItems consecutive postings (one 16-byte load away from the tail)
__device__ __forceinline__ void load_docs(const FtPosSubterm& s, uint64_t i0, uint32_t (&docs)[kFtPassItems], bool (&live)[kFtPassItems]) {
	static_assert(kFtPassItems == 4, "the vector load reads four document ids");
	if (i0 + kFtPassItems <= s.n) {
		const uint4 v = *reinterpret_cast<const uint4*>(s.doc + i0);   // i0 % 4 == 0 and the list is 256-byte aligned
		docs[0] = v.x;
		docs[1] = v.y;
		docs[2] = v.z;
		docs[3] = v.w;
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) live[k] = true;
	} else {
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) {
			live[k] = i0 + k < s.n;
			docs[k] = live[k] ? s.doc[i0 + k] : 0u;
		}
	}
}

__device__ __forceinline__ void fill_words(uint32_t* ptr, uint64_t n, uint32_t value, uint64_t gtid, uint64_t gsize) {
	if (!ptr || !n) return;
	const uint64_t n4 = n / 4;
	uint4* p4 = reinterpret_cast<uint4*>(ptr);   // every scratch region starts on a 256-byte boundary
	const uint4 v = make_uint4(value, value, value, value);
	for (uint64_t i = gtid; i < n4; i += gsize) p4[i] = v;
	for (uint64_t i = n4 * 4 + gtid; i < n; i += gsize) ptr[i] = value;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- restricting bitmask + pre-scores
__device__ __forceinline__ bool ft_preselect_on(const FtPlan& p) {   // mergerimpl.h:486-490, the half only the device knows
	return p.prescore && __hip_atomic_load(&p.sync[kFtSyncPop], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > p.merge_limit;
}

// One workgroup per range of kFtRangeDocs documents; everything below lives in LDS until the range's mask words / scores are written.
//   restrictingMask_ = ~docsExcluded_ (mergerimpl.h:328-330; bits past total_docs stay 0 so that the popcount is exact)
//   AND term   calcTermBitmask (:252-274): any occurrence with a relevant field (checkFieldsRelevance, phrasemergerimpl.h:93-125); mask &= it
//   NOT term   excludeTermFromBitmask (:276-287)
//   pre-score  calcTermScores (:289-324) for every term that is not a NOT, when the host half of the 2-phase gate held: the first sub-term
//              (SortSubterms order) with a positive field boost adds min(proc16, 65535 / 4), saturating at 65535; then (:416-423) documents
//              outside the mask / removed score 0 and the rest is histogrammed
constexpr uint32_t kFtRangeSubs = 512;   // segments staged in LDS (queries with more sub-terms read the range index from HBM)
__global__ __launch_bounds__(256) void ft_ranges(FtPlan p) {
	constexpr uint32_t kWords = kFtRangeDocs / 32;
	__shared__ uint32_t s_mask[kWords], s_term[kWords], s_seen[kWords];
	__shared__ uint16_t s_score[kFtRangeDocs];
	__shared__ uint32_t s_keys[256], s_cnts[256], s_part[4];
	__shared__ uint32_t s_lo[kFtRangeSubs], s_hi[kFtRangeSubs];
	const uint32_t tid = threadIdx.x, range = blockIdx.x;
	const uint64_t d_begin = uint64_t(range) * kFtRangeDocs;
	const uint32_t docs_here = uint32_t(p.total_docs - d_begin < kFtRangeDocs ? p.total_docs - d_begin : kFtRangeDocs);
	for (uint32_t w = tid; w < kWords; w += 256) {
		const uint32_t d0 = w * 32;
		uint32_t bits = 0;
		if (d0 < docs_here) {
			const uint32_t cnt = docs_here - d0 < 32 ? docs_here - d0 : 32;
			if (!p.excluded) {
				bits = cnt == 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u);
			} else {
				for (uint32_t b = 0; b < cnt; ++b) bits |= (p.excluded[d_begin + d0 + b] ? 0u : 1u) << b;
			}
		}
		s_mask[w] = bits;
	}
	if (p.prescore) {
		for (uint32_t i = tid; i < kFtRangeDocs; i += 256) s_score[i] = 0;
		s_keys[tid] = 0;   // a score of 0 is never inserted
		s_cnts[tid] = 0;
	}
	// this range's segment [lo, hi) of every posting list, fetched up front (two dependent loads each, all in flight together); the merged
	// postings in front of the range = where its bucket of surviving postings starts (ft_rank_all)
	uint32_t before = 0;
	for (uint32_t si = tid; si < p.n_subs; si += 256) {
		const FtPosSubterm& s = p.subs[si];
		const uint32_t lo = range < s.n_ranges ? s.range_off[range] : uint32_t(s.n);
		if (si < kFtRangeSubs) {
			s_lo[si] = lo;
			s_hi[si] = range + 1 < s.n_ranges ? s.range_off[range + 1] : uint32_t(s.n);
		}
		if (s.qp != 0) before += lo;   // NOT terms are not merged
	}
	before = wave_sum(before);
	if ((tid & 63) == 0) s_part[tid >> 6] = before;
	__syncthreads();
	if (tid == 0) p.bucket_off[range] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
	__syncthreads();   // s_part is reused for the popcount below
	for (uint32_t t = 0; t < (p.simple ? 0u : p.nterms); ++t) {
		const FtTermCfg& term = p.terms[t];
		const int op = term.op;
		const bool want_score = p.prescore && op != 3;
		if (op == 1 && !want_score) continue;   // an OR term only matters to the pre-score
		for (uint32_t w = tid; w < kWords; w += 256) {
			s_term[w] = 0;
			s_seen[w] = 0;
		}
		__syncthreads();
		const bool need_entries = (op == 2 && !term.all_pos_boost) || (want_score && !term.same_boost);
		auto visit = [&](const FtPosSubterm& s, uint32_t i, uint32_t d) {
			const uint32_t local = uint32_t(d - d_begin);
			const uint32_t bit = 1u << (local & 31);
			if (op == 3) {
				atomicAnd(&s_mask[local >> 5], ~bit);
				return;
			}
			bool rel = term.all_pos_boost;
			float mb = term.field_boost[0];
			if (need_entries) {   // maxFieldsBoost (phrasemergerimpl.h:127-160) / relevance of the occurrence
				mb = 0.0f;
				rel = false;
				for (uint32_t e = s.ent_off[i], e1 = s.ent_off[i + 1]; e < e1; ++e) {
					const float fb = term.field_boost[s.ent_field[e]];
					mb = fmaxf(mb, fb);
					rel = rel || fb != 0.0f;
				}
				if (term.same_boost) mb = term.field_boost[0];
				if (term.all_pos_boost) rel = true;
			}
			if (op == 2 && rel) atomicOr(&s_term[local >> 5], bit);
			if (want_score && mb > 0.0f) {
				// termMask: documents are unique inside a sub-term and earlier sub-terms are behind a barrier, so the first one wins
				const uint32_t old = atomicOr(&s_seen[local >> 5], bit);
				if (!(old & bit)) {
					const float proc = s.proc * mb * term.opts_boost;
					uint32_t p16 = uint32_t(int32_t(proc)) & 0xFFFFu;   // static_cast<uint16_t>(float) as x86 evaluates it
					p16 = p16 < 65535u / 4 ? p16 : 65535u / 4;
					const uint32_t cur = s_score[local];
					p16 = p16 < 65535u - cur ? p16 : 65535u - cur;
					s_score[local] = uint16_t(cur + p16);
				}
			}
		};
		// Sub-terms go in SortSubterms order behind barriers ("the first one holding the document wins"), but their document ids do not depend
		// on each other: the first 1024 postings of up to four sub-terms are fetched together — one load latency per group, not per sub-term.
		constexpr int kGroup = 4;
		for (uint32_t g0 = term.sub_begin; g0 < term.sub_end; g0 += kGroup) {
			uint32_t lo[kGroup], hi[kGroup], dd[kGroup][4];
#pragma unroll
			for (int j = 0; j < kGroup; ++j) {
				const uint32_t si = g0 + j;
				lo[j] = hi[j] = 0;
				if (si < term.sub_end) {
					if (si < kFtRangeSubs) {
						lo[j] = s_lo[si];
						hi[j] = s_hi[si];
					} else {
						const FtPosSubterm& s = p.subs[si];
						lo[j] = range < s.n_ranges ? s.range_off[range] : uint32_t(s.n);
						hi[j] = range + 1 < s.n_ranges ? s.range_off[range + 1] : uint32_t(s.n);
					}
				}
			}
#pragma unroll
			for (int j = 0; j < kGroup; ++j) {
				const uint32_t* doc = g0 + j < term.sub_end ? p.subs[g0 + j].doc : nullptr;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const uint32_t idx = lo[j] + uint32_t(q) * 256 + tid;
					dd[j][q] = idx < hi[j] ? doc[idx] : 0u;
				}
			}
#pragma unroll
			for (int j = 0; j < kGroup; ++j) {
				if (g0 + j >= term.sub_end) break;   // uniform
				const FtPosSubterm& s = p.subs[g0 + j];
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const uint32_t idx = lo[j] + uint32_t(q) * 256 + tid;
					if (idx < hi[j]) visit(s, idx, dd[j][q]);
				}
				for (uint32_t base = lo[j] + 4 * 256; base < hi[j]; base += 4 * 256) {   // the rest of a long segment
					uint32_t idx[4], d4[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						idx[q] = base + uint32_t(q) * 256 + tid;
						d4[q] = idx[q] < hi[j] ? s.doc[idx[q]] : 0u;
					}
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						if (idx[q] < hi[j]) visit(s, idx[q], d4[q]);
					}
				}
				__syncthreads();   // the next sub-term of the term sees this one's documents
			}
		}
		if (op == 2) {   // restrictingMask_ &= termMask (an AND term without postings empties the range)
			for (uint32_t w = tid; w < kWords; w += 256) s_mask[w] &= s_term[w];
			__syncthreads();
		}
	}
	// ---- the range's mask words + their popcount (the device half of the 2-phase gate)
	uint32_t c = 0;
	for (uint32_t w = tid; w < kWords; w += 256) {
		const uint64_t gw = d_begin / 32 + w;
		if (gw < p.nwords) {
			p.mask[gw] = s_mask[w];
			c += __popc(s_mask[w]);
		}
	}
	c = wave_sum(c);
	if ((tid & 63) == 0) s_part[tid >> 6] = c;
	__syncthreads();
	if (tid == 0) {
		const uint32_t tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
		if (tot) atomicAdd(&p.sync[kFtSyncPop], tot);
	}
	if (!p.prescore) return;
	// ---- scores: masked-out / removed documents score 0 (mergerimpl.h:416-423); histogram of the rest through a small LDS hash table
	for (uint32_t q = tid; q < kFtRangeDocs / 4; q += 256) {
		const uint32_t l0 = q * 4;
		if (l0 >= docs_here) break;
		const uint32_t mw = s_mask[l0 >> 5];
		uint32_t rm = 0;
		if (p.removed) rm = *reinterpret_cast<const uint32_t*>(p.removed + d_begin + l0);   // 4 flags; reads past the end stay inside the allocation
		uint32_t sc[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool in = l0 + k < docs_here && ((mw >> ((l0 + k) & 31)) & 1u) && !((rm >> (8 * k)) & 0xFFu);
			sc[k] = in ? uint32_t(s_score[l0 + k]) : 0u;
		}
		*reinterpret_cast<uint2*>(p.score + d_begin + l0) = make_uint2(sc[0] | (sc[1] << 16), sc[2] | (sc[3] << 16));   // the array is padded to whole mask words
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t v = sc[k];
			if (!v) continue;
			uint32_t h = (v * 2654435761u) >> 24;
			int probes = 0;
			for (; probes < 256; ++probes, h = (h + 1) & 255u) {
				uint32_t cur = s_keys[h];
				if (cur != v) {
					if (cur != 0u) continue;
					cur = atomicCAS(&s_keys[h], 0u, v);
					if (cur != 0u && cur != v) continue;
				}
				atomicAdd(&s_cnts[h], 1u);
				break;
			}
			if (probes == 256) atomicAdd(&p.hist[v], 1u);   // more than 256 distinct scores in one range
		}
	}
	__syncthreads();
	if (s_keys[tid]) atomicAdd(&p.hist[s_keys[tid]], s_cnts[tid]);
}

// mergerimpl.h:433-446: walk the scores downwards until maxMergedDocs documents are covered.  A score sc is visited iff the documents
// strictly above it are fewer than maxMergedDocs; minScore = the lowest visited score >= 1, minScoreDocs = maxMergedDocs - (documents above
// it).  `above` is monotone, so the boundary falls inside ONE 64-score chunk: chunk sums by coalesced loads + wave reductions, a suffix
// scan over the 1024 chunks, then one wavefront resolves the boundary chunk.
__global__ __launch_bounds__(1024) void ft_preselect_pick(FtPlan p) {
	if (!ft_preselect_on(p)) return;
	__shared__ unsigned long long suffix[1024];   // documents in this chunk and every higher one
	__shared__ uint32_t chunk_sum[1024];
	__shared__ int s_chunk;
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	{   // thread t sums chunk t = scores [64 t, 64 t + 64): sixteen independent 16-byte loads in flight
		const uint4* h4 = reinterpret_cast<const uint4*>(p.hist) + size_t(t) * 16;
		uint4 v[16];
#pragma unroll
		for (int i = 0; i < 16; ++i) v[i] = h4[i];
		uint32_t c = 0;
#pragma unroll
		for (int i = 0; i < 16; ++i) c += v[i].x + v[i].y + v[i].z + v[i].w;
		chunk_sum[t] = c;
	}
	if (t == 0) s_chunk = -1;
	__syncthreads();
	suffix[t] = chunk_sum[t];
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {   // inclusive suffix sums
		const unsigned long long v = t + off < 1024 ? suffix[t + off] : 0;
		__syncthreads();
		suffix[t] += v;
		__syncthreads();
	}
	// the lowest chunk whose TOP score is visited (documents in higher chunks < maxMergedDocs); chunk 0 only counts through scores >= 1
	const unsigned long long above_chunk = suffix[t] - chunk_sum[t];
	const bool top_visited = above_chunk < p.max_merged;
	const bool below_visited = t > 0 && suffix[t] < p.max_merged;   // would the top of chunk t - 1 be visited too?
	if (top_visited && !below_visited) s_chunk = t;   // exactly one thread: `above` is monotone
	__syncthreads();
	uint32_t* pick = p.sync + kFtSyncPick;
	const int c = s_chunk;
	if (c < 0) {   // unreachable (the top chunk has nothing above it), kept as the reference's initial values
		if (t == 0) {
			pick[0] = 65535u;
			pick[1] = 0;
		}
		return;
	}
	if (wave != 0) return;
	const uint32_t sc = uint32_t(c * 64 + lane);
	const uint32_t h = p.hist[sc];
	uint32_t incl = h;   // inclusive suffix over the lanes: documents with a score in [sc, top of the chunk]
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t o = __shfl_down(incl, off, 64);
		if (lane + off < 64) incl += o;
	}
	const unsigned long long above = (suffix[c] - chunk_sum[c]) + (incl - h);
	const bool visited = sc >= 1 && above < p.max_merged;
	const unsigned long long vis = __ballot(visited);
	if (!vis) {   // only score 0 of chunk 0 left: nothing with a positive score
		if (lane == 0) {
			pick[0] = 65535u;
			pick[1] = 0;
		}
		return;
	}
	const int low = __ffsll((long long)vis) - 1;   // visited lanes form a suffix of the wave: the lowest one is minScore
	if (lane == low) {
		pick[0] = sc;
		pick[1] = uint32_t(p.max_merged - above);
	}
}

// mergerimpl.h:448-462: kFtApplyWords mask words per thread (the ordered prefix chain is as long as the grid: fewer, fatter workgroups);
// ties at minScore are kept in document order up to minScoreDocs
__global__ __launch_bounds__(256) void ft_preselect_apply(FtPlan p) {
	if (!ft_preselect_on(p)) return;
	const uint32_t ticket = grab_ticket(p.sync + kFtSyncPreTicket);
	const uint64_t w0 = (uint64_t(ticket) * 256 + threadIdx.x) * kFtApplyWords;
	const uint32_t min_score = p.sync[kFtSyncPick], min_docs = p.sync[kFtSyncPick + 1];
	uint32_t bits[kFtApplyWords], gt[kFtApplyWords], tie[kFtApplyWords];
	uint32_t ties = 0;
#pragma unroll
	for (int j = 0; j < kFtApplyWords; ++j) {
		bits[j] = gt[j] = tie[j] = 0;
		const uint64_t w = w0 + j;
		if (w >= p.nwords) continue;
		bits[j] = p.mask[w];
		// the 32 scores of the word as four 16-byte loads (the score array is padded to a whole word); only masked-in documents count
		const uint4* s4 = reinterpret_cast<const uint4*>(p.score + w * 32);
		uint4 v[4];
#pragma unroll
		for (int q = 0; q < 4; ++q) v[q] = s4[q];
		const uint32_t half[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w,
								   v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
#pragma unroll
		for (int q = 0; q < 16; ++q) {
			const uint32_t lo = half[q] & 0xFFFFu, hi = half[q] >> 16;
			gt[j] |= (uint32_t(lo > min_score) << (2 * q)) | (uint32_t(hi > min_score) << (2 * q + 1));
			tie[j] |= (uint32_t(lo == min_score) << (2 * q)) | (uint32_t(hi == min_score) << (2 * q + 1));
		}
		gt[j] &= bits[j];
		tie[j] &= bits[j];
		ties += __popc(tie[j]);
	}
	uint32_t grand;
	const uint32_t excl = ordered_prefix(ties, ticket, p.lookback_pre, p.sync + kFtSyncError, &grand);
	uint32_t allowed = min_docs > excl ? min_docs - excl : 0;
#pragma unroll
	for (int j = 0; j < kFtApplyWords; ++j) {
		if (w0 + j >= p.nwords) continue;
		uint32_t keep = gt[j], tj = tie[j];
		while (tj && allowed) {
			const uint32_t low = tj & (0u - tj);
			keep |= low;
			tj ^= low;
			--allowed;
		}
		if (keep != bits[j]) p.mask[w0 + j] = keep;
	}
}

// ---------------------------------------------------------------------------------------------- mergeTerm / mergeSimple
// calcTermRank of every eligible posting of the query (restrictingMask_, DocRemoved); the postings that rank non-zero leave as 16-byte
// records in the bucket of their document range.
// The gathers of one posting form a dependent chain (doc -> mask word -> removed flag | entries -> words in field).  The cheap half
// (document, mask bit, removed flag) is streamed for kFtRankTiles x 1024 postings per workgroup, four postings per thread and tile with
// every stage issued for all of them before anything is consumed; the survivors are COMPACTED in LDS and the expensive half
// (calcTermRank, five dependent gathers) then runs once over the compact list, one posting per lane.  After a preselect ~1 % of the
// postings survive: evaluated in place, nearly every wavefront held one and paid the chain up to four times in a row (once per
// item slot) — 38 us for 3.9 M postings; a first attempt with fatter threads made that 64 us.
constexpr int kFtRankTiles = 2;
__global__ __launch_bounds__(256) void ft_rank_all(FtPlan p) {
	__shared__ uint32_t s_cnt;
	__shared__ uint16_t s_item[kFtRankTiles * kFtBlockPostings];   // tile << 10 | thread << 2 | slot
	__shared__ uint32_t s_doc[kFtRankTiles * kFtBlockPostings];
	__shared__ uint32_t s_sub[kFtRankTiles], s_base[kFtRankTiles];
	if (threadIdx.x == 0) s_cnt = 0;
	if (threadIdx.x < kFtRankTiles) {
		const uint32_t tile = blockIdx.x * kFtRankTiles + threadIdx.x;
		if (tile < p.merge_blocks) {
			const FtGridEntry ge = grid_entry(p.merge_grid, p.n_merge_entries, tile);
			s_sub[threadIdx.x] = ge.sub;
			s_base[threadIdx.x] = ge.block_base;
		}
	}
	__syncthreads();
	uint32_t docs[kFtRankTiles][kFtPassItems];
	bool live[kFtRankTiles][kFtPassItems];
#pragma unroll
	for (int g = 0; g < kFtRankTiles; ++g) {
		const uint32_t tile = blockIdx.x * kFtRankTiles + g;
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) {
			live[g][k] = false;
			docs[g][k] = 0;
		}
		if (tile >= p.merge_blocks) continue;
		const FtPosSubterm& s = p.subs[s_sub[g]];
		const uint64_t i0 = uint64_t(tile - s_base[g]) * kFtBlockPostings + uint64_t(threadIdx.x) * kFtPassItems;
		if (i0 < s.n) load_docs(s, i0, docs[g], live[g]);
	}
	{
		uint32_t mw[kFtRankTiles][kFtPassItems];
#pragma unroll
		for (int g = 0; g < kFtRankTiles; ++g) {
#pragma unroll
			for (int k = 0; k < kFtPassItems; ++k) mw[g][k] = live[g][k] ? p.mask[docs[g][k] >> 5] : 0u;
		}
#pragma unroll
		for (int g = 0; g < kFtRankTiles; ++g) {
#pragma unroll
			for (int k = 0; k < kFtPassItems; ++k) live[g][k] = live[g][k] && ((mw[g][k] >> (docs[g][k] & 31)) & 1u);   // restrictingMask_
		}
	}
	if (p.removed && p.check_removed && !ft_preselect_on(p)) {   // needToCheckRemoved_ is false once the preselect has run (mergerimpl.h:463)
		uint8_t rm[kFtRankTiles][kFtPassItems];
#pragma unroll
		for (int g = 0; g < kFtRankTiles; ++g) {
#pragma unroll
			for (int k = 0; k < kFtPassItems; ++k) rm[g][k] = live[g][k] ? p.removed[docs[g][k]] : uint8_t(0);
		}
#pragma unroll
		for (int g = 0; g < kFtRankTiles; ++g) {
#pragma unroll
			for (int k = 0; k < kFtPassItems; ++k) live[g][k] = live[g][k] && !rm[g][k];
		}
	}
#pragma unroll
	for (int g = 0; g < kFtRankTiles; ++g) {
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) {
			if (!live[g][k]) continue;
			const uint32_t at = atomicAdd(&s_cnt, 1u);
			s_item[at] = uint16_t((uint32_t(g) << 10) | (threadIdx.x << 2) | uint32_t(k));
			s_doc[at] = docs[g][k];
		}
	}
	__syncthreads();
	const uint32_t cnt = s_cnt;
	const int lane = threadIdx.x & 63;
	for (uint32_t e0 = 0; e0 < cnt; e0 += 256) {   // uniform trip count: the wavefront votes below
		const uint32_t e = e0 + threadIdx.x;
		bool want = false;
		uint32_t d = 0, row = 0, idx = 0;
		float rank = 0.0f;
		uint8_t field = 0;
		if (e < cnt) {
			const uint32_t item = s_item[e], g = item >> 10, local = item & 1023u;
			const uint32_t tile = blockIdx.x * kFtRankTiles + g;
			const FtPosSubterm& s = p.subs[s_sub[g]];
			const FtTermCfg& t = p.terms[s.term];
			const uint64_t i = uint64_t(tile - s_base[g]) * kFtBlockPostings + local;
			d = s_doc[e];
			rank = ft_term_rank(t, s, s.ent_off[i], s.ent_off[i + 1], d, &field);
			want = rank != 0.0f;
			row = s.row;
			idx = uint32_t(i);
		}
		// one device atomic per (wavefront, document range): neighbouring postings share a range, whole-corpus merges would otherwise
		// hammer a few hundred counters with millions of same-address atomics
		const uint32_t rg = d >> kFtRangeShift;
		unsigned long long pending = __ballot(want);
		while (pending) {
			const int leader = __ffsll((long long)pending) - 1;
			const uint32_t lrg = uint32_t(__shfl(int(rg), leader, 64));
			const bool mine = want && rg == lrg;
			const unsigned long long same = __ballot(mine);
			uint32_t base = 0;
			if (lane == leader) base = atomicAdd(&p.bucket_cnt[lrg], uint32_t(__popcll(same)));
			base = uint32_t(__shfl(int(base), leader, 64));
			if (mine) {
				const uint32_t pos = base + uint32_t(__popcll(same & ((1ull << lane) - 1ull)));
				p.b_rec[uint64_t(p.bucket_off[lrg]) + pos] = make_uint4(d, idx, __float_as_uint(rank), row | (uint32_t(field) << 16));
				want = false;
			}
			pending &= ~same;
		}
	}
}

// 16-bit minimum in an LDS table of packed halves (there are no 16-bit LDS atomics; contention is one posting per (document, sub-term))
__device__ __forceinline__ void lds_min_u16(uint32_t* words, uint32_t idx, uint32_t v) {
	uint32_t* w = words + (idx >> 1);
	const uint32_t sh = (idx & 1u) * 16u;
	uint32_t old = *w;
	while (((old >> sh) & 0xFFFFu) > v) {
		const uint32_t nw = (old & ~(0xFFFFu << sh)) | (v << sh);
		const uint32_t prev = atomicCAS(w, old, nw);
		if (prev == old) break;
		old = prev;
	}
}
__device__ __forceinline__ uint32_t lds_get_u16(const uint32_t* words, uint32_t idx) { return (words[idx >> 1] >> ((idx & 1u) * 16u)) & 0xFFFFu; }
__device__ __forceinline__ void lds_set_u16(uint32_t* words, uint32_t idx, uint32_t v) {   // plain store of one half (ds_write_b16)
	reinterpret_cast<uint16_t*>(words)[idx] = uint16_t(v);
}

// addDoc order (merger.h:161-180), document side.  One workgroup per range: the row (sub-term) of every document's first surviving
// posting, counted per (row, range); the LAST workgroup to finish replaces the table by its exclusive prefix in row-major order — the merge
// slot of the first document of every (row, range) — and publishes the number of merged documents.  Also hands the next kernels / the next
// merge their zeroed tables (entry rows, histogram, look-back words).
constexpr uint32_t kFtAdderRowsLds = 1024;
__global__ __launch_bounds__(256) void ft_adders(FtPlan p) {
	__shared__ uint32_t s_tab[kFtRangeDocs / 2];   // first row of every document of the range, 16 bits each
	__shared__ uint32_t s_rowcnt[kFtAdderRowsLds];
	__shared__ uint32_t s_part[4];
	__shared__ uint32_t s_last;
	const uint32_t tid = threadIdx.x, range = blockIdx.x;
	{
		const uint64_t gtid = uint64_t(blockIdx.x) * blockDim.x + tid, gsize = uint64_t(gridDim.x) * blockDim.x;
		fill_words(reinterpret_cast<uint32_t*>(p.e_rank), uint64_t(p.n_rows) * p.max_merged, 0u, gtid, gsize);
		if (p.prescore) {   // ft_preselect_apply was their last reader
			fill_words(p.hist, 65536, 0u, gtid, gsize);
			fill_words(reinterpret_cast<uint32_t*>(p.lookback_pre), ((p.nwords + 256 * kFtApplyWords - 1) / (256 * kFtApplyWords)) * 2, 0u, gtid, gsize);
		}
	}
	const uint32_t n = p.bucket_cnt[range];
	const bool lds_rows = p.n_rows <= kFtAdderRowsLds;
	for (uint32_t w = tid; w < kFtRangeDocs / 2; w += 256) s_tab[w] = 0xFFFFFFFFu;
	for (uint32_t r = tid; r < p.n_rows; r += 256) {
		if (lds_rows) {
			s_rowcnt[r] = 0;
		} else {
			p.adders[uint64_t(r) * p.n_ranges + range] = 0;   // this workgroup owns the column
		}
	}
	__syncthreads();
	const uint4* rec = p.b_rec + p.bucket_off[range];
	for (uint32_t e = tid; e < n; e += 256) {
		const uint4 r = rec[e];
		lds_min_u16(s_tab, r.x & (kFtRangeDocs - 1), r.w & 0xFFFFu);
	}
	__syncthreads();
	for (uint32_t e = tid; e < n; e += 256) {
		const uint4 r = rec[e];
		const uint32_t row = r.w & 0xFFFFu;
		if (lds_get_u16(s_tab, r.x & (kFtRangeDocs - 1)) != row) continue;   // one posting per (document, row): exactly one record adds the document
		if (lds_rows) {
			atomicAdd(&s_rowcnt[row], 1u);
		} else {
			atomicAdd(&p.adders[uint64_t(row) * p.n_ranges + range], 1u);
		}
	}
	__syncthreads();
	if (lds_rows) {
		for (uint32_t r = tid; r < p.n_rows; r += 256) p.adders[uint64_t(r) * p.n_ranges + range] = s_rowcnt[r];
	}
	// ---- the last workgroup scans the table
	__threadfence();
	__syncthreads();
	if (tid == 0) s_last = atomicAdd(&p.sync[kFtSyncDoneAdders], 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	if (!s_last) return;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other workgroups' counts (their release: the fence before the counter)
	const uint64_t total = uint64_t(p.n_rows) * p.n_ranges, per = (total + 255) / 256;
	const uint64_t a = std::min<uint64_t>(total, uint64_t(tid) * per), b = std::min<uint64_t>(total, a + per);
	uint32_t local = 0;
	for (uint64_t j = a; j < b; ++j) local += p.adders[j];
	const uint32_t incl = wave_inclusive_scan(local, int(tid & 63));
	if ((tid & 63) == 63) s_part[tid >> 6] = incl;
	__syncthreads();
	uint32_t running = incl - local;
	for (uint32_t w = 0; w < (tid >> 6); ++w) running += s_part[w];
	for (uint64_t j = a; j < b; ++j) {
		const uint32_t v = p.adders[j];
		p.adders[j] = running;
		running += v;
	}
	if (tid == 255) p.sync[kFtSyncNumDocs] = running < p.max_merged ? running : p.max_merged;   // the last thread ends on the grand total
}

// mergerimpl.h:20-37; fullPos()/fullField() truncate the 64-bit PosType to uint32_t exactly like the reference's accessors
__device__ __forceinline__ unsigned ft_positions_distance(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb) {
	unsigned res = 0xFFFFFFFFu;
	uint32_t i = 0, j = 0;
	while (i < na && j < nb) {
		const uint64_t pa = a[i], pb = b[j];
		const uint32_t fa = uint32_t(pa), fb = uint32_t(pb);
		const bool sign = fa > fb;
		if (uint32_t(pa >> 28) == uint32_t(pb >> 28)) {
			const unsigned dst = sign ? fa - fb : fb - fa;
			if (dst < res) {
				res = dst;
				if (res <= 1) break;
			}
		}
		if (sign) {
			++j;
		} else {
			++i;
		}
	}
	return res == 0xFFFFFFFFu ? 0 : res;
}

// One merged document: its row of the entry table replayed in sub-term order = the order mergeTerm / mergeSimple met its postings.
constexpr uint32_t kFtReplayRows = 128;   // sub-term descriptors staged in LDS (queries with more merged sub-terms read the plan from HBM)
__device__ __forceinline__ void ft_replay_doc(const FtPlan& p, uint32_t sl, uint32_t doc, const uint64_t* const* s_fpos, const uint32_t* const* s_pos_off,
											  const uint16_t* s_qp) {
	bool created = false;
	float proc = 0.f, rank = 0.f;
	uint8_t field = 0;
	const uint64_t* last_ptr = nullptr;
	const uint64_t* next_ptr = nullptr;
	uint32_t last_cnt = 0, next_cnt = 0;
	uint16_t switched_term = 0, last_counted = 0, terms_counter = 0;
	// Most documents meet ONE sub-term, a few two or three, out of many: each lane first collects WHICH of its rows are occupied (rank
	// loads, 64 rows per mask word) and then walks only those, in row order = the order mergeTerm met the postings.
	for (uint32_t row0 = 0; row0 < p.n_rows; row0 += 64) {
		unsigned long long occupied = 0;
		const uint32_t rows_here = p.n_rows - row0 < 64 ? p.n_rows - row0 : 64;
		for (uint32_t j0 = 0; j0 < rows_here; j0 += 8) {
			float ahead[8];
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j) ahead[j] = j0 + j < rows_here ? p.e_rank[uint64_t(row0 + j0 + j) * p.max_merged + sl] : 0.0f;
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j) occupied |= (unsigned long long)(ahead[j] != 0.0f) << (j0 + j);
		}
		while (occupied) {
			const uint32_t row = row0 + uint32_t(__ffsll((long long)occupied) - 1);
			occupied &= occupied - 1;
			const uint64_t cell = uint64_t(row) * p.max_merged + sl;
			const float r = p.e_rank[cell];
			const uint8_t fld = p.e_field[cell];
			if (p.simple) {   // mergeSimple, mergerimpl.h:232-240: strict <, so the first maximum (and its field) wins
				if (!created) {
					created = true;
					proc = r;
					field = fld;
				} else if (proc < r) {
					proc = r;
					field = fld;
				}
				continue;
			}
			const uint32_t i = p.e_idx[cell];
			const uint64_t* fpos;
			const uint32_t* pos_off;
			uint16_t qp;
			if (row < kFtReplayRows) {
				fpos = s_fpos[row];
				pos_off = s_pos_off[row];
				qp = s_qp[row];
			} else {
				const FtPosSubterm& s = p.subs[p.merge_grid[row].sub];
				fpos = s.fpos;
				pos_off = s.pos_off;
				qp = s.qp;
			}
			const uint32_t po0 = pos_off[i], po1 = pos_off[i + 1];
			const uint64_t* pos = fpos + po0;
			const uint32_t npos = po1 - po0;
			if (!created) {   // addDoc (mergerimpl.h:160-164)
				created = true;
				proc = r;
				field = fld;
				rank = r;
				next_ptr = pos;
				next_cnt = npos;
				switched_term = qp;
				last_counted = qp;
				terms_counter = 1;
				continue;
			}
			// ---- document already merged: mergerimpl.h:165-189
			if (switched_term < qp) {   // switchToNextWord (merger.h:218-226) ran before every term since: idempotent after the first time
				if (next_cnt) {
					last_ptr = next_ptr;
					last_cnt = next_cnt;
					next_cnt = 0;
					rank = 0.f;
				}
				switched_term = qp;
			}
			if (last_counted < qp) {   // InreaseTermsCounter
				terms_counter = uint16_t(terms_counter + 1);
				last_counted = qp;
			}
			unsigned dist = ft_positions_distance(last_ptr, last_cnt, pos, npos);
			dist = dist > 1u ? dist : 1u;
			const float norm_dist = ft_bound(float(1.0 / double(float(dist))), p.distance_weight, p.distance_boost);
			const float final_rank = norm_dist * r;
			if (final_rank > rank) {
				proc -= rank;
				proc += final_rank;
				next_ptr = pos;
				next_cnt = npos;
				rank = final_rank;
			}
		}
	}
	// addFullMatchBoost (merger.h:100-109): a document whose best field holds exactly as many words as the query has parts — and, for a
	// multi-term query, that met every part (canBeBoostedByFullMatch, mergerimpl.h:527-531) — is boosted.  Done here because the word counts
	// are resident: on the host it was one cache miss per merged document.
	{
		const FtTermCfg& t0 = p.terms[0];
		const float words = t0.words[size_t(doc) * t0.num_fields + field];
		const bool full = p.simple ? words == 1.0f : (terms_counter == p.nterms && words == float(p.nterms));
		if (full) proc = float(double(proc) * p.full_match_boost);
	}
	p.out_proc[sl] = proc;
	p.out_field[sl] = field;
	p.out_terms_counter[sl] = terms_counter;
}

// Slots, entry rows and the per-document replay of one document range.  Dynamic LDS: the 16-bit document table (first row, then the
// position in the sorted key list) followed by the key list of the range's first postings ((row << 13 | document), then the slot).
__global__ __launch_bounds__(256) void ft_finish(FtPlan p) {
	extern __shared__ uint32_t ft_finish_lds[];
	uint32_t* s_tab = ft_finish_lds;                       // [kFtRangeDocs / 2]
	uint32_t* s_keys = ft_finish_lds + kFtRangeDocs / 2;   // [kFtRangeDocs]
	__shared__ const uint64_t* s_fpos[kFtReplayRows];
	__shared__ const uint32_t* s_pos_off[kFtReplayRows];
	__shared__ uint16_t s_qp[kFtReplayRows];
	__shared__ uint32_t s_nadd, s_last;
	const uint32_t tid = threadIdx.x, range = blockIdx.x;
	const uint32_t n = p.bucket_cnt[range];
	if (n) {
		for (uint32_t row = tid; row < p.n_rows && row < kFtReplayRows; row += 256) {   // two dependent loads per row, once per workgroup
			const FtPosSubterm& s = p.subs[p.merge_grid[row].sub];
			s_fpos[row] = s.fpos;
			s_pos_off[row] = s.pos_off;
			s_qp[row] = s.qp;
		}
		for (uint32_t w = tid; w < kFtRangeDocs / 2; w += 256) s_tab[w] = 0xFFFFFFFFu;
		if (tid == 0) s_nadd = 0;
		__syncthreads();
		uint4* rec = p.b_rec + p.bucket_off[range];
		for (uint32_t e = tid; e < n; e += 256) {
			const uint4 r = rec[e];
			lds_min_u16(s_tab, r.x & (kFtRangeDocs - 1), r.w & 0xFFFFu);
		}
		__syncthreads();
		for (uint32_t e = tid; e < n; e += 256) {   // the range's first postings: key (row, document); the record remembers that it adds
			const uint4 r = rec[e];
			const uint32_t row = r.w & 0xFFFFu, dl = r.x & (kFtRangeDocs - 1);
			if (lds_get_u16(s_tab, dl) != row) continue;
			s_keys[atomicAdd(&s_nadd, 1u)] = (row << kFtRangeShift) | dl;
			rec[e].w = r.w | 0x80000000u;
		}
		__syncthreads();
		const uint32_t A = s_nadd;   // <= kFtRangeDocs: one per document
		uint32_t N = 2;
		while (N < A) N <<= 1;
		for (uint32_t q = A + tid; q < N; q += 256) s_keys[q] = 0xFFFFFFFFu;
		__syncthreads();
		for (uint32_t k = 2; k <= N; k <<= 1) {   // bitonic sort, ascending
			for (uint32_t j = k >> 1; j > 0; j >>= 1) {
				for (uint32_t t = tid; t < N / 2; t += 256) {
					const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
					const uint32_t x = s_keys[i], y = s_keys[l];
					if ((x > y) == ((i & k) == 0)) {
						s_keys[i] = y;
						s_keys[l] = x;
					}
				}
				__syncthreads();
			}
		}
		// rank inside the row = position - first position of the row (binary search); parked in the document table, whose first-row
		// entries are no longer needed
		for (uint32_t q = tid; q < A; q += 256) {
			const uint32_t key = s_keys[q], first_of_row = key & ~(kFtRangeDocs - 1);
			uint32_t lo = 0, hi = q;   // lower bound of first_of_row in [0, q]
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_keys[mid] < first_of_row) {
					lo = mid + 1;
				} else {
					hi = mid;
				}
			}
			lds_set_u16(s_tab, key & (kFtRangeDocs - 1), q - lo);
		}
		__syncthreads();
		const uint32_t d_begin = range << kFtRangeShift;
		for (uint32_t q = tid; q < A; q += 256) {   // key -> slot; document -> its position in the list
			const uint32_t key = s_keys[q], row = key >> kFtRangeShift, dl = key & (kFtRangeDocs - 1);
			const uint32_t slot = p.adders[uint64_t(row) * p.n_ranges + range] + lds_get_u16(s_tab, dl);
			s_keys[q] = slot;
			lds_set_u16(s_tab, dl, q);
			if (slot < p.max_merged) p.out_doc[slot] = d_begin + dl;
		}
		__syncthreads();
		for (uint32_t e = tid; e < n; e += 256) {   // every posting of a merged document into the document's row, column = its sub-term
			const uint4 r = rec[e];
			const uint32_t sl = s_keys[lds_get_u16(s_tab, r.x & (kFtRangeDocs - 1))];
			if (sl >= p.max_merged) continue;   // met after the limit was hit: never added
			const uint64_t cell = uint64_t(r.w & 0xFFFFu) * p.max_merged + sl;
			p.e_rank[cell] = __uint_as_float(r.z);
			p.e_idx[cell] = r.y;
			p.e_field[cell] = uint8_t((r.w >> 16) & 0xFFu);
		}
		__syncthreads();   // the rows are read back by this workgroup only
		for (uint32_t e = tid; e < n; e += 256) {
			const uint4 r = rec[e];
			if (!(r.w >> 31)) continue;
			const uint32_t sl = s_keys[lds_get_u16(s_tab, r.x & (kFtRangeDocs - 1))];
			if (sl < p.max_merged) ft_replay_doc(p, sl, r.x, s_fpos, s_pos_off, s_qp);
		}
	}
	// ---- leave the shared tables clean for the next merge; the last workgroup writes the result header
	__syncthreads();
	if (tid == 0) {
		if (n) p.bucket_cnt[range] = 0;
		__threadfence();
		s_last = atomicAdd(&p.sync[kFtSyncDoneFinish], 1u) == gridDim.x - 1 ? 1u : 0u;
	}
	__syncthreads();
	if (!s_last) return;
	if (tid == 0) {
		p.out_header[0] = p.sync[kFtSyncNumDocs];
		p.out_header[1] = p.sync[kFtSyncError];
		p.out_header[2] = ft_preselect_on(p) ? 1u : 0u;
		p.out_header[3] = 0;
	}
	__syncthreads();
	if (tid < kFtSyncWords) p.sync[tid] = 0;
}

// ---------------------------------------------------------------------------------------------- launch train
hipError_t launch_ft_merge(const FtPlan& p, hipStream_t st) {
	constexpr size_t kFinishLds = (kFtRangeDocs / 2 + kFtRangeDocs) * sizeof(uint32_t);
	static std::atomic<uint64_t> raised{0};
	if (hipError_t e = raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&ft_finish), kFinishLds); e != hipSuccess) return e;
	hipLaunchKernelGGL(ft_ranges, dim3(p.n_ranges), dim3(256), 0, st, p);
	if (!p.simple && p.prescore) {
		hipLaunchKernelGGL(ft_preselect_pick, dim3(1), dim3(1024), 0, st, p);
		hipLaunchKernelGGL(ft_preselect_apply, dim3(uint32_t((p.nwords + 256 * kFtApplyWords - 1) / (256 * kFtApplyWords))), dim3(256), 0, st, p);
	}
	if (p.merge_blocks) hipLaunchKernelGGL(ft_rank_all, dim3((p.merge_blocks + kFtRankTiles - 1) / kFtRankTiles), dim3(256), 0, st, p);
	hipLaunchKernelGGL(ft_adders, dim3(p.n_ranges), dim3(256), 0, st, p);
	hipLaunchKernelGGL(ft_finish, dim3(p.n_ranges), dim3(256), kFinishLds, st, p);
	return hipGetLastError();
}

}  // namespace rxgpu
