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
//    a per-word range index, pulls the range's postings into registers in one round (256-posting blocks, every load independent) and
//    resolves everything in LDS — bit arrays for the masks, a 16-bit score and a "scored in this term" byte per document, sub-terms in
//    order behind workgroup barriers — without one global atomic  ->  ft_ranges.  (The first cut scattered postings into per-term arrays
//    with device-scope atomics: 140 us for a 3 x 3 query; atomics on random addresses resolve past the per-XCD L2s at ~28 G/s.)
//  * admission (addDoc until maxMergedDocs, merger.h:161-180): a document is added by its first posting (in global posting order) that is
//    eligible and has a non-zero rank; it gets the next merge slot if fewer than maxMergedDocs documents were added before it, and once the
//    limit is hit nothing is added any more.  Global posting order is (sub-term row, document) — every list ascends by document — so the
//    slot of a document is  #documents first met in an earlier row  +  #documents first met in the same row with a smaller id.
//    ft_rank_all ranks every eligible posting (calcTermRank) and drops the survivors (rank != 0; ~1 % after a preselect) into per-document-
//    range buckets; ft_adders, one workgroup per range of 8192 documents, finds every document's first row (16-bit minimum in LDS) and
//    counts them per (row, range); the exclusive prefix of that small table in row-major order = the slot bases (summed by every workgroup
//    of ft_finish for itself when the table is small, by ft_slot_bases otherwise).
//    (The first cuts ran this posting-side: an atomicMin table over all documents plus three passes over ALL postings — count, ordered
//    prefix, scatter — 45 us of a 128 us merge, each pass bound by its dependent gathers, not by bytes.)
//  * per-document state (`proc -= rank; proc += finalRank` on every strict improvement, switchToNextWord between terms, termsCounter):
//    a document meets at most one posting per sub-term and all its postings sit in ONE bucket, so ft_finish — again one workgroup per
//    range — ranks the range's first postings inside (row, range) (per-row bitmaps + popcount prefix up to 8 sub-terms, an LDS sort of
//    (row, document) keys beyond), adds the slot bases, drops the range's survivors into a per-slot row indexed by sub-term and, behind a
//    workgroup barrier, replays each of its documents in sub-term order with the reference's float operations  ->  same bits.
//  * preselect ties at the threshold score are kept in document order: ordered prefix (ft_preselect_apply).
//
// Launch train of a multi-term query: ft_ranges, [ft_preselect_apply], ft_rank_all, ft_adders, [ft_slot_bases], ft_finish (+ ft_import in
// front and ft_export behind: plan and result travel through one pinned staging buffer by copy kernels); the 2-phase
// gate's popcount test is evaluated ON THE DEVICE (no host round trip), the result leaves in one packed buffer.  A Simple() query:
// ft_ranges (mask only), ft_rank_all, ft_adders, [ft_slot_bases], ft_finish.  No fill kernel: the tables a merge reads before it writes (histogram,
// entry-row occupancy, bucket counters, look-back and synchronisation words) are handed back ZEROED by the kernel that read them last.
//
// Bound: HBM gathers: per posting 4 B doc + 8 B entry offsets + 9 B per (field, tf, firstPos) entry streamed, 4 B
// words-in-field + the mask word gathered; 16 B per surviving posting written and read back.
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


constexpr int kFtApplyWords = 4;   // mask words per thread in ft_preselect_apply

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


// One workgroup per range of kFtRangeDocs documents; everything below lives in LDS until the range's mask words / scores are written.
//   restrictingMask_ = ~docsExcluded_ (mergerimpl.h:328-330; bits past total_docs stay 0 so that the popcount is exact)
//   AND term   calcTermBitmask (:252-274): any occurrence with a relevant field (checkFieldsRelevance, phrasemergerimpl.h:93-125); mask &= it
//   NOT term   excludeTermFromBitmask (:276-287)
//   pre-score  calcTermScores (:289-324) for every term that is not a NOT, when the host half of the 2-phase gate held: the first sub-term
//              (SortSubterms order) with a positive field boost adds min(proc16, 65535 / 4), saturating at 65535; then (:416-423) documents
//              outside the mask / removed score 0 and the rest is histogrammed
constexpr uint32_t kFtRangeSubs = 128;   // sub-terms whose segment, list pointer and proc are staged in LDS (more: read from HBM)
constexpr uint32_t kFtHistRep = 8;       // counters per key of the LDS score histogram (on distinct banks)
// LDS per workgroup ~30 KB -> 5 workgroups per CU.  The kernel is a chain of dependent phases (range index, posting stage, terms behind
// barriers, mask, scores, histogram): with Q queries in one grid its throughput is the number of workgroups a CU holds.  (512 staged
// sub-terms and 16 counters per key cost 48.6 KB = 3 per CU; queries with more than 128 sub-terms read the rest of their plan from HBM.)
constexpr uint32_t kFtStageBlocks = 32;  // 256-posting blocks of the range prefetched into registers (one posting per thread and block)
__global__ __launch_bounds__(256, 5) void ft_ranges(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	constexpr uint32_t kWords = kFtRangeDocs / 32;
	__shared__ uint32_t s_mask[kWords], s_term[kWords];
	__shared__ uint16_t s_score[kFtRangeDocs];
	__shared__ uint32_t s_keys[256], s_part[4];
	__shared__ uint32_t s_lo[kFtRangeSubs], s_hi[kFtRangeSubs], s_blk0[kFtRangeSubs + 1];
	__shared__ const uint32_t* s_docptr[kFtRangeSubs];
	__shared__ float s_proc[kFtRangeSubs];
	struct BlockInfo {            // one 256-posting block of a staged sub-term's segment
		const uint32_t* first;    // &doc[lo + 256 k]
		uint32_t count;           // postings in the block (1..256)
		uint32_t pad;
	};
	__shared__ __attribute__((aligned(16))) BlockInfo s_binfo[kFtStageBlocks];
	__shared__ __attribute__((aligned(16))) uint32_t s_rep[256 * kFtHistRep];   // score histogram: kFtHistRep counters per key (after the term pass)
	// During the term pass the same space holds one byte per document: the number (+1) of the last term that scored it.  "Already scored in
	// this term" is then a plain byte compare — documents are unique inside a sub-term and sub-terms sit behind barriers, so no two threads
	// touch one byte — where a bit set took an LDS atomic with return per posting: at 6400 postings per range and three ranges per CU the
	// atomics alone were ~10 us of this kernel.
	uint8_t* s_scored = reinterpret_cast<uint8_t*>(s_rep);   // [kFtRangeDocs]
	static_assert(sizeof(s_rep) >= kFtRangeDocs, "one byte per document fits the counter space");
	const uint32_t tid = threadIdx.x, range = blockIdx.x + p.range_begin;   // (a document-range shard runs its own ranges only)
	FT_STAMP(p, 0);
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
		for (uint32_t i = tid; i < kFtRangeDocs / 2; i += 256) reinterpret_cast<uint32_t*>(s_score)[i] = 0;
		for (uint32_t i = tid; i < kFtRangeDocs / 4; i += 256) reinterpret_cast<uint32_t*>(s_scored)[i] = 0;
		s_keys[tid] = 0;   // a score of 0 is never inserted
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
			s_docptr[si] = s.doc;
			s_proc[si] = s.proc;
		}
		if (s.qp != 0) before += lo;   // NOT terms are not merged
	}
	before = wave_sum(before);
	if ((tid & 63) == 0) s_part[tid >> 6] = before;
	__syncthreads();
	if (tid == 0) p.bucket_off[range] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
	__syncthreads();   // s_part is reused below
	FT_STAMP(p, 1);
	const uint32_t nterms = p.simple ? 0u : p.nterms;
	const uint32_t ns = nterms ? (p.n_subs < kFtRangeSubs ? p.n_subs : kFtRangeSubs) : 0u;
	// ---- the staged segments cut into blocks of 256 postings (a block never spans two sub-terms); the first kFtStageBlocks blocks are
	// fetched NOW into registers — one posting per thread and block, every load independent of the others — and the terms below run out
	// of registers.  Which sub-term a block belongs to is a wave-uniform fact: the term loop tests it with scalar compares.  (History:
	// walking the sub-terms one after the other, each behind its own load round trips, took 23 us of this kernel's 40 for a 3 x 3 query;
	// a flat layout with a per-posting owner search spent 6 us in VALU compares — a wave64 operation costs four cycles — plus an LDS stage.)
	if (ns) {
		const uint32_t a = 2 * tid, b = 2 * tid + 1;
		const uint32_t na = a < ns ? (s_hi[a] - s_lo[a] + 255) / 256 : 0u, nb = b < ns ? (s_hi[b] - s_lo[b] + 255) / 256 : 0u;
		const uint32_t incl = wave_inclusive_scan(na + nb, int(tid & 63));
		if ((tid & 63) == 63) s_part[tid >> 6] = incl;
		__syncthreads();
		uint32_t excl = incl - (na + nb);
		for (uint32_t w = 0; w < (tid >> 6); ++w) excl += s_part[w];
		if (a < ns) s_blk0[a] = excl;
		if (b < ns) s_blk0[b] = excl + na;
		if (b + 1 == ns) {   // the thread that holds the last staged sub-term also writes the end marker
			s_blk0[ns] = excl + na + nb;
		} else if (a + 1 == ns) {
			s_blk0[ns] = excl + na;
		}
		for (uint32_t which = 0; which < 2; ++which) {   // block descriptors of this thread's two sub-terms
			const uint32_t si = which ? b : a, nblk = which ? nb : na, b0 = which ? excl + na : excl;
			if (si >= ns) continue;
			for (uint32_t k = 0; k < nblk && b0 + k < kFtStageBlocks; ++k) {
				const uint32_t first = s_lo[si] + 256 * k;
				s_binfo[b0 + k] = BlockInfo{s_docptr[si] + first, s_hi[si] - first < 256u ? s_hi[si] - first : 256u, 0u};
			}
		}
		__syncthreads();
	}
	FT_STAMP(p, 7);
	const uint32_t total_blocks = ns ? s_blk0[ns] : 0u;
	const uint32_t n_staged = total_blocks < kFtStageBlocks ? total_blocks : kFtStageBlocks;
	uint32_t dd[kFtStageBlocks];   // local document id of the thread's posting in block q, or 0xFFFFFFFF
	{
		uint32_t raw[kFtStageBlocks];
		bool have[kFtStageBlocks];
#pragma unroll
		for (uint32_t q = 0; q < kFtStageBlocks; ++q) {
			const uint4 info = *reinterpret_cast<const uint4*>(&s_binfo[q]);   // uniform address: one broadcast read, all 32 of them independent
			const uint32_t* first = reinterpret_cast<const uint32_t*>((uint64_t(info.y) << 32) | info.x);
			have[q] = q < n_staged && tid < info.z;
			raw[q] = have[q] ? first[tid] : 0u;
		}
#pragma unroll
		for (uint32_t q = 0; q < kFtStageBlocks; ++q) dd[q] = have[q] ? raw[q] - uint32_t(d_begin) : 0xFFFFFFFFu;
	}
	FT_STAMP(p, 6);
	// the removed flags of the thread's documents (four per word), wanted after the terms: requested now, in flight behind the term pass
	constexpr uint32_t kSteps = kFtRangeDocs / 4 / 256;   // 8 steps of four documents per thread
	uint32_t rm8[kSteps];
#pragma unroll
	for (uint32_t j = 0; j < kSteps; ++j) {
		const uint32_t l0 = (j * 256 + tid) * 4;
		rm8[j] = (p.prescore && p.removed && l0 < docs_here) ? *reinterpret_cast<const uint32_t*>(p.removed + d_begin + l0) : 0u;   // reads past the end stay inside the allocation
	}
	for (uint32_t t = 0; t < nterms; ++t) {
		const FtTermCfg& term = p.terms[t];
		const int op = term.op;
		const bool want_score = p.prescore && op != 3;
		if (op == 1 && !want_score) continue;   // an OR term only matters to the pre-score
		for (uint32_t w = tid; w < kWords; w += 256) s_term[w] = 0;
		const uint32_t epoch = (t % 255u) + 1u;   // byte tag of this term (the bytes start at 0: set with the scores above)
		if (t && t % 255u == 0) {   // the tags wrap: no stale one may survive
			for (uint32_t w = tid; w < kFtRangeDocs / 4; w += 256) reinterpret_cast<uint32_t*>(s_scored)[w] = 0;
		}
		__syncthreads();
		// the term's configuration in registers: `term` lives in global memory, and a load per posting sat on the critical path
		const bool all_pos_boost = term.all_pos_boost != 0, same_boost = term.same_boost != 0;
		const float boost0 = term.field_boost[0], opts_boost = term.opts_boost;
		// a phrase part (its rows come from ft_phrase.hip): every document counts (GetMergedDocsBitmask, phrasemerger.h:309-317) and scores
		// CalcProc16 without the 65535 / 4 cap of a term (GetMergedDocsScore, :326-333)
		const bool is_phrase = term.phrase != 0;
		const uint32_t phrase16 = term.phrase_proc16;
		const bool need_entries = (op == 2 && !all_pos_boost) || (want_score && !same_boost);
		auto visit = [&](const FtPosSubterm& s, float sproc, uint32_t i, uint32_t local) {
			const uint32_t bit = 1u << (local & 31);
			if (op == 3) {
				atomicAnd(&s_mask[local >> 5], ~bit);
				return;
			}
			bool rel = all_pos_boost;
			float mb = boost0;
			if (need_entries) {   // maxFieldsBoost (phrasemergerimpl.h:127-160) / relevance of the occurrence
				mb = 0.0f;
				rel = false;
				for (uint32_t e = s.ent_off[i], e1 = s.ent_off[i + 1]; e < e1; ++e) {
					const float fb = term.field_boost[s.ent_field[e]];
					mb = fmaxf(mb, fb);
					rel = rel || fb != 0.0f;
				}
				if (same_boost) mb = boost0;
				if (all_pos_boost) rel = true;
			}
			if (op == 2 && rel) atomicOr(&s_term[local >> 5], bit);
			if (want_score && mb > 0.0f) {
				// termMask: documents are unique inside a sub-term and earlier sub-terms are behind a barrier, so the first one wins
				if (s_scored[local] != epoch) {
					s_scored[local] = uint8_t(epoch);
					const float proc = sproc * mb * opts_boost;
					uint32_t p16 = uint32_t(int32_t(proc)) & 0xFFFFu;   // static_cast<uint16_t>(float) as x86 evaluates it
					p16 = p16 < 65535u / 4 ? p16 : 65535u / 4;
					if (is_phrase) p16 = phrase16;
					const uint32_t cur = s_score[local];
					p16 = p16 < 65535u - cur ? p16 : 65535u - cur;
					s_score[local] = uint16_t(cur + p16);
				}
			}
		};
		// sub-terms in SortSubterms order behind barriers ("the first one holding the document wins")
		const uint32_t sub_begin = term.sub_begin, sub_end = term.sub_end;
		for (uint32_t si = sub_begin; si < sub_end; ++si) {
			const FtPosSubterm& s = p.subs[si];
			uint32_t lo, hi, from_global;
			float sproc;
			if (si < ns) {
				lo = s_lo[si];
				hi = s_hi[si];
				sproc = s_proc[si];
				const uint32_t b0 = s_blk0[si], b1raw = s_blk0[si + 1];
				const uint32_t b1 = b1raw < n_staged ? b1raw : n_staged;   // staged blocks of this sub-term: [b0, b1)
				const bool fast = !(op == 3 || need_entries);
				const float proc = sproc * boost0 * opts_boost;
				uint32_t p16c = uint32_t(int32_t(proc)) & 0xFFFFu;   // static_cast<uint16_t>(float) as x86 evaluates it
				p16c = p16c < 65535u / 4 ? p16c : 65535u / 4;
				if (is_phrase) p16c = phrase16;
				const bool scoring = want_score && boost0 > 0.0f;
#pragma unroll
				for (uint32_t g = 0; g < kFtStageBlocks; g += 4) {
					if (g + 4 <= b0 || g >= b1) continue;   // uniform: none of the four blocks is this sub-term's
					bool ok[4];
					uint32_t loc[4], tag[4];
#pragma unroll
					for (uint32_t q = 0; q < 4; ++q) {
						ok[q] = g + q >= b0 && g + q < b1 && dd[g + q] != 0xFFFFFFFFu;
						loc[q] = ok[q] ? dd[g + q] : 0u;
					}
					if (!fast) {
#pragma unroll
						for (uint32_t q = 0; q < 4; ++q) {
							if (ok[q]) visit(s, sproc, lo + 256 * (g + q - b0) + tid, loc[q]);
						}
						continue;
					}
					// the common case (no per-entry field test), four postings per thread: their LDS atomics and score updates are
					// independent chains that overlap
					if (op == 2) {   // all_pos_boost: every occurrence is relevant
#pragma unroll
						for (uint32_t q = 0; q < 4; ++q) {
							if (ok[q]) atomicOr(&s_term[loc[q] >> 5], 1u << (loc[q] & 31));
						}
					}
					if (scoring) {
						// Reads without a condition (a lane without a posting reads document 0 and drops the result): under `if (ok)` each
						// of them became its own exec-masked block with its own lgkmcnt(0), i.e. eight LDS round trips in a row per
						// group instead of two.  The four documents of a thread are distinct, and so are those of different threads
						// (one sub-term), so every read may precede every write.
						uint32_t cur[4];
#pragma unroll
						for (uint32_t q = 0; q < 4; ++q) tag[q] = s_scored[loc[q]];
#pragma unroll
						for (uint32_t q = 0; q < 4; ++q) cur[q] = s_score[loc[q]];
#pragma unroll
						for (uint32_t q = 0; q < 4; ++q) {
							if (!ok[q] || tag[q] == epoch) continue;   // (an earlier sub-term of the term holds the document)
							const uint32_t add = p16c < 65535u - cur[q] ? p16c : 65535u - cur[q];
							s_scored[loc[q]] = uint8_t(epoch);
							s_score[loc[q]] = uint16_t(cur[q] + add);
						}
					}
				}
				from_global = lo + 256 * (b1 > b0 ? b1 - b0 : 0u);   // what did not fit the stage
				if (from_global > hi) from_global = hi;
			} else {
				lo = range < s.n_ranges ? s.range_off[range] : uint32_t(s.n);
				hi = range + 1 < s.n_ranges ? s.range_off[range + 1] : uint32_t(s.n);
				sproc = s.proc;
				from_global = lo;
			}
			for (uint32_t base = from_global; base < hi; base += 4 * 256) {
				uint32_t idx[4], d4[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					idx[q] = base + uint32_t(q) * 256 + tid;
					d4[q] = idx[q] < hi ? s.doc[idx[q]] : 0u;
				}
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					if (idx[q] < hi) visit(s, sproc, idx[q], uint32_t(d4[q] - d_begin));
				}
			}
			__syncthreads();   // the next sub-term of the term sees this one's documents
		}
		if (op == 2) {   // restrictingMask_ &= termMask (an AND term without postings empties the range)
			const uint32_t* syn_mask = term.syn_mask;   // termMask |= synMask of the part's multi-word synonyms (mergerimpl.h:347-361), ft_syn_masks
			for (uint32_t w = tid; w < kWords; w += 256) {
				const uint64_t gw = d_begin / 32 + w;
				s_mask[w] &= s_term[w] | ((syn_mask && gw < p.nwords) ? syn_mask[gw] : 0u);
			}
			__syncthreads();
		}
		FT_STAMP(p, 11 + (t < 4 ? t : 4));
	}
	FT_STAMP(p, 2);
	// ---- the range's mask words + their popcount (the device half of the 2-phase gate).  The global stores come last: a barrier behind
	// them would wait for their acknowledgement
	uint32_t c = 0;
	for (uint32_t w = tid; w < kWords; w += 256) {
		if (d_begin / 32 + w < p.nwords) c += __popc(s_mask[w]);
	}
	c = wave_sum(c);
	if ((tid & 63) == 0) s_part[tid >> 6] = c;
	__syncthreads();
	if (tid == 0) {
		const uint32_t tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
		if (tot) atomicAdd(&p.sync[kFtSyncPop], tot);
	}
	for (uint32_t w = tid; w < kWords; w += 256) {
		const uint64_t gw = d_begin / 32 + w;
		if (gw < p.nwords) p.mask[gw] = s_mask[w];
	}
	FT_STAMP(p, 3);
	if (!p.prescore) return;
	// ---- scores: masked-out / removed documents score 0 (mergerimpl.h:416-423); histogram of the rest through a small LDS hash table.
	// Few distinct scores occur (a handful of proc values and their sums), so plain counters would take 64-way same-address atomics from
	// every wavefront: each key gets kFtHistRep counters on different banks (the posting stage is free by now), lane l adds to counter l % kFtHistRep
	uint32_t* hist_copy = p.hist + size_t(range % kFtHistCopies) * kFtHistStride;
	for (uint32_t i = tid; i < 256 * kFtHistRep; i += 256) s_rep[i] = 0;
	__syncthreads();
	// masked scores first (eight 8-byte LDS reads, the global stores; the masked value goes back into the LDS copy), then the counting
	// loop — deliberately NOT unrolled: with the probe loop inlined 32 times this phase was 8400 instructions and, at three wavefronts
	// per SIMD, took 6 us on instruction issue alone
#pragma unroll
	for (uint32_t j = 0; j < kSteps; ++j) {
		const uint32_t l0 = (j * 256 + tid) * 4;
		const uint32_t mw = s_mask[(l0 >> 5) & (kWords - 1)];
		const uint2 four = *reinterpret_cast<const uint2*>(&s_score[l0]);
		const uint32_t raw[4] = {four.x & 0xFFFFu, four.x >> 16, four.y & 0xFFFFu, four.y >> 16};
		uint32_t sc[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool in = l0 + k < docs_here && ((mw >> ((l0 + k) & 31)) & 1u) && !((rm8[j] >> (8 * k)) & 0xFFu);
			sc[k] = in ? raw[k] : 0u;
		}
		const uint2 packed = make_uint2(sc[0] | (sc[1] << 16), sc[2] | (sc[3] << 16));
		*reinterpret_cast<uint2*>(&s_score[l0]) = packed;   // this thread's own four documents
		if (l0 < docs_here) *reinterpret_cast<uint2*>(p.score + d_begin + l0) = packed;   // the array is padded to whole mask words
	}
#pragma unroll 1
	for (uint32_t j = 0; j < kSteps; ++j) {
		const uint32_t l0 = (j * 256 + tid) * 4;
		const uint2 four = *reinterpret_cast<const uint2*>(&s_score[l0]);
		const uint32_t sc[4] = {four.x & 0xFFFFu, four.x >> 16, four.y & 0xFFFFu, four.y >> 16};
		uint32_t h[4], cur[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			h[k] = (sc[k] * 2654435761u) >> 24;
			cur[k] = sc[k] ? s_keys[h[k]] : 0u;
		}
		uint32_t slow = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (!sc[k]) continue;
			if (cur[k] == sc[k]) {   // the key is there already (after the first few documents of the range it always is)
				atomicAdd(&s_rep[h[k] * kFtHistRep + (tid & (kFtHistRep - 1))], 1u);
			} else {
				slow |= 1u << k;
			}
		}
		while (slow) {   // a new key, or a slot taken by another one: probe
			const int k = __ffs(int(slow)) - 1;
			slow &= slow - 1;
			const uint32_t v = k == 0 ? sc[0] : k == 1 ? sc[1] : k == 2 ? sc[2] : sc[3];
			uint32_t hh = k == 0 ? h[0] : k == 1 ? h[1] : k == 2 ? h[2] : h[3];
			int probes = 0;
			for (; probes < 256; ++probes, hh = (hh + 1) & 255u) {
				uint32_t c = s_keys[hh];
				if (c != v) {
					if (c != 0u) continue;
					c = atomicCAS(&s_keys[hh], 0u, v);
					if (c != 0u && c != v) continue;
				}
				atomicAdd(&s_rep[hh * kFtHistRep + (tid & (kFtHistRep - 1))], 1u);
				break;
			}
			if (probes == 256) {   // more than 256 distinct scores in one range (a register-side count of the common values was slower: 15 us)
				atomicAdd(&hist_copy[v], 1u);
				atomicAdd(&hist_copy[65536 + (v >> 6)], 1u);
			}
		}
	}
	__syncthreads();
	FT_STAMP(p, 4);
	const uint32_t key = s_keys[tid];
	uint32_t total = 0;
	if (key) {
#pragma unroll
		for (uint32_t r = 0; r < kFtHistRep; ++r) total += s_rep[tid * kFtHistRep + r];
		atomicAdd(&hist_copy[key], total);
	}
	// the chunk totals: the keys of one chunk are combined inside the workgroup first — every workgroup holds the same few scores, and
	// same-address device atomics from 600 workgroups on two or three chunk counters took 24 us when each key added on its own
	__syncthreads();   // the replicated counters have been read: their space becomes the chunk table
	uint32_t* s_ck = s_rep;          // [256] chunk + 1 (0 = free)
	uint32_t* s_cc = s_rep + 256;    // [256] documents
	s_ck[tid] = 0;
	s_cc[tid] = 0;
	__syncthreads();
	if (key) {
		const uint32_t ck = (key >> 6) + 1;
		uint32_t h = (ck * 2654435761u) >> 24;
		for (int probes = 0; probes < 256; ++probes, h = (h + 1) & 255u) {   // at most 256 keys: a free slot always turns up
			uint32_t cur = s_ck[h];
			if (cur != ck) {
				if (cur != 0u) continue;
				cur = atomicCAS(&s_ck[h], 0u, ck);
				if (cur != 0u && cur != ck) continue;
			}
			atomicAdd(&s_cc[h], total);
			break;
		}
	}
	__syncthreads();
	if (s_ck[tid]) atomicAdd(&hist_copy[65536 + s_ck[tid] - 1], s_cc[tid]);
	FT_STAMP(p, 5);
}


// mergerimpl.h:448-462: kFtApplyWords mask words per thread (the ordered prefix chain is as long as the grid: fewer, fatter workgroups);
// ties at minScore are kept in document order up to minScoreDocs
__global__ __launch_bounds__(256) void ft_preselect_apply(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (!ft_preselect_on(p)) return;
	const uint32_t ticket = grab_ticket(p.sync + kFtSyncPreTicket);
	// a document-range shard walks its own mask words (tickets = document order inside the shard); nwords_end bounds the last workgroup
	const uint64_t word_begin = uint64_t(p.range_begin) * (kFtRangeDocs / 32);
	const uint64_t nwords_end = p.range_count ? min(p.nwords, word_begin + uint64_t(p.range_count) * (kFtRangeDocs / 32)) : p.nwords;
	const uint64_t w0 = word_begin + (uint64_t(ticket) * 256 + threadIdx.x) * kFtApplyWords;
	uint32_t min_score, min_docs;
	ft_pick_threshold(p, &min_score, &min_docs);
	if (p.shard_hist) {   // the ties the shards in front of this one keep at the threshold score (document order = shard order): taken off the quota
		uint32_t used = 0;
		for (uint32_t sh = 0; sh < p.shard_index; ++sh) used += p.shard_hist[size_t(p.shard_pos[sh]) * kFtFoldWords + min_score];
		min_docs = min_docs > used ? min_docs - used : 0u;
	}
	uint32_t bits[kFtApplyWords], gt[kFtApplyWords], tie[kFtApplyWords];
	uint32_t ties = 0;
#pragma unroll
	for (int j = 0; j < kFtApplyWords; ++j) {
		bits[j] = gt[j] = tie[j] = 0;
		const uint64_t w = w0 + j;
		if (w >= nwords_end) continue;
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
		if (w0 + j >= nwords_end) continue;
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
__global__ __launch_bounds__(256) void ft_rank_all(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (blockIdx.x * kFtRankTiles >= p.merge_blocks) return;   // the grid is the widest query's
	__shared__ uint32_t s_cnt;
	__shared__ uint16_t s_item[kFtRankTiles * kFtBlockPostings];   // tile << 10 | thread << 2 | slot
	__shared__ uint32_t s_doc[kFtRankTiles * kFtBlockPostings];
	__shared__ uint32_t s_sub[kFtRankTiles], s_base[kFtRankTiles];
	// the tiles' sub-term and term descriptors, copied once: calcTermRank reads a dozen of their fields per posting, and from global
	// memory every first touch was one more round trip in front of the entry gathers
	__shared__ FtPosSubterm s_subd[kFtRankTiles];
	__shared__ FtTermCfg s_termd[kFtRankTiles];
	static_assert(sizeof(FtPosSubterm) % 4 == 0 && sizeof(FtTermCfg) % 4 == 0, "descriptors are copied word by word");
	FT_STAMP(p, 16);
	if (threadIdx.x == 0) s_cnt = 0;
	{
		const uint32_t g = threadIdx.x >> 7, l = threadIdx.x & 127;   // 128 threads per tile: binary search by all (broadcast loads), copy by words
		static_assert(kFtRankTiles == 2, "two tiles, 128 threads each");
		const uint32_t tile = blockIdx.x * kFtRankTiles + g;
		if (tile < p.merge_blocks) {
			const FtGridEntry ge = grid_entry(p.merge_grid, p.n_merge_entries, tile);
			if (l == 0) {
				s_sub[g] = ge.sub;
				s_base[g] = ge.block_base;
			}
			const uint32_t* src_s = reinterpret_cast<const uint32_t*>(p.subs + ge.sub);
			if (l < sizeof(FtPosSubterm) / 4) reinterpret_cast<uint32_t*>(&s_subd[g])[l] = src_s[l];
			const uint32_t term = p.subs[ge.sub].term;
			const uint32_t* src_t = reinterpret_cast<const uint32_t*>(p.terms + term);
			if (l < sizeof(FtTermCfg) / 4) reinterpret_cast<uint32_t*>(&s_termd[g])[l] = src_t[l];
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
		const FtPosSubterm& s = s_subd[g];
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
	FT_STAMP(p, 17);
	const uint32_t cnt = s_cnt;
	const int lane = threadIdx.x & 63;
	// Two postings per thread and step when more than one step is due (dense merges rank every posting of the tile): the two
	// calcTermRank gather chains are independent and overlap.  Uniform trip count: the wavefront votes below.
	const uint32_t per_step = cnt > 256 ? 512u : 256u;
	for (uint32_t e0 = 0; e0 < cnt; e0 += per_step) {
		bool want[2] = {false, false};
		uint32_t d[2] = {0, 0}, row[2] = {0, 0}, idx[2] = {0, 0};
		float rank[2] = {0.0f, 0.0f};
		uint8_t field[2] = {0, 0};
#pragma unroll
		for (int u = 0; u < 2; ++u) {
			const uint32_t e = e0 + uint32_t(u) * 256 + threadIdx.x;
			if ((u == 0 || per_step == 512) && e < cnt) {
				const uint32_t item = s_item[e], g = item >> 10, local = item & 1023u;
				const uint32_t tile = blockIdx.x * kFtRankTiles + g;
				const FtPosSubterm& s = s_subd[g];
				const FtTermCfg& t = s_termd[g];
				const uint64_t i = uint64_t(tile - s_base[g]) * kFtBlockPostings + local;
				d[u] = s_doc[e];
				if (s.suppressed) {   // mergerimpl.h:144-151: no rank; the record only counts the term for a document that is merged already
					rank[u] = __uint_as_float(kFtSuppressedRank);
					field[u] = 0;
				} else if (s.pre_rank) {   // a phrase row: mergePhrase takes the PhraseMerger's rank and field as they are (mergerimpl.h:46-62)
					rank[u] = s.pre_rank[i];
					field[u] = s.pre_field[i];
				} else {
					rank[u] = ft_term_rank(t, s, s.ent_off[i], s.ent_off[i + 1], d[u], &field[u]);
				}
				want[u] = rank[u] != 0.0f;
				row[u] = s.row;
				idx[u] = uint32_t(i);
			}
		}
		// one device atomic per (wavefront, document range): neighbouring postings share a range, whole-corpus merges would otherwise
		// hammer a few hundred counters with millions of same-address atomics
#pragma unroll
		for (int u = 0; u < 2; ++u) {
			const uint32_t rg = d[u] >> kFtRangeShift;
			unsigned long long pending = __ballot(want[u]);
			while (pending) {
				const int leader = __ffsll((long long)pending) - 1;
				const uint32_t lrg = uint32_t(__shfl(int(rg), leader, 64));
				const bool mine = want[u] && rg == lrg;
				const unsigned long long same = __ballot(mine);
				uint32_t base = 0;
				if (lane == leader) base = atomicAdd(&p.bucket_cnt[lrg], uint32_t(__popcll(same)));
				base = uint32_t(__shfl(int(base), leader, 64));
				if (mine) {
					const uint32_t pos = base + uint32_t(__popcll(same & ((1ull << lane) - 1ull)));
					p.b_rec[uint64_t(p.bucket_off[lrg]) + pos] = make_uint4(d[u], idx[u], __float_as_uint(rank[u]), row[u] | (uint32_t(field[u]) << 16));
					want[u] = false;
				}
				pending &= ~same;
			}
		}
	}
	FT_STAMP(p, 18);
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
__global__ __launch_bounds__(256) void ft_adders(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	__shared__ uint32_t s_tab[kFtRangeDocs / 2];   // first row of every document of the range, 16 bits each
	__shared__ uint32_t s_rowcnt[kFtAdderRowsLds];
	const uint32_t tid = threadIdx.x, range = blockIdx.x + p.range_begin;
	FT_STAMP(p, 24);
	if (p.prescore) {   // ft_preselect_apply was the last reader of the histograms and of the look-back words
		const uint64_t gtid = uint64_t(blockIdx.x) * blockDim.x + tid, gsize = uint64_t(gridDim.x) * blockDim.x;
		fill_words(p.hist, uint64_t(kFtHistCopies) * kFtHistStride, 0u, gtid, gsize);
		fill_words(reinterpret_cast<uint32_t*>(p.lookback_pre), ((p.nwords + 256 * kFtApplyWords - 1) / (256 * kFtApplyWords)) * 2, 0u, gtid, gsize);
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
	FT_STAMP(p, 25);
	const uint4* rec = p.b_rec + p.bucket_off[range];
	for (uint32_t e = tid; e < n; e += 256) {
		const uint4 r = rec[e];
		if (r.z == kFtSuppressedRank) continue;   // a suppressed sub-term never adds a document
		lds_min_u16(s_tab, r.x & (kFtRangeDocs - 1), r.w & 0xFFFFu);
	}
	__syncthreads();
	FT_STAMP(p, 26);
	for (uint32_t e = tid; e < n; e += 256) {
		const uint4 r = rec[e];
		const uint32_t row = r.w & 0xFFFFu;
		if (r.z == kFtSuppressedRank) continue;
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
	FT_STAMP(p, 27);
}

// The merge slot of the first document of (row, range) = the entries of ft_adders' table in front of it, in row-major order.  Up to
// kFtFinishRows merged sub-terms (and 8192 entries) the table is small and every workgroup of ft_finish adds up its own bases (one pass, all
// rows at once); larger queries run ft_slot_bases, which turns the table into its prefix.
constexpr uint32_t kFtFinishRows = 16;
constexpr uint32_t kFtSparseRecords = 768;   // buckets up to this size are replayed from an LDS copy of their records
constexpr uint32_t kFtSparsePostings = 6;   // ... if no document of theirs has more postings than this
constexpr uint32_t kFtBitmapRows = 8;   // up to this many merged sub-terms ft_finish ranks the first postings with per-row bitmaps instead of a sort
template <typename Plan>
__host__ __device__ inline bool ft_own_bases(const Plan& p) { return p.n_rows <= kFtFinishRows && uint64_t(p.n_rows) * p.n_ranges <= kFtRangeDocs; }

// The table of ft_adders -> its exclusive prefix in row-major order = the merge slot of the first document of every (row, range), and the
// number of merged documents.  One workgroup, behind a kernel boundary: an in-kernel hand-over (every workgroup releasing at agent scope
// before an arrival counter) cost 29 us — each release writes back the dirty lines of its XCD's L2.
// Tiles of 2048 entries: eight consecutive loads per thread in flight, one workgroup scan, eight stores.
__global__ __launch_bounds__(256) void ft_slot_bases(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (!p.sparse && ft_own_bases(p)) return;   // this query's ft_finish sums the small table itself (the sparse train always comes here)
	__shared__ uint32_t s_part[4];
	const uint32_t tid = threadIdx.x;
	const uint64_t total = uint64_t(p.n_rows) * p.n_ranges;
	uint32_t carry = 0;
	for (uint64_t base = 0; base < total; base += 256 * 8) {
		const uint64_t j0 = base + uint64_t(tid) * 8;
		uint32_t v[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = j0 + k < total ? p.adders[j0 + k] : 0u;
		uint32_t local = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k) local += v[k];
		const uint32_t incl = wave_inclusive_scan(local, int(tid & 63));
		if ((tid & 63) == 63) s_part[tid >> 6] = incl;
		__syncthreads();
		uint32_t running = carry + incl - local;
		for (uint32_t w = 0; w < (tid >> 6); ++w) running += s_part[w];
		carry += s_part[0] + s_part[1] + s_part[2] + s_part[3];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			if (j0 + k < total) p.adders[j0 + k] = running;
			running += v[k];
		}
		__syncthreads();   // s_part is rewritten by the next tile
	}
	if (tid == 0) p.sync[kFtSyncNumDocs] = carry < p.max_merged ? carry : p.max_merged;
}

__device__ __forceinline__ void ft_replay_step(FtPlanK& p, FtReplayState& st, uint32_t row, float r, uint8_t fld, uint32_t i, const uint64_t* const* s_fpos,
											   const uint32_t* const* s_pos_off, const uint32_t* s_qp, uint32_t sl, float& max_term_rank) {
	uint32_t qp = 0;
	FtPosList pos;
	if (!p.simple || p.max_areas) ft_replay_locate(p, row, i, s_fpos, s_pos_off, s_qp, qp, pos.ptr, pos.n);
	if (p.max_areas && __float_as_uint(r) != kFtSuppressedRank) ft_areas_of_posting(p, sl, pos.ptr, pos.n, r, max_term_rank);
	ft_replay_apply(p, st, r, fld, qp, pos);
}
// the document's row of the entry table, walked in sub-term order
__device__ __forceinline__ void ft_replay_doc(FtPlanK& p, uint32_t sl, uint32_t doc, const uint64_t* const* s_fpos, const uint32_t* const* s_pos_off,
											  const uint32_t* s_qp) {
	FtReplayState st;
	float max_term_rank = 0.f;   // AreasInDocument::maxTermRank_
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
			ft_replay_step(p, st, row, p.e_rank[cell], p.e_field[cell], (p.simple && !p.max_areas) ? 0u : p.e_idx[cell], s_fpos, s_pos_off, s_qp, sl, max_term_rank);
		}
	}
	ft_replay_finish(p, st, sl, doc);
}

// Slots, entry rows and the per-document replay of one document range.  Dynamic LDS: the 16-bit document table (first row, then the
// position in the sorted key list) followed by the key list of the range's first postings ((row << 13 | document), then the slot).
__global__ __launch_bounds__(256) void ft_finish(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	extern __shared__ uint32_t ft_finish_lds[];
	uint32_t* s_tab = ft_finish_lds;                       // [kFtRangeDocs / 2]
	uint32_t* s_keys = ft_finish_lds + kFtRangeDocs / 2;   // [kFtRangeDocs]
	__shared__ const uint64_t* s_fpos[kFtReplayRows];
	__shared__ const uint32_t* s_pos_off[kFtReplayRows];
	__shared__ uint32_t s_qp[kFtReplayRows];   // ft_row_qpw
	__shared__ uint32_t s_nadd, s_last;
	__shared__ uint32_t s_red[4][kFtFinishRows], s_rowbase[kFtFinishRows];
	const uint32_t tid = threadIdx.x, range = blockIdx.x + p.range_begin;
	const bool own_bases = ft_own_bases(p);   // the table is small: every workgroup sums what lies in front of its own entries
	FT_STAMP(p, 32);
	// everything the workgroup fetches unconditionally is issued together: record count and bucket offset, the replay descriptors, and
	// (own_bases) the whole table of ft_adders — pulled into the key area of the LDS, which is idle until the first postings are keyed.
	// base(row) = every entry of the rows before it + this row's entries left of the range: one wavefront per row adds up (row total,
	// part left of the range).  (Loops that consumed each load before issuing the next paid a memory round trip per row or per column
	// chunk: 7-10 us; testing every entry against every row's position in registers: 9 us of VALU work.)
	const uint32_t n = p.bucket_cnt[range];
	const uint32_t bucket_off = p.bucket_off[range];
	uint32_t tv[kFtRangeDocs / 256];
	if (own_bases) {
		const uint32_t total = p.n_rows * p.n_ranges;
#pragma unroll
		for (uint32_t k = 0; k < kFtRangeDocs / 256; ++k) tv[k] = k * 256 + tid < total ? p.adders[k * 256 + tid] : 0u;
	}
	for (uint32_t row = tid; row < p.n_rows && row < kFtReplayRows; row += 256) {   // two dependent loads per row, once per workgroup
		const FtPosSubterm& s = p.subs[p.merge_grid[row].sub];
		s_fpos[row] = s.fpos;
		s_pos_off[row] = s.pos_off;
		s_qp[row] = ft_row_qpw(s);
	}
	// up to kFtBitmapRows merged sub-terms (and mergeLimit below 65535): the compact layout — 16-bit slots, 32 KB of LDS, 4 workgroups per CU
	const bool few_rows = p.n_rows <= kFtBitmapRows && p.max_merged < 0xFFFFu;
	uint32_t* s_table = few_rows ? ft_finish_lds : s_keys;   // ft_adders' table while the bases are summed (the slots are set up afterwards)
	if (n) {
		if (tid == 0) s_nadd = 0;
		if (own_bases) {
			const uint32_t total = p.n_rows * p.n_ranges;
#pragma unroll
			for (uint32_t k = 0; k < kFtRangeDocs / 256; ++k) {
				if (k * 256 + tid < total) s_table[k * 256 + tid] = tv[k];
			}
		}
		__syncthreads();
		if (own_bases) {
			const uint32_t lane = tid & 63;
			for (uint32_t r = tid >> 6; r < p.n_rows; r += 4) {
				uint32_t rs = 0, ps = 0;
				for (uint32_t c = lane; c < p.n_ranges; c += 64) {
					const uint32_t x = s_table[r * p.n_ranges + c];
					rs += x;
					ps += c < range ? x : 0u;
				}
				rs = wave_sum(rs);
				ps = wave_sum(ps);
				if (lane == 0) {
					s_red[0][r] = rs;
					s_red[1][r] = ps;
				}
			}
		}
		__syncthreads();
		if (own_bases && tid == 0) {
			uint32_t before_rows = 0;
			for (uint32_t r = 0; r < p.n_rows; ++r) {
				s_rowbase[r] = before_rows + s_red[1][r];
				before_rows += s_red[0][r];
			}
		}
		__syncthreads();
		// the mergeLimit cut (addDoc until maxMergedDocs): slots ascend with (row, range), so when even the first slot of every row of this
		// range lies at or beyond the limit none of its documents is merged — a whole-corpus single-term merge adds its 20 000 documents in
		// the first dozen ranges, and the other six hundred have nothing to sort, scatter or replay
		bool reaches = false;
		for (uint32_t r = tid; r < p.n_rows; r += 256) {
			reaches = reaches || (own_bases ? s_rowbase[r] : p.adders[uint64_t(r) * p.n_ranges + range]) < p.max_merged;
		}
		const uint32_t nw = __syncthreads_or(reaches ? 1 : 0) ? n : 0u;
		uint4* rec = p.b_rec + bucket_off;
		const uint32_t d_begin = range << kFtRangeShift;
		if (few_rows) {
			// ---- up to kFtBitmapRows merged sub-terms: no sort.  s_slot[doc] = first row, then the slot (16 bits: max_merged < 65535, a slot at
			// or beyond it means "never merged" and is stored as 0xFFFF); one bitmap of the range's documents per row marks the first
			// postings, a prefix over the bitmap words gives every one its rank inside (row, range)
			uint32_t* s_slot = ft_finish_lds;                                  // [kFtRangeDocs] halves
			uint32_t* s_bits = ft_finish_lds + kFtRangeDocs / 2;               // [kFtBitmapRows][256]
			uint32_t* s_pref = s_bits + kFtBitmapRows * (kFtRangeDocs / 32);   // [kFtBitmapRows][256]
			// (the table of ft_adders lay here: dead since the barrier behind the row bases)
			for (uint32_t w = tid; w < kFtRangeDocs / 2; w += 256) s_slot[w] = 0xFFFFFFFFu;
			for (uint32_t w = tid; w < kFtBitmapRows * (kFtRangeDocs / 32); w += 256) s_bits[w] = 0;
			__syncthreads();
			FT_STAMP(p, 33);
			for (uint32_t e = tid; e < nw; e += 256) {
				const uint4 r = rec[e];
				if (r.z != kFtSuppressedRank) lds_min_u16(s_slot, r.x & (kFtRangeDocs - 1), r.w & 0xFFFFu);
			}
			__syncthreads();
			FT_STAMP(p, 34);
			for (uint32_t e = tid; e < nw; e += 256) {   // the range's first postings; the record remembers that it adds its document
				const uint4 r = rec[e];
				const uint32_t row = r.w & 0xFFFFu, dl = r.x & (kFtRangeDocs - 1);
				if (lds_get_u16(s_slot, dl) != row || r.z == kFtSuppressedRank) continue;
				atomicOr(&s_bits[row * (kFtRangeDocs / 32) + (dl >> 5)], 1u << (dl & 31));
				rec[e].w = r.w | 0x80000000u;
			}
			__syncthreads();
			FT_STAMP(p, 35);
			{   // exclusive prefix of the popcounts along every bitmap: thread t owns word t
				uint32_t cnt[kFtBitmapRows], incl[kFtBitmapRows];
#pragma unroll
				for (uint32_t r = 0; r < kFtBitmapRows; ++r) {
					cnt[r] = r < p.n_rows ? __popc(s_bits[r * (kFtRangeDocs / 32) + tid]) : 0u;
					incl[r] = wave_inclusive_scan(cnt[r], int(tid & 63));
					if ((tid & 63) == 63) s_red[tid >> 6][r] = incl[r];
				}
				__syncthreads();
#pragma unroll
				for (uint32_t r = 0; r < kFtBitmapRows; ++r) {
					uint32_t excl = incl[r] - cnt[r];
					for (uint32_t w = 0; w < (tid >> 6); ++w) excl += s_red[w][r];
					s_pref[r * (kFtRangeDocs / 32) + tid] = excl;
				}
			}
			__syncthreads();
			FT_STAMP(p, 36);
			for (uint32_t e = tid; e < nw; e += 256) {   // first posting -> slot of its document
				const uint4 r = rec[e];
				if (!(r.w >> 31)) continue;
				const uint32_t row = r.w & 0xFFFFu, dl = r.x & (kFtRangeDocs - 1);
				const uint32_t word = row * (kFtRangeDocs / 32) + (dl >> 5);
				const uint32_t rank = s_pref[word] + __popc(s_bits[word] & ((1u << (dl & 31)) - 1u));
				const uint32_t slot = (own_bases ? s_rowbase[row] : p.adders[uint64_t(row) * p.n_ranges + range]) + rank;
				lds_set_u16(s_slot, dl, slot < 0xFFFFu ? slot : 0xFFFFu);   // every document with an unsuppressed record has exactly one first posting (the others keep 0xFFFF: never merged)
				if (slot < p.max_merged) p.out_doc[slot] = d_begin + dl;
			}
			__syncthreads();
			FT_STAMP(p, 37);
		} else {
			// ---- many sub-terms: 16-bit document table + a sorted list of (row, document) keys
			for (uint32_t w = tid; w < kFtRangeDocs / 2; w += 256) s_tab[w] = 0xFFFFFFFFu;
			__syncthreads();
			FT_STAMP(p, 33);
			for (uint32_t e = tid; e < nw; e += 256) {
				const uint4 r = rec[e];
				if (r.z != kFtSuppressedRank) lds_min_u16(s_tab, r.x & (kFtRangeDocs - 1), r.w & 0xFFFFu);
			}
			__syncthreads();
			FT_STAMP(p, 34);
			for (uint32_t e = tid; e < nw; e += 256) {   // the range's first postings: key (row, document); the record remembers that it adds
				const uint4 r = rec[e];
				const uint32_t row = r.w & 0xFFFFu, dl = r.x & (kFtRangeDocs - 1);
				if (lds_get_u16(s_tab, dl) != row || r.z == kFtSuppressedRank) continue;
				s_keys[atomicAdd(&s_nadd, 1u)] = (row << kFtRangeShift) | dl;
				rec[e].w = r.w | 0x80000000u;
			}
			__syncthreads();
			FT_STAMP(p, 35);
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
			FT_STAMP(p, 36);
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
			FT_STAMP(p, 37);
			for (uint32_t q = tid; q < A; q += 256) {   // key -> slot; document -> its position in the list
				const uint32_t key = s_keys[q], row = key >> kFtRangeShift, dl = key & (kFtRangeDocs - 1);
				const uint32_t slot = (own_bases ? s_rowbase[row] : p.adders[uint64_t(row) * p.n_ranges + range]) + lds_get_u16(s_tab, dl);
				s_keys[q] = slot;
				lds_set_u16(s_tab, dl, q);
				if (slot < p.max_merged) p.out_doc[slot] = d_begin + dl;
			}
			__syncthreads();
		}
		// (a document whose only records are a suppressed sub-term's has no first posting: slot 0xFFFFFFFF = never merged)
		auto slot_of = [&](uint32_t dl) -> uint32_t {
			if (few_rows) {
				const uint32_t v = lds_get_u16(ft_finish_lds, dl);
				return v == 0xFFFFu ? 0xFFFFFFFFu : v;
			}
			const uint32_t at = lds_get_u16(s_tab, dl);
			return at == 0xFFFFu ? 0xFFFFFFFFu : s_keys[at];
		};
		// ---- a sparse bucket (the usual one after a preselect: a few hundred records): the records go into LDS and the thread of every
		// first posting collects its document's other postings from there — no entry rows in global memory, i.e. no scatter, no gather
		// chain in front of the position loads, nothing to hand back zeroed.  A document with more than kFtSparsePostings postings
		// sends the whole range down the general path.
		bool sparse = nw <= kFtSparseRecords;
		if (sparse) {
			// behind the slots of the bitmap layout; in the sorted layout the list of slots is no longer than the bucket, i.e. ends in front
			uint32_t* sparse_base = ft_finish_lds + (few_rows ? kFtRangeDocs / 2 : kFtRangeDocs);   // behind the 16-bit slots / inside the idle tail of the key list
			uint4* s_rec = reinterpret_cast<uint4*>(sparse_base);   // [kFtSparseRecords]; .x = document in the range | next record << 13
			uint32_t* s_head = sparse_base + kFtSparseRecords * 4;   // [kHeads] first record of the documents with these low bits
			constexpr uint32_t kHeads = 512, kNil = 1023;
			uint16_t* s_todo = reinterpret_cast<uint16_t*>(s_head + kHeads);   // [kFtSparseRecords] the first postings of the documents to replay
			static_assert(kFtSparseRecords <= kNil && kFtSparseRecords * 4 + kHeads + kFtSparseRecords / 2 <= kFtRangeDocs / 2, "LDS layout of the sparse replay");
			for (uint32_t w = tid; w < kHeads; w += 256) s_head[w] = kNil;
			if (tid == 0) s_nadd = 0;
			__syncthreads();
			for (uint32_t e = tid; e < nw; e += 256) {   // the documents' records chained through the copy; one thread per document to replay
				uint4 r = rec[e];
				const uint32_t dl = r.x & (kFtRangeDocs - 1);
				r.x = dl | (atomicExch(&s_head[dl & (kHeads - 1)], e) << kFtRangeShift);
				s_rec[e] = r;
				if ((r.w >> 31) && slot_of(dl) < p.max_merged) s_todo[atomicAdd(&s_nadd, 1u)] = uint16_t(e);
			}
			__syncthreads();
			FT_STAMP(p, 38);
			const uint32_t todo = s_nadd;
			const FtTermCfg& t0 = p.terms[0];
			const bool one_field = t0.num_fields == 1;
			int over = 0;
			for (uint32_t q = tid; q < todo; q += 256) {
				const uint32_t dl = s_rec[s_todo[q]].x & (kFtRangeDocs - 1);
				float words0 = 0.f;
				if (one_field) words0 = t0.words[d_begin + dl];   // in flight during the walk
				// the document's postings: keys (sub-term row, record) collected along the chain, ordered by a sorting network
				uint32_t key[kFtSparsePostings];
#pragma unroll
				for (uint32_t k = 0; k < kFtSparsePostings; ++k) key[k] = 0xFFFFFFFFu;
				uint32_t cnt = 0;
				for (uint32_t j = s_head[dl & (kHeads - 1)]; j != kNil;) {
					const uint4 o = s_rec[j];
					if ((o.x & (kFtRangeDocs - 1)) == dl) {
						const uint32_t v = ((o.w & 0xFFFFu) << 16) | j;
#pragma unroll
						for (uint32_t k = 0; k < kFtSparsePostings; ++k) key[k] = cnt == k ? v : key[k];
						++cnt;
					}
					j = o.x >> kFtRangeShift;
				}
				if (cnt > kFtSparsePostings) {
					over = 1;
					continue;
				}
#define RX_CSWAP(a, b)                                \
	{                                                 \
		const uint32_t lo = key[a] < key[b] ? key[a] : key[b]; \
		const uint32_t hi = key[a] < key[b] ? key[b] : key[a]; \
		key[a] = lo;                                  \
		key[b] = hi;                                  \
	}
				static_assert(kFtSparsePostings == 6, "the network below orders six keys");
				RX_CSWAP(0, 5) RX_CSWAP(1, 3) RX_CSWAP(2, 4) RX_CSWAP(1, 2) RX_CSWAP(3, 4) RX_CSWAP(0, 3) RX_CSWAP(2, 5) RX_CSWAP(0, 1) RX_CSWAP(2, 3)
				RX_CSWAP(4, 5) RX_CSWAP(1, 2) RX_CSWAP(3, 4)
#undef RX_CSWAP
				// everything the walk will read is requested first: the position ranges of all the postings are in flight together
				float pr[kFtSparsePostings];
				uint8_t pf[kFtSparsePostings];
				uint32_t pq[kFtSparsePostings];
				const uint64_t* pp[kFtSparsePostings];
				uint32_t pn[kFtSparsePostings];
#pragma unroll
				for (uint32_t k = 0; k < kFtSparsePostings; ++k) {   // in sub-term order = the order mergeTerm met the postings
					pr[k] = 0.f;
					pf[k] = 0;
					pq[k] = 0;
					pp[k] = nullptr;
					pn[k] = 0;
					if (k >= cnt) continue;
					const uint4 o = s_rec[key[k] & 0xFFFFu];
					pr[k] = __uint_as_float(o.z);
					pf[k] = uint8_t((o.w >> 16) & 0xFFu);
					if (!p.simple || p.max_areas) ft_replay_locate(p, key[k] >> 16, o.y, s_fpos, s_pos_off, s_qp, pq[k], pp[k], pn[k]);
				}
				if (p.max_areas) {   // MergeDataAreas: the document's areas from its postings in merge order (before the ranks: independent of them)
					float max_term_rank = 0.f;
					const uint32_t sl_a = slot_of(dl);
#pragma unroll
					for (uint32_t k = 0; k < kFtSparsePostings; ++k) {
						if (k < cnt && __float_as_uint(pr[k]) != kFtSuppressedRank) ft_areas_of_posting(p, sl_a, pp[k], pn[k], pr[k], max_term_rank);
					}
				}
				uint32_t longest = 0;
#pragma unroll
				for (uint32_t k = 0; k < kFtSparsePostings; ++k) longest = pn[k] > longest ? pn[k] : longest;
				if (cnt > 1 && longest <= 4) {
					// ... and so are the positions (up to four per posting, the usual case): the walk, which compares the lists of two
					// consecutive terms at a time, then runs on registers instead of paying one memory round trip per posting
					FtPosRegs lists[kFtSparsePostings];
#pragma unroll
					for (uint32_t k = 0; k < kFtSparsePostings; ++k) {
						lists[k].n = pn[k];
#pragma unroll
						for (uint32_t i = 0; i < 4; ++i) lists[k].v[i] = i < pn[k] ? pp[k][i] : 0ull;
					}
					FtReplayStateT<FtPosRegs> st;
#pragma unroll
					for (uint32_t k = 0; k < kFtSparsePostings; ++k) {
						if (k < cnt) ft_replay_apply(p, st, pr[k], pf[k], pq[k], lists[k]);
					}
					ft_replay_finish(p, st, slot_of(dl), d_begin + dl, one_field, words0);
				} else {
					FtReplayState st;
#pragma unroll
					for (uint32_t k = 0; k < kFtSparsePostings; ++k) {
						FtPosList pos;
						pos.ptr = pp[k];
						pos.n = pn[k];
						if (k < cnt) ft_replay_apply(p, st, pr[k], pf[k], pq[k], pos);
					}
					ft_replay_finish(p, st, slot_of(dl), d_begin + dl, one_field, words0);
				}
			}
			sparse = !__syncthreads_or(over);   // a document with more postings than the network orders: the general path redoes the range
		}
		if (!sparse) {
		FT_STAMP(p, 38);
		for (uint32_t e = tid; e < nw; e += 256) {   // every posting of a merged document into the document's row, column = its sub-term
			const uint4 r = rec[e];
			const uint32_t sl = slot_of(r.x & (kFtRangeDocs - 1));
			if (sl >= p.max_merged) continue;   // met after the limit was hit: never added
			const uint64_t cell = uint64_t(r.w & 0xFFFFu) * p.max_merged + sl;
			p.e_rank[cell] = __uint_as_float(r.z);
			p.e_idx[cell] = r.y;
			p.e_field[cell] = uint8_t((r.w >> 16) & 0xFFu);
		}
		__syncthreads();   // the rows are read back by this workgroup only
		FT_STAMP(p, 39);
		for (uint32_t e = tid; e < nw; e += 256) {
			const uint4 r = rec[e];
			if (!(r.w >> 31)) continue;
			const uint32_t sl = slot_of(r.x & (kFtRangeDocs - 1));
			if (sl < p.max_merged) ft_replay_doc(p, sl, r.x, s_fpos, s_pos_off, s_qp);
		}
		__syncthreads();
		for (uint32_t e = tid; e < nw; e += 256) {   // the entry rows go back ZEROED: the occupancy test of the next merge relies on it
			const uint4 r = rec[e];
			const uint32_t sl = slot_of(r.x & (kFtRangeDocs - 1));
			if (sl < p.max_merged) p.e_rank[uint64_t(r.w & 0xFFFFu) * p.max_merged + sl] = 0.0f;
		}
		}
	}
	// ---- leave the shared tables clean for the next merge; the last workgroup writes the result header
	__syncthreads();
	FT_STAMP(p, 40);
	if (tid == 0) {
		if (n) p.bucket_cnt[range] = 0;
		// no release: the last workgroup reads nothing the others wrote in this kernel (header values come from the kernels before)
		s_last = atomicAdd(&p.sync[kFtSyncDoneFinish], 1u) == gridDim.x - 1 ? 1u : 0u;
	}
	__syncthreads();
	FT_STAMP(p, 41);
	if (!s_last) return;
	if (own_bases) {   // the number of merged documents = the whole table, cut at maxMergedDocs (ft_slot_bases did not run)
		const uint64_t total = uint64_t(p.n_rows) * p.n_ranges;
		uint32_t sum = 0;
		for (uint64_t j = tid; j < total; j += 256) sum += p.adders[j];
		sum = wave_sum(sum);
		if ((tid & 63) == 0) s_red[tid >> 6][0] = sum;
		__syncthreads();
		if (tid == 0) {
			const uint32_t all = s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0];
			p.sync[kFtSyncNumDocs] = all < p.max_merged ? all : p.max_merged;
		}
	}
	if (tid == 0) {
		p.out_header[0] = p.sync[kFtSyncNumDocs];
		p.out_header[1] = p.sync[kFtSyncError];
		p.out_header[2] = ft_preselect_on(p) ? 1u : 0u;
		p.out_header[3] = 0;
	}
	__syncthreads();
	if (tid < kFtSyncWords) p.sync[tid] = 0;
}

// The packed result (header + four arrays, ~11 B per merged document) leaves through a copy kernel writing 16-byte words straight into
// the caller's pinned host buffer: a hipMemcpyAsync of the same 220 KB took ~40 us from enqueue to completion (copy-engine start-up),
// a third of the merge.  Only the header and the first numDocs entries of every array are written.
__global__ __launch_bounds__(256) void ft_export(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	if (!p.host_out) return;
	const uint32_t n = p.out_header[0] < p.max_merged ? p.out_header[0] : p.max_merged;
	const uint4* src = reinterpret_cast<const uint4*>(p.out_header);
	uint4* dst = reinterpret_cast<uint4*>(p.host_out);
	// region r of the packed layout: [start, start + bytes actually used), in 16-byte words
	const size_t a16 = 16, m4 = (size_t(p.max_merged) * 4 + 255) & ~size_t(255), m2 = (size_t(p.max_merged) * 2 + 255) & ~size_t(255);
	const size_t starts[5] = {0, 256, 256 + m4, 256 + 2 * m4, 256 + 2 * m4 + m2};
	const size_t used[5] = {a16, size_t(n) * 4, size_t(n) * 4, size_t(n) * 2, size_t(n)};
	const size_t gtid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, gsize = size_t(gridDim.x) * blockDim.x;
#pragma unroll
	for (int r = 0; r < 5; ++r) {
		const size_t w0 = starts[r] / 16, w1 = (starts[r] + used[r] + 15) / 16;
		for (size_t w = w0 + gtid; w < w1; w += gsize) dst[w] = src[w];
	}
}

// The plan (sub-term and term descriptors, grid entries, per-field parameters: a few KB) comes in the same way: one small workgroup
// reads it from the pinned staging buffer and writes it where the kernels expect it — no copy-engine transfer in front of the train.
__global__ __launch_bounds__(256) void ft_import(const uint4* host_plan, uint4* dev_plan, uint32_t n16) {
	for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n16; w += gridDim.x * blockDim.x) dev_plan[w] = host_plan[w];
}
// the plans of a batch (and the FtPlan array itself) from their pinned staging buffers into HBM in ONE launch: blockIdx.y = piece
__global__ __launch_bounds__(256) void ft_import_pieces(FtImportBatch b) {
	const uint32_t j = blockIdx.y;
	const uint4* src = reinterpret_cast<const uint4*>(b.src[j]);
	uint4* dst = reinterpret_cast<uint4*>(b.dst[j]);
	for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < b.n16[j]; w += gridDim.x * blockDim.x) dst[w] = src[w];
}
hipError_t launch_ft_import_batch(const FtImportBatch& b, hipStream_t st) {
	if (b.n) hipLaunchKernelGGL(ft_import_pieces, dim3(4, b.n), dim3(256), 0, st, b);
	return hipGetLastError();
}
hipError_t launch_ft_import(const void* host_plan, void* dev_plan, size_t bytes, hipStream_t st) {
	const uint32_t n16 = uint32_t((bytes + 15) / 16);
	hipLaunchKernelGGL(ft_import, dim3((n16 + 255) / 256 < 16 ? (n16 + 255) / 256 : 16), dim3(256), 0, st, reinterpret_cast<const uint4*>(host_plan),
					   reinterpret_cast<uint4*>(dev_plan), n16);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- launch train
// buildRestrictingBitmask's synonym half (mergerimpl.h:347-361): for every AND part with multi-word synonyms, the documents that hold EVERY
// term of ONE of its synonyms (calcTermBitmask per term :252-274: any occurrence with a relevant field; AccumulateAnd over the synonym's
// terms; OR over the part's synonyms).  One workgroup per range of kFtRangeDocs documents, bitmaps in LDS; runs in front of ft_ranges only
// when the query has such parts.
__global__ __launch_bounds__(256) void ft_syn_masks(const FtPlan* plans) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	constexpr uint32_t kWords = kFtRangeDocs / 32;
	__shared__ uint32_t s_or[kWords], s_and[kWords], s_tmp[kWords];
	const uint32_t tid = threadIdx.x, range = blockIdx.x;
	const uint64_t d_begin = uint64_t(range) * kFtRangeDocs;
	for (uint32_t j = 0; j < p.n_syn_jobs; ++j) {
		const FtSynMaskJob job = p.syn_jobs[j];
		for (uint32_t w = tid; w < kWords; w += 256) s_or[w] = 0;
		for (uint32_t k = job.syn_begin; k < job.syn_end; ++k) {
			const FtSynonym syn = p.syns[p.job_syns[k]];
			for (uint32_t w = tid; w < kWords; w += 256) s_and[w] = 0xFFFFFFFFu;
			for (uint32_t t = syn.term_begin; t < syn.term_end; ++t) {
				const FtTermCfg& term = p.terms[t];
				for (uint32_t w = tid; w < kWords; w += 256) s_tmp[w] = 0;
				__syncthreads();
				for (uint32_t si = term.sub_begin; si < term.sub_end; ++si) {
					const FtPosSubterm& sub = p.subs[si];
					const uint32_t lo = range < sub.n_ranges ? sub.range_off[range] : uint32_t(sub.n);
					const uint32_t hi = range + 1 < sub.n_ranges ? sub.range_off[range + 1] : uint32_t(sub.n);
					for (uint32_t i = lo + tid; i < hi; i += 256) {
						bool rel = term.all_pos_boost != 0;
						if (!rel) {   // checkFieldsRelevance (phrasemergerimpl.h:93-125)
							for (uint32_t e = sub.ent_off[i], e1 = sub.ent_off[i + 1]; e < e1 && !rel; ++e) rel = term.field_boost[sub.ent_field[e]] != 0.0f;
						}
						if (rel) {
							const uint32_t local = uint32_t(sub.doc[i] - d_begin);
							atomicOr(&s_tmp[local >> 5], 1u << (local & 31));
						}
					}
				}
				__syncthreads();
				for (uint32_t w = tid; w < kWords; w += 256) s_and[w] &= s_tmp[w];
			}
			__syncthreads();
			if (syn.term_end > syn.term_begin) {   // (AccumulateAnd over no term leaves an empty mask: nothing to OR)
				for (uint32_t w = tid; w < kWords; w += 256) s_or[w] |= s_and[w];
			}
			__syncthreads();
		}
		for (uint32_t w = tid; w < kWords; w += 256) {
			const uint64_t gw = d_begin / 32 + w;
			if (gw < p.nwords) job.out[gw] = s_or[w];
		}
		__syncthreads();
	}
}

// plans: the Q plans in HBM; host_plans: the same on the host (grid sizes).  All Q merges run over ONE index (same documents, so the same
// document ranges and mask words).  Queries with multi-word synonyms run alone (ft_syn_masks in front); phrases ran before (ft_phrase.hip).
// phase < 0: the whole train; 0 / 1 / 2: the pieces a sharded merge exchanges between (rxgpu_internal.h)
hipError_t launch_ft_merge_phase(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, int phase, hipStream_t st) {
	constexpr size_t kFinishLds = (kFtRangeDocs / 2 + kFtRangeDocs) * sizeof(uint32_t), kFinishLdsFew = kFtRangeDocs * sizeof(uint32_t);
	static std::atomic<uint64_t> raised{0};
	if (hipError_t e = raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&ft_finish), kFinishLds); e != hipSuccess) return e;
	if (!nq) return hipSuccess;
	uint32_t rank_blocks = 0;
	bool any_pre = false, any_bases = false, any_many_rows = false;
	for (uint32_t q = 0; q < nq; ++q) {
		const FtPlan& p = host_plans[q];
		any_many_rows = any_many_rows || !(p.n_rows <= kFtBitmapRows && p.max_merged < 0xFFFFu);   // ft_finish's compact layout: 32 KB instead of 48
		rank_blocks = std::max(rank_blocks, (p.merge_blocks + kFtRankTiles - 1) / kFtRankTiles);
		any_pre = any_pre || (!p.simple && p.prescore);
		any_bases = any_bases || !ft_own_bases(p);
	}
	const FtPlan& p0 = host_plans[0];
	const uint32_t ranges = p0.range_count ? p0.range_count : p0.n_ranges;   // a document-range shard: its own ranges (nq == 1)
	const uint64_t words = p0.range_count ? uint64_t(p0.range_count) * (kFtRangeDocs / 32) : p0.nwords;
	if (phase < 0 || phase == 0) {
		if (p0.n_syn_jobs) hipLaunchKernelGGL(ft_syn_masks, dim3(p0.n_ranges, 1), dim3(256), 0, st, plans);   // (nq == 1: the caller's rule)
		hipLaunchKernelGGL(ft_ranges, dim3(ranges, nq), dim3(256), 0, st, plans);
	}
	if (phase < 0 || phase == 1) {
		if (any_pre) {
			hipLaunchKernelGGL(ft_preselect_apply, dim3(uint32_t((words + 256 * kFtApplyWords - 1) / (256 * kFtApplyWords)), nq), dim3(256), 0, st, plans);
		}
		if (rank_blocks) hipLaunchKernelGGL(ft_rank_all, dim3(rank_blocks, nq), dim3(256), 0, st, plans);
		hipLaunchKernelGGL(ft_adders, dim3(ranges, nq), dim3(256), 0, st, plans);
	}
	if (phase < 0 || phase == 2) {
		if (any_bases) hipLaunchKernelGGL(ft_slot_bases, dim3(1, nq), dim3(256), 0, st, plans);
		hipLaunchKernelGGL(ft_finish, dim3(ranges, nq), dim3(256), any_many_rows ? kFinishLds : kFinishLdsFew, st, plans);
	}
	return hipGetLastError();
}

hipError_t launch_ft_merge(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, hipStream_t st) { return launch_ft_merge_phase(plans, host_plans, nq, -1, st); }
void launch_ft_slot_bases(const FtPlan* plans, uint32_t nq, hipStream_t st) { hipLaunchKernelGGL(ft_slot_bases, dim3(1, nq), dim3(256), 0, st, plans); }

// ---------------------------------------------------------------------------------------------- document-range shards: what travels between the kernels
// One shard's pre-score histogram, its kFtHistCopies copies added up (fine counters, then the chunk counters), and the popcount of its
// mask words behind them: what the all-gather carries.
__global__ __launch_bounds__(256) void ft_shard_fold(const FtPlan* plans, uint32_t* dst) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < kFtFoldWords; i += gridDim.x * 256) {
		uint32_t v = 0;
		if (i < kFtHistStride) {
			if (p.hist) {
#pragma unroll
				for (uint32_t k = 0; k < kFtHistCopies; ++k) v += p.hist[size_t(k) * kFtHistStride + i];
			}
		} else if (i == kFtHistStride) {
			v = p.sync[kFtSyncPop];
		}
		dst[i] = v;
	}
}
// ... and back: copy 0 of the histogram = the sum over the shards (the other copies zero), the popcount = the sum: ft_preselect_on and
// ft_pick_threshold then decide on the WHOLE index, every shard alike.  pos[s] = where shard s lies in the gathered buffer.
__global__ __launch_bounds__(256) void ft_shard_hist_combine(const FtPlan* plans, const uint32_t* gathered, const uint32_t* pos, uint32_t n_shards) {
	FtPlanK& p = FT_PLAN_OF_QUERY(plans);
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i <= kFtHistStride; i += gridDim.x * 256) {
		uint32_t v = 0;
		for (uint32_t sh = 0; sh < n_shards; ++sh) v += gathered[size_t(pos[sh]) * kFtFoldWords + i];
		if (i < kFtHistStride) {
			if (p.hist) {
				p.hist[i] = v;
#pragma unroll
				for (uint32_t k = 1; k < kFtHistCopies; ++k) p.hist[size_t(k) * kFtHistStride + i] = 0u;
			}
		} else {
			p.sync[kFtSyncPop] = v;
		}
	}
}
// the table of ft_adders: every shard filled its own columns (the rest zero) — the sum is the table of the whole index
__global__ __launch_bounds__(256) void ft_shard_table_sum(uint32_t* table, const uint32_t* gathered, const uint32_t* pos, uint32_t n_shards, uint64_t n, uint64_t stride) {
	for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
		uint32_t v = 0;
		for (uint32_t sh = 0; sh < n_shards; ++sh) v += gathered[size_t(pos[sh]) * stride + i];
		table[i] = v;
	}
}
void launch_ft_shard_fold(const FtPlan* plan, uint32_t* dst, hipStream_t st) { hipLaunchKernelGGL(ft_shard_fold, dim3(64), dim3(256), 0, st, plan, dst); }
void launch_ft_shard_hist_combine(const FtPlan* plan, const uint32_t* gathered, const uint32_t* pos, uint32_t n_shards, hipStream_t st) {
	hipLaunchKernelGGL(ft_shard_hist_combine, dim3(64), dim3(256), 0, st, plan, gathered, pos, n_shards);
}
void launch_ft_shard_table_sum(uint32_t* table, const uint32_t* gathered, const uint32_t* pos, uint32_t n_shards, uint64_t n, uint64_t stride, hipStream_t st) {
	if (!n) return;
	hipLaunchKernelGGL(ft_shard_table_sum, dim3(uint32_t(std::min<uint64_t>(256, (n + 255) / 256))), dim3(256), 0, st, table, gathered, pos, n_shards, n, stride);
}

// the results' way out (after the merge's timing bracket: the roofline of the merge kernels does not include the transfer)
hipError_t launch_ft_export(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, hipStream_t st) {
	bool any = false;
	for (uint32_t q = 0; q < nq; ++q) any = any || host_plans[q].host_out;
	if (any) hipLaunchKernelGGL(ft_export, dim3(nq > 4 ? 16 : 64, nq), dim3(256), 0, st, plans);
	return hipGetLastError();
}

}  // namespace rxgpu
