// Device helpers shared by the two ft_fast merge trains (ft_merge.hip: the dense train; ft_sparse.hip: the train for sparsely hit document
// ranges): the plan in the constant address space, the pre-score threshold (mergerimpl.h:433-446), the positions distance
// (mergerimpl.h:20-37) and the per-document replay of mergeTerm / mergeSimple / mergePhrase (mergerimpl.h:39-250).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rxgpu_internal.h"
#include "ft_rank.hip.h"

namespace rxgpu {

// Every kernel of the train takes the plans of a BATCH of Q merges over one index (grid.y = query; a single merge is a batch of one).  The
// plans lie in HBM — uploaded with the rest of the plan by the import kernel — and are read through the constant address space: uniform
// scalar loads on demand, exactly what a by-value kernel argument compiles to, so one merge costs what it cost when the plan travelled as
// the argument, and Q merges share the launch floors (~1.5-2 us of boundary plus the ramp of a 600-workgroup grid per kernel, five
// kernels: a 400 k-posting merge is latency from end to end) and fill the device together.
typedef const FtPlan __attribute__((address_space(4))) FtPlanK;
#define FT_PLAN_OF_QUERY(plans) (*(FtPlanK*)((plans) + blockIdx.y))

// phase stamps of one workgroup (100 MHz wall clock), see rxgpu_ft_read_stats
#define FT_STAMP(p, k)                                                                                    \
	do {                                                                                                  \
		if ((p).dbg && blockIdx.x == (p).dbg_block && threadIdx.x == 0) (p).dbg[k] = wall_clock64();      \
	} while (0)

__device__ __forceinline__ bool ft_preselect_on(FtPlanK& p) {   // mergerimpl.h:486-490, the half only the device knows
	return p.prescore && __hip_atomic_load(&p.sync[kFtSyncPop], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > p.merge_limit;
}

// mergerimpl.h:433-446: walk the scores downwards until maxMergedDocs documents are covered.  A score sc is visited iff the documents
// strictly above it are fewer than maxMergedDocs; minScore = the lowest visited score >= 1, minScoreDocs = maxMergedDocs - (documents above
// it).  `above` is monotone, so the boundary falls inside ONE chunk of 64 scores: ft_ranges also keeps the 1024 chunk totals
// (behind the fine counters of every histogram copy), every workgroup of ft_preselect_apply finds the boundary chunk from those (one 16-byte load per thread, a suffix scan)
// and one wavefront resolves it on the chunk's 64 fine counters — 4 KB + 256 B read per workgroup instead of a kernel of its own
// (a single workgroup summing the 256 KB histogram: 11 us).
__device__ __forceinline__ void ft_pick_threshold(FtPlanK& p, uint32_t* out_score, uint32_t* out_docs) {
	__shared__ uint32_t s_wave_tot[4], s_found[2], s_res[2];
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	uint32_t c[4] = {0u, 0u, 0u, 0u};   // chunks 4 t .. 4 t + 3, summed over the copies
#pragma unroll
	for (uint32_t k = 0; k < kFtHistCopies; ++k) {
		const uint4 c4 = reinterpret_cast<const uint4*>(p.hist + size_t(k) * kFtHistStride + 65536)[t];
		c[0] += c4.x;
		c[1] += c4.y;
		c[2] += c4.z;
		c[3] += c4.w;
	}
	const uint32_t mine = c[0] + c[1] + c[2] + c[3];   // every document is counted once: the sums stay below 2^32
	uint32_t incl = mine;   // inclusive suffix over the lanes
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t o = __shfl_down(incl, off, 64);
		if (lane + off < 64) incl += o;
	}
	if (lane == 0) s_wave_tot[wave] = incl;
	if (t == 0) s_found[0] = 0xFFFFFFFFu;
	__syncthreads();
	uint32_t above = incl - mine;   // documents in the chunks above this thread's four
	for (int w = wave + 1; w < 4; ++w) above += s_wave_tot[w];
#pragma unroll
	for (int k = 3; k >= 0; --k) {
		const uint32_t g = uint32_t(4 * t + k);
		// the lowest chunk whose TOP score is visited (documents in higher chunks < maxMergedDocs); chunk 0 only counts through scores >= 1
		const bool top_visited = above < p.max_merged;
		const bool below_visited = g > 0 && above + c[k] < p.max_merged;   // would the top of chunk g - 1 be visited too?
		if (top_visited && !below_visited) {   // exactly one (thread, k): `above` is monotone
			s_found[0] = g;
			s_found[1] = above;
		}
		above += c[k];
	}
	__syncthreads();
	const uint32_t g = s_found[0];
	if (wave == 0) {
		uint32_t res_score = 65535u, res_docs = 0;   // the reference's initial values (unreachable for g: the top chunk has nothing above it)
		if (g != 0xFFFFFFFFu) {
			const uint32_t sc = g * 64 + uint32_t(lane);
			uint32_t h = 0;
#pragma unroll
			for (uint32_t k = 0; k < kFtHistCopies; ++k) h += p.hist[size_t(k) * kFtHistStride + sc];
			uint32_t fine = h;   // inclusive suffix over the lanes: documents with a score in [sc, top of the chunk]
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const uint32_t o = __shfl_down(fine, off, 64);
				if (lane + off < 64) fine += o;
			}
			const uint32_t above_sc = s_found[1] + (fine - h);
			const bool visited = sc >= 1 && above_sc < p.max_merged;
			const unsigned long long vis = __ballot(visited);
			if (vis) {   // visited lanes form a suffix of the wave: the lowest one is minScore; none: only score 0 of chunk 0 is left
				const int low = __ffsll((long long)vis) - 1;
				res_score = uint32_t(__shfl(int(sc), low, 64));
				res_docs = p.max_merged - uint32_t(__shfl(int(above_sc), low, 64));
			}
		}
		if (lane == 0) {
			s_res[0] = res_score;
			s_res[1] = res_docs;
		}
	}
	__syncthreads();
	*out_score = s_res[0];
	*out_docs = s_res[1];
	__syncthreads();   // the shared words may be reused by the caller's next shared-memory helper
}

// mergerimpl.h:20-37; fullPos()/fullField() truncate the 64-bit PosType to uint32_t exactly like the reference's accessors
__device__ __forceinline__ unsigned ft_positions_distance_regs(const uint64_t (&ra)[4], uint32_t na, const uint64_t (&rb)[4], uint32_t nb) {
	unsigned res = 0xFFFFFFFFu;
	uint32_t i = 0, j = 0;
	while (i < na && j < nb) {
		const uint64_t pa = i == 0 ? ra[0] : i == 1 ? ra[1] : i == 2 ? ra[2] : ra[3];
		const uint64_t pb = j == 0 ? rb[0] : j == 1 ? rb[1] : j == 2 ? rb[2] : rb[3];
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
__device__ __forceinline__ unsigned ft_positions_distance(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb) {
	if (na <= 4 && nb <= 4) {   // the usual case: both lists fetched at once (eight independent loads), the walk runs on registers
		uint64_t ra[4], rb[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			ra[k] = k < na ? a[k] : 0ull;
			rb[k] = k < nb ? b[k] : 0ull;
		}
		return ft_positions_distance_regs(ra, na, rb, nb);
	}
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
struct FtPosList {   // a posting's positions where the index keeps them
	const uint64_t* ptr = nullptr;
	uint32_t n = 0;
};
struct FtPosRegs {   // ... fetched ahead (n <= 4)
	uint64_t v[4] = {0, 0, 0, 0};
	uint32_t n = 0;
};
__device__ __forceinline__ unsigned ft_positions_distance(const FtPosList& a, const FtPosList& b) { return ft_positions_distance(a.ptr, a.n, b.ptr, b.n); }
__device__ __forceinline__ unsigned ft_positions_distance(const FtPosRegs& a, const FtPosRegs& b) { return ft_positions_distance_regs(a.v, a.n, b.v, b.n); }
template <typename Pos>
struct FtReplayStateT {
	bool created = false;
	float proc = 0.f, rank = 0.f;
	uint8_t field = 0;
	Pos last, next;
	uint16_t switched_term = 0, last_counted = 0, terms_counter = 0;
	// multi-word synonyms (mergerimpl.h:509-555): the qp the document was created at, the synonyms whose end has been applied to it,
	// MergerDocumentData::containsFullMultiWordSynonym
	uint16_t created_qp = 0, syn_done = 0;
	bool contains_full = false;
};
// The loop behind every synonym's terms (mergerimpl.h:516-523), applied lazily: a document created by a synonym's term keeps its term count
// only if it met every term of the synonym that just ended.  Called with the qp of the posting about to be applied (or past the last).
template <typename Pos>
__device__ __forceinline__ void ft_replay_synonym_ends(FtPlanK& p, FtReplayStateT<Pos>& st, uint32_t qp) {
	while (st.syn_done < p.n_syn && p.syns[st.syn_done].end_qp < qp) {
		if (st.created && st.created_qp > p.n_part_qp) {
			if (st.terms_counter < p.syns[st.syn_done].nterms) {
				st.terms_counter = 0;
			} else {
				st.contains_full = true;
			}
		}
		st.syn_done = uint16_t(st.syn_done + 1);
	}
}
using FtReplayState = FtReplayStateT<FtPosList>;
// one posting of the document, met in sub-term order: (rank r, field fld, posting index i) of sub-term row `row`
// the positions of posting i of sub-term row `row`, and the query position of its term
__device__ __forceinline__ void ft_replay_locate(FtPlanK& p, uint32_t row, uint32_t i, const uint64_t* const* s_fpos, const uint32_t* const* s_pos_off,
												 const uint32_t* s_qp, uint32_t& qp, const uint64_t*& pos, uint32_t& npos) {
	const uint64_t* fpos;
	const uint32_t* pos_off;
	if (row < kFtReplayRows) {
		fpos = s_fpos[row];
		pos_off = s_pos_off[row];
		qp = s_qp[row];
	} else {
		const FtPosSubterm& s = p.subs[p.merge_grid[row].sub];
		fpos = s.fpos;
		pos_off = s.pos_off;
		qp = ft_row_qpw(s);
	}
	const uint32_t po0 = pos_off[i], po1 = pos_off[i + 1];
	pos = fpos + po0;
	npos = po1 - po0;
}
// addAreas (merger.h:196-204) for one posting of merged document `sl`: every position -> AreasInDocument::AddWord(Area(pos, pos + 1, arrayIdx),
// field, rank, maxAreasInDoc) until one is refused, then UpdateRank(rank) (areaholder.h:125-132, 76-95).  AreasInField::Insert: the new word
// joins the area inserted last when Area::Concat says so (same array index, touching or overlapping, :14-29); otherwise it is appended while
// fewer than maxAreasInDoc areas are held, and once they are it overwrites the oldest in turn — only for a term rank above the best the
// document has seen (maxTermRank_), else the word is refused and the rest of the posting skipped.  The areas of a (document, field) are
// the thread's own words in HBM: one thread replays one document.
__device__ __forceinline__ void ft_areas_of_posting(FtPlanK& p, uint32_t sl, const uint64_t* pos, uint32_t npos, float rank, float& max_term_rank) {
	const uint32_t nf = p.area_fields, cap = p.max_areas;
	for (uint32_t i = 0; i < npos; ++i) {
		const uint64_t w = pos[i];
		const uint32_t wpos = uint32_t(w) & 0x0FFFFFFFu, arr = uint32_t(w >> 28) & 0x0FFFFFFFu, field = uint32_t(w >> 56);
		if (field >= nf) break;   // (cannot happen: the upload checks fields)
		uint32_t* hdr = p.area_hdr + (size_t(sl) * nf + field) * 2;
		uint32_t* areas = p.out_areas + (size_t(sl) * nf + field) * cap * 3;
		const uint32_t held = hdr[0], index = hdr[1];
		const uint32_t a_start = wpos, a_end = wpos + 1;
		bool ok = false;
		if (index > 0) {   // Concat with the area inserted last
			uint32_t* prev = areas + size_t((index - 1) % cap) * 3;
			const uint32_t ps = prev[0], pe = prev[1];
			if (prev[2] == arr && ((a_start <= pe && a_start >= ps) || (a_end <= pe && a_end >= ps) || (ps > a_start && pe < a_end))) {
				if (ps > a_start) prev[0] = a_start;
				if (pe < a_end) prev[1] = a_end;
				ok = true;
			}
		}
		if (!ok) {
			if (held == cap) {
				if (rank > max_term_rank) {
					uint32_t* slot = areas + size_t(index % cap) * 3;
					slot[0] = a_start;
					slot[1] = a_end;
					slot[2] = arr;
					hdr[1] = index + 1;
					ok = true;
				}
			} else {
				uint32_t* slot = areas + size_t(held) * 3;
				slot[0] = a_start;
				slot[1] = a_end;
				slot[2] = arr;
				hdr[0] = held + 1;
				hdr[1] = index + 1;
				ok = true;
			}
		}
		if (!ok) break;
	}
	if (rank > max_term_rank) max_term_rank = rank;
}

// one posting of the document, met in sub-term order: rank r in field fld, positions `pos` (not read for a simple merge)
// (qpw = ft_row_qpw of the posting's row: the query position, whether the row is a phrase's, the last plain term in front of that phrase)
template <typename Pos>
__device__ __forceinline__ void ft_replay_apply(FtPlanK& p, FtReplayStateT<Pos>& st, float r, uint8_t fld, uint32_t qpw, const Pos& pos) {
	const uint16_t qp = uint16_t(qpw & 0x7FFFu);
	if (p.n_syn) ft_replay_synonym_ends(p, st, qp);
	if (__float_as_uint(r) == kFtSuppressedRank) {   // mergerimpl.h:144-151: a merged document counts the term, nothing else
		if (st.created && st.last_counted < qp) {
			st.terms_counter = uint16_t(st.terms_counter + 1);
			st.last_counted = qp;
		}
		return;
	}
	if (p.simple) {   // mergeSimple, mergerimpl.h:232-240: strict <, so the first maximum (and its field) wins
		if (!st.created) {
			st.created = true;
			st.proc = r;
			st.field = fld;
		} else if (st.proc < r) {
			st.proc = r;
			st.field = fld;
		}
		return;
	}
	if (qpw & 0x8000u) {   // ---- mergePhrase (mergerimpl.h:39-90): the document's phrase rank, no distance, no switchToNextWord of its own
		if (!st.created) {   // :60-73: MergerDocumentData(rank = the PhraseMerger's rank, 0 after its last term), lastTermPositions = the phrase's
			st.created = true;
			st.proc = r;
			st.field = fld;
			st.rank = 0.f;
			st.last = pos;
			st.next.n = 0;
			st.switched_term = qp;
			st.last_counted = qp;
			st.terms_counter = 1;
			st.created_qp = qp;
			return;
		}
		// the switchToNextWord calls of the plain terms between the document's last posting and this phrase (merger.h:218-226) come first
		const uint16_t prev_term = uint16_t(qpw >> 16);
		if (st.switched_term < prev_term) {
			if (st.next.n) {
				st.last = st.next;
				st.next.n = 0;
				st.rank = 0.f;
			}
			st.switched_term = prev_term;
		}
		if (st.last_counted < qp) {
			st.terms_counter = uint16_t(st.terms_counter + 1);
			st.last_counted = qp;
		}
		st.proc += r;          // :77-80 (nextTermPositions stays: the next plain term's switchToNextWord swaps it in over the phrase's)
		st.last = pos;
		st.rank = 0.f;
		return;
	}
	if (!st.created) {   // addDoc (mergerimpl.h:160-164)
		st.created = true;
		st.proc = r;
		st.field = fld;
		st.rank = r;
		st.next = pos;
		st.switched_term = qp;
		st.last_counted = qp;
		st.terms_counter = 1;
		st.created_qp = qp;
		return;
	}
	// ---- document already merged: mergerimpl.h:165-189
	if (st.switched_term < qp) {   // switchToNextWord (merger.h:218-226) ran before every term since: idempotent after the first time
		if (st.next.n) {
			st.last = st.next;
			st.next.n = 0;
			st.rank = 0.f;
		}
		st.switched_term = qp;
	}
	if (st.last_counted < qp) {   // InreaseTermsCounter
		st.terms_counter = uint16_t(st.terms_counter + 1);
		st.last_counted = qp;
	}
	unsigned dist = ft_positions_distance(st.last, pos);
	dist = dist > 1u ? dist : 1u;
	const float norm_dist = ft_bound(float(1.0 / double(float(dist))), p.distance_weight, p.distance_boost);
	const float final_rank = norm_dist * r;
	if (final_rank > st.rank) {
		st.proc -= st.rank;
		st.proc += final_rank;
		st.next = pos;
		st.rank = final_rank;
	}
}
// addFullMatchBoost (merger.h:100-109): a document whose best field holds exactly as many words as the query has parts — and, for a
// multi-term query, that met every part (canBeBoostedByFullMatch, mergerimpl.h:527-531) — is boosted.  Done here because the word counts
// are resident: on the host it was one cache miss per merged document.
template <typename Pos>
__device__ __forceinline__ void ft_replay_finish(FtPlanK& p, FtReplayStateT<Pos>& st, uint32_t sl, uint32_t doc, bool have_words = false,
												 float words0 = 0.f) {
	if (p.n_syn) {
		ft_replay_synonym_ends(p, st, 0xFFFFFFFFu);
		if (st.created_qp > p.n_part_qp && !st.contains_full) {   // only parts of a multi-word synonym: removed (mergerimpl.h:533-555; the host compacts)
			p.out_proc[sl] = 0.f;
			p.out_field[sl] = st.field;
			p.out_terms_counter[sl] = 0xFFFFu;
			return;
		}
	}
	float proc = st.proc;
	const FtTermCfg& t0 = p.terms[0];
	const float words = have_words ? words0 : t0.words[size_t(doc) * t0.num_fields + st.field];
	const bool full = p.simple ? words == 1.0f : (st.terms_counter == p.n_parts && words == float(p.query_len));
	if (full) proc = float(double(proc) * p.full_match_boost);
	p.out_proc[sl] = proc;
	p.out_field[sl] = st.field;
	p.out_terms_counter[sl] = st.terms_counter;
}

}  // namespace rxgpu
