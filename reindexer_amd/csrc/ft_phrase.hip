// PhraseMerger<IdCont, MergeData, uint32_t>::Merge (cpp_src/core/ft/ft_fast/phrasemergerimpl.h:161-329, phrasemerger.h:11-55, 107-140) on
// gfx950: one phrase ("w1 w2 w3"~d) of a ft_fast query, merged into the posting-list ROWS the main merge (ft_merge.hip) then treats like
// dictionary words (mergePhrase, mergerimpl.h:39-90, reads nothing but each document's rank, field and lastPhrasePositions).
//
// The reference walks (term, sub-term, posting) in order; what that order decides is
//   * which documents are merged at all: preselectDocsContainingAllTerms (:259-303) = the documents every term of the phrase holds, minus
//     the removed and the excluded ones — a per-document fact (membership by binary search inside the word's range index);
//   * mergeData_ order and the maxMergedDocs_ cut (:181-183, 209-215): a document is added by its first posting of the FIRST term, in
//     (sub-term, document) order, that is preselected and has a non-zero rank, while fewer than maxMergedDocs_ were added
//     ->  ft_phrase_admit: one flag per posting of the first term, ordered prefix over the launch (ft_scan.hip.h) = the slot;
//   * the per-document state (rank / proc updates, AddPositions, MergeWithDist, SwitchPositions): every document is on its own
//     ->  ft_phrase_docs: one thread per admitted document walks the terms and their sub-terms in order with the reference's float
//     operations; its two position lists live in a workspace sized by the admission pass;
//   * mergePhrase walks mergeData_ in order: ft_phrase_pack (one workgroup) keeps the documents with a non-zero rank, in slot order,
//     as one list per sub-term of the first term (ascending documents inside: the slot order IS (sub-term, document)), plus the range
//     index the merge kernels expect.
// Not the hot path of the merger (phrase documents are an intersection: few): written for exactness, not for the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rxgpu_internal.h"
#include "ft_rank.hip.h"
#include "ft_scan.hip.h"

namespace rxgpu {

namespace {

// posting of document d in a list, or 0xFFFFFFFF (the range index narrows the search to one range of kFtRangeDocs documents)
__device__ __forceinline__ uint32_t phrase_find(const FtPosSubterm& s, uint32_t d) {
	const uint32_t rg = d >> kFtRangeShift;
	if (rg >= s.n_ranges) return 0xFFFFFFFFu;
	uint32_t lo = s.range_off[rg], hi = rg + 1 < s.n_ranges ? s.range_off[rg + 1] : uint32_t(s.n);
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (s.doc[mid] < d) {
			lo = mid + 1;
		} else {
			hi = mid;
		}
	}
	return lo < uint32_t(s.n) && s.doc[lo] == d ? lo : 0xFFFFFFFFu;
}

__device__ __forceinline__ float phrase_rank(const FtPhrasePlan& p, uint32_t t, const FtPosSubterm& s, uint32_t i, uint32_t d, uint8_t* field) {
	return ft_term_rank(p.terms[t], s, s.ent_off[i], s.ent_off[i + 1], d, field);
}

// preselectDocsContainingAllTerms for one document that the first term holds
__device__ bool phrase_preselected(const FtPhrasePlan& p, uint32_t d) {
	if (p.removed && p.removed[d]) return false;
	if (p.excluded && p.excluded[d]) return false;
	for (uint32_t t = 1; t < p.nterms; ++t) {
		bool found = false;
		for (uint32_t si = p.terms[t].sub_begin; si < p.terms[t].sub_end && !found; ++si) found = phrase_find(p.subs[si], d) != 0xFFFFFFFFu;
		if (!found) return false;
	}
	return true;
}

// the positions a document can carry out of one term: the sum of its occurrences over the term's sub-terms (AddPositions appends them
// all; MergePositionsWithDist emits every position of the right list at most once per call)
__device__ uint32_t phrase_cap(const FtPhrasePlan& p, uint32_t d) {
	uint32_t cap = 0;
	for (uint32_t t = 0; t < p.nterms; ++t) {
		uint32_t sum = 0;
		for (uint32_t si = p.terms[t].sub_begin; si < p.terms[t].sub_end; ++si) {
			const FtPosSubterm& s = p.subs[si];
			const uint32_t i = phrase_find(s, d);
			if (i != 0xFFFFFFFFu) sum += s.pos_off[i + 1] - s.pos_off[i];
		}
		cap = sum > cap ? sum : cap;
	}
	return cap;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- admission (first term)
__global__ __launch_bounds__(256) void ft_phrase_admit(FtPhrasePlan p) {
	const uint32_t ticket = grab_ticket(p.sync + 0);
	const FtGridEntry ge = grid_entry(p.grid, p.n_grid, ticket);
	const FtPosSubterm& s = p.subs[ge.sub];
	const uint32_t row = ge.sub - p.terms[0].sub_begin;
	const uint64_t i0 = uint64_t(ticket - ge.block_base) * kFtBlockPostings + uint64_t(threadIdx.x) * kFtPassItems;
	uint32_t doc[kFtPassItems];
	bool flag[kFtPassItems];
	uint32_t count = 0;
#pragma unroll
	for (int k = 0; k < kFtPassItems; ++k) {
		flag[k] = false;
		doc[k] = 0;
		const uint64_t i = i0 + uint64_t(k);
		if (i >= s.n) continue;
		const uint32_t d = s.doc[i];
		doc[k] = d;
		if (!phrase_preselected(p, d)) continue;
		uint8_t f;
		if (phrase_rank(p, 0, s, uint32_t(i), d, &f) == 0.0f) continue;   // fp::IsZero(termRank): the posting adds nothing
		bool first = true;   // an earlier sub-term of the first term added the document already
		for (uint32_t r = 0; r < row && first; ++r) {
			const FtPosSubterm& e = p.subs[p.terms[0].sub_begin + r];
			const uint32_t j = phrase_find(e, d);
			if (j != 0xFFFFFFFFu && phrase_rank(p, 0, e, j, d, &f) != 0.0f) first = false;
		}
		flag[k] = first;
		count += first ? 1u : 0u;
	}
	uint32_t grand = 0;
	uint32_t slot = ordered_prefix(count, ticket, p.lookback, p.sync + 1, &grand);
#pragma unroll
	for (int k = 0; k < kFtPassItems; ++k) {
		if (!flag[k]) continue;
		if (slot < p.max_merged) {   // NumDocsMerged() < maxMergedDocs_
			const uint32_t cap = phrase_cap(p, doc[k]);
			p.slot_doc[slot] = doc[k];
			p.slot_row[slot] = row;
			p.slot_cap[slot] = cap;
			atomicAdd(reinterpret_cast<unsigned long long*>(p.sync + 4), (unsigned long long)cap);
		}
		++slot;
	}
	if (ticket + 1 == p.grid_blocks && threadIdx.x == 0) p.sync[2] = grand < p.max_merged ? grand : p.max_merged;
}

// ---------------------------------------------------------------------------------------------- the admitted documents, term by term
namespace {

// MergePositionsWithDist (phrasemerger.h:24-55) for PositionsVector results: the positions of the new word that follow a position of the
// phrase so far within `dist` in the same field; returns the smallest such distance (INT_MAX if none)
__device__ int phrase_merge_with_dist(const uint64_t* left, uint32_t nl, const uint64_t* right, uint32_t nr, unsigned dist, uint64_t* out, uint32_t* nout) {
	unsigned min_dist = 0x7FFFFFFFu;
	uint32_t j = 0, n = *nout;
	for (uint32_t i = 0; i < nl; ++i) {
		const uint64_t l = left[i];
		const uint32_t lpos = uint32_t(l), lfield = uint32_t(l >> 28);   // PosType::fullPos / fullField (idrelset.h:20-23)
		while (j < nr && uint32_t(right[j]) < lpos) ++j;
		if (j == nr) break;
		while (j < nr) {
			const uint64_t r = right[j];
			if (uint32_t(r >> 28) != lfield || uint32_t(r) - lpos > dist) break;
			const unsigned dd = uint32_t(r) - lpos;
			min_dist = dd < min_dist ? dd : min_dist;
			out[n++] = r;
			++j;
		}
	}
	*nout = n;
	return int(min_dist);
}

// SwitchPositions (phrasemerger.h:129-136): sort, unique.  The list is a handful of ascending runs (one per sub-term): insertion sort.
__device__ uint32_t phrase_sort_unique(uint64_t* v, uint32_t n) {
	for (uint32_t i = 1; i < n; ++i) {
		const uint64_t x = v[i];
		uint32_t j = i;
		while (j > 0 && v[j - 1] > x) {
			v[j] = v[j - 1];
			--j;
		}
		v[j] = x;
	}
	uint32_t m = 0;
	for (uint32_t i = 0; i < n; ++i) {
		if (m == 0 || v[m - 1] != v[i]) v[m++] = v[i];
	}
	return m;
}

}  // namespace

__global__ __launch_bounds__(64) void ft_phrase_docs(FtPhrasePlan p, uint32_t admitted) {
	const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
	if (slot >= admitted) return;
	const uint32_t d = p.slot_doc[slot], cap = p.slot_cap[slot];
	const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long*>(p.sync + 6), 2ull * cap);
	uint64_t* last = p.ws + at;
	uint64_t* next = last + cap;
	uint32_t nl = 0, nn = 0;
	float proc = 0.f, rk = 0.f;
	uint8_t field = 0;
	bool created = false, alive = true;
	for (uint32_t t = 0; t < p.nterms && alive; ++t) {
		const unsigned dist = unsigned(p.distance[t]);
		for (uint32_t si = p.terms[t].sub_begin; si < p.terms[t].sub_end; ++si) {
			const FtPosSubterm& s = p.subs[si];
			const uint32_t i = phrase_find(s, d);
			if (i == 0xFFFFFFFFu) continue;
			uint8_t f = 0;
			const float rank = phrase_rank(p, t, s, i, d, &f);
			if (rank == 0.0f) continue;
			const uint64_t* pos = s.fpos + s.pos_off[i];
			const uint32_t np = s.pos_off[i + 1] - s.pos_off[i];
			if (t == 0) {   // mergePhraseTerm, isFirstTerm (:204-223)
				if (!created) {
					created = true;
					proc = rank;
					field = f;
					rk = rank;
				} else if (rank > rk) {
					rk = rank;
					proc = rank;
				}
				for (uint32_t k = 0; k < np && nn < cap; ++k) next[nn++] = pos[k];   // InitFrom / AddPositions
			} else {        // :224-241
				const int min_dist = phrase_merge_with_dist(last, nl, pos, np, dist, next, &nn);
				if (nn == 0) continue;
				const float norm_dist = ft_bound(float(1.0 / double(min_dist < 1 ? 1 : min_dist)), p.distance_weight, p.distance_boost);
				const float final_rank = norm_dist * rank;
				if (final_rank > rk) {
					proc -= rk;
					rk = final_rank;
					proc += final_rank;
				}
			}
		}
		if (nn == 0) {   // :246-253: the phrase breaks off in this document
			alive = false;
			proc = 0.f;
			nl = 0;
		} else {
			nl = phrase_sort_unique(next, nn);
			uint64_t* tmp = last;
			last = next;
			next = tmp;
			nn = 0;
		}
		rk = 0.f;
	}
	p.slot_proc[slot] = proc;
	p.slot_field[slot] = field;
	p.slot_pos[slot] = uint64_t(last - p.ws);
	p.slot_npos[slot] = nl;
}

// ---------------------------------------------------------------------------------------------- rows for the main merge
constexpr uint32_t kPackThreads = 1024, kPackRowsLds = 4096;
__global__ __launch_bounds__(kPackThreads) void ft_phrase_pack(FtPhrasePlan p) {
	__shared__ uint32_t s_cnt[kPackRowsLds], s_first[kPackRowsLds];
	__shared__ uint32_t s_wave[2][kPackThreads / 64];
	__shared__ uint32_t s_tot[2];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t S = p.sync[2];
	const uint32_t per = (S + kPackThreads - 1) / kPackThreads;
	const uint32_t lo = tid * per < S ? tid * per : S, hi = lo + per < S ? lo + per : S;
	for (uint32_t r = tid; r < p.n_rows0; r += kPackThreads) s_cnt[r] = 0;
	__syncthreads();
	uint32_t docs = 0, poss = 0;
	for (uint32_t sl = lo; sl < hi; ++sl) {
		if (p.slot_proc[sl] == 0.0f) continue;   // mergePhrase skips fp::IsZero(proc) (mergerimpl.h:47-50)
		++docs;
		poss += p.slot_npos[sl];
		atomicAdd(&s_cnt[p.slot_row[sl]], 1u);
	}
	const uint32_t di = wave_inclusive_scan(docs, int(lane)), pi = wave_inclusive_scan(poss, int(lane));
	if (lane == 63) {
		s_wave[0][wave] = di;
		s_wave[1][wave] = pi;
	}
	__syncthreads();
	uint32_t dex = di - docs, pex = pi - poss;
	for (uint32_t w = 0; w < wave; ++w) {
		dex += s_wave[0][w];
		pex += s_wave[1][w];
	}
	if (tid == kPackThreads - 1) {
		s_tot[0] = dex + docs;
		s_tot[1] = pex + poss;
	}
	if (tid == 0) {   // rows in order: first compact index and padded base (the first term has at most 4096 sub-terms)
		uint32_t first = 0, base = 0;
		for (uint32_t r = 0; r < p.n_rows0; ++r) {
			const uint32_t c = s_cnt[r];
			s_first[r] = first;
			p.row_base[r] = base;
			p.row_cnt[r] = c;
			p.out_header[4 + r] = c;
			first += c;
			base += (c + 1 + kFtPhraseRowPad - 1) / kFtPhraseRowPad * kFtPhraseRowPad;
		}
	}
	__syncthreads();
	uint32_t ci = dex, po = pex;
	for (uint32_t sl = lo; sl < hi; ++sl) {
		const float proc = p.slot_proc[sl];
		if (proc == 0.0f) continue;
		const uint32_t r = p.slot_row[sl], at = p.row_base[r] + (ci - s_first[r]), np = p.slot_npos[sl];
		p.out_doc[at] = p.slot_doc[sl];
		p.out_rank[at] = proc;
		p.out_field[at] = p.slot_field[sl];
		p.out_pos_off[at] = po;
		const uint64_t* src = p.ws + p.slot_pos[sl];
		for (uint32_t k = 0; k < np; ++k) p.out_fpos[po + k] = src[k];
		// the entry behind a row's last document closes its position run: it is the offset of the next kept document, whatever its row
		if (ci + 1 == s_first[r] + s_cnt[r]) p.out_pos_off[at + 1] = po + np;
		++ci;
		po += np;
	}
	__syncthreads();
	// the range index of every row (first posting with doc >= k * kFtRangeDocs), like rxgpu_ft_set_word_positions builds for a word
	const uint32_t per_row = p.n_ranges + 1;
	for (uint32_t q = tid; q < p.n_rows0 * per_row; q += kPackThreads) {
		const uint32_t r = q / per_row, k = q % per_row;
		const uint32_t* dd = p.out_doc + p.row_base[r];
		const uint64_t bound = uint64_t(k) * kFtRangeDocs;
		uint32_t a = 0, b = s_cnt[r];
		while (a < b) {
			const uint32_t mid = (a + b) >> 1;
			if (uint64_t(dd[mid]) < bound) {
				a = mid + 1;
			} else {
				b = mid;
			}
		}
		p.out_range_off[q] = a;
	}
	if (tid == 0) {
		p.out_header[0] = S;
		p.out_header[1] = p.sync[1];
		p.out_header[2] = s_tot[0];
		p.out_header[3] = s_tot[1];
	}
}

hipError_t launch_ft_phrase_admit(const FtPhrasePlan& p, hipStream_t st) {
	if (!p.grid_blocks) return hipSuccess;
	hipLaunchKernelGGL(ft_phrase_admit, dim3(p.grid_blocks), dim3(256), 0, st, p);
	return hipGetLastError();
}
hipError_t launch_ft_phrase_docs(const FtPhrasePlan& p, uint32_t admitted, hipStream_t st) {
	if (!admitted) return hipSuccess;
	hipLaunchKernelGGL(ft_phrase_docs, dim3((admitted + 63) / 64), dim3(64), 0, st, p, admitted);
	return hipGetLastError();
}
hipError_t launch_ft_phrase_pack(const FtPhrasePlan& p, hipStream_t st) {
	hipLaunchKernelGGL(ft_phrase_pack, dim3(1), dim3(kPackThreads), 0, st, p);
	return hipGetLastError();
}

}  // namespace rxgpu
