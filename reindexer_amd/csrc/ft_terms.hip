// ft_fast multi-term merge on gfx950: Merger::Merge for queries that are not Simple() (cpp_src/core/ft/ft_fast/mergerimpl.h:466-566),
// terms only (phrases and multi-word synonyms stay on the CPU merger).
//
//   buildRestrictingBitmask  (mergerimpl.h:326-384)  -> ft_mask_init / ft_term_mask / ft_mask_and / ft_mask_exclude
//   preselectMostRelevantDocs (:386-464) + calcTermScores (:289-324) -> ft_prescore / ft_prescore_finalize / ft_preselect_pick / ft_preselect_apply
//   mergeTerm (:107-192) + PositionsDistance (:20-37) + switchToNextWord (merger.h:218-226) -> ft_term_pass
//
// The reference loop is sequential and order dependent in three places; each is reproduced exactly:
//  * admission: documents are added in (term, sub-term, posting) order until maxMergedDocs.  Within one sub-term every posting is
//    a different document, so one launch per sub-term is race free; new documents get their merge slot from an ORDERED prefix
//    count across the launch (decoupled look-back over ticket-ordered workgroups), cut at maxMergedDocs;
//  * per document `proc -= rank; proc += finalRank` on every strict improvement, sub-term after sub-term: launches are
//    stream-ordered, the float operations are the reference's, so the bits are too;
//  * preselect keeps, among documents tied at the threshold score, the first ones in document order: ordered prefix again.
// switchToNextWord is applied lazily (the first time a term touches a slot): between two touches the eager loop is idempotent.
//
// Bound: HBM / atomics: per posting 4 B doc + 8 B entry offsets + 9 B per (field, tf, firstPos) entry + 8 B position
// offsets + 8 B per position streamed; 4 B words-in-field, 4 B slot index and the mask word gathered; ~40 B of slot state
// read-modify-written for documents that are already merged.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "rxgpu_internal.h"
#include "ft_rank.hip.h"

namespace rxgpu {

namespace {

constexpr unsigned long long kLbPrefix = 1ull << 63;
constexpr unsigned long long kLbAggregate = 1ull << 62;
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

__device__ __forceinline__ bool mask_bit(const uint32_t* m, uint32_t d) { return (m[d >> 5] >> (d & 31)) & 1u; }

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

template <typename S>
__device__ __forceinline__ const S& find_subterm(const S* subs, uint32_t nsub, uint64_t gp) {
	uint32_t lo = 0, hi = nsub - 1;
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (subs[mid].gp_base <= gp) {
			lo = mid;
		} else {
			hi = mid - 1;
		}
	}
	return subs[lo];
}

}  // namespace

// ---------------------------------------------------------------------------------------------- restricting bitmask
// restrictingMask_ = ~docsExcluded_ (mergerimpl.h:328-330); bits past total_docs stay 0 so that PopCount() is exact
__global__ __launch_bounds__(256) void ft_mask_init(uint32_t* mask, const uint8_t* excluded, uint64_t total_docs) {
	const uint64_t w = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const uint64_t d0 = w * 32;
	if (d0 >= total_docs) return;
	uint32_t bits = 0;
	const uint32_t cnt = uint32_t(total_docs - d0 < 32 ? total_docs - d0 : 32);
	if (!excluded) {
		bits = cnt == 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u);
	} else {
		for (uint32_t b = 0; b < cnt; ++b) bits |= (excluded[d0 + b] ? 0u : 1u) << b;
	}
	mask[w] = bits;
}

// calcTermBitmask (mergerimpl.h:252-274): any occurrence with a relevant field (checkFieldsRelevance, phrasemergerimpl.h:93-125)
__global__ __launch_bounds__(256) void ft_term_mask(const FtPosSubterm* subs, uint32_t nsub, uint64_t total, const float* field_boost, uint32_t num_fields,
													uint32_t* term_mask) {
	const uint64_t gp = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (gp >= total) return;
	bool all_pos = true;
	for (uint32_t f = 0; f < num_fields; ++f) all_pos = all_pos && field_boost[f] != 0.0f;
	const FtPosSubterm& s = find_subterm(subs, nsub, gp);
	const uint64_t i = gp - s.gp_base;
	bool rel = all_pos;
	if (!rel) {
		for (uint32_t e = s.ent_off[i], e1 = s.ent_off[i + 1]; e < e1 && !rel; ++e) rel = field_boost[s.ent_field[e]] != 0.0f;
	}
	if (rel) {
		const uint32_t d = s.doc[i];
		atomicOr(&term_mask[d >> 5], 1u << (d & 31));
	}
}

__global__ __launch_bounds__(256) void ft_mask_and(uint32_t* mask, const uint32_t* term_mask, uint64_t nwords) {
	const uint64_t w = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (w < nwords) mask[w] &= term_mask[w];
}

// excludeTermFromBitmask (mergerimpl.h:276-287)
__global__ __launch_bounds__(256) void ft_mask_exclude(const FtPosSubterm* subs, uint32_t nsub, uint64_t total, uint32_t* mask) {
	const uint64_t gp = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (gp >= total) return;
	const FtPosSubterm& s = find_subterm(subs, nsub, gp);
	const uint32_t d = s.doc[gp - s.gp_base];
	atomicAnd(&mask[d >> 5], ~(1u << (d & 31)));
}

__global__ __launch_bounds__(256) void ft_mask_popcount(const uint32_t* mask, uint64_t nwords, uint32_t* out) {
	uint32_t c = 0;
	for (uint64_t w = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords; w += uint64_t(gridDim.x) * blockDim.x) c += __popc(mask[w]);
	c = wave_sum(c);
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ---------------------------------------------------------------------------------------------- preselect
// calcTermScores (mergerimpl.h:289-324) for ONE sub-term (documents are unique inside it; sub-terms and terms run in stream order)
__global__ __launch_bounds__(256) void ft_prescore(FtPosSubterm s, const uint32_t* mask, uint32_t* term_mask, uint16_t* score, const float* field_boost,
												   uint32_t same_boost, float opts_boost) {
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= s.n) return;
	const uint32_t d = s.doc[i];
	if (!mask_bit(mask, d)) return;
	float mb = field_boost[0];
	if (!same_boost) {   // maxFieldsBoost (phrasemergerimpl.h:127-160)
		mb = 0.0f;
		for (uint32_t e = s.ent_off[i], e1 = s.ent_off[i + 1]; e < e1; ++e) mb = fmaxf(mb, field_boost[s.ent_field[e]]);
	}
	if (mb > 0.0f && !mask_bit(term_mask, d)) {
		const float proc = s.proc * mb * opts_boost;
		uint32_t p16 = uint32_t(int32_t(proc)) & 0xFFFFu;   // static_cast<uint16_t>(float) as x86 evaluates it
		p16 = p16 < 65535u / 4 ? p16 : 65535u / 4;
		const uint32_t cur = score[d];
		p16 = p16 < 65535u - cur ? p16 : 65535u - cur;
		score[d] = uint16_t(cur + p16);
		atomicOr(&term_mask[d >> 5], 1u << (d & 31));
	}
}

// mergerimpl.h:416-423: zero the score of masked-out / removed documents, histogram of the rest.  Scores take few distinct values,
// so the counts are aggregated per wave, then per workgroup in a small LDS table, and only then added to the global histogram.
__global__ __launch_bounds__(256) void ft_prescore_finalize(FtPreselect p) {
	__shared__ uint32_t keys[256];
	__shared__ uint32_t cnts[256];
	keys[threadIdx.x] = 0;   // a score of 0 is never inserted
	cnts[threadIdx.x] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	for (uint64_t base = uint64_t(blockIdx.x) * 256; base < p.total_docs; base += uint64_t(gridDim.x) * 256) {
		const uint64_t d = base + threadIdx.x;
		uint32_t sc = 0;
		if (d < p.total_docs) {
			sc = p.score[d];
			if (sc && (!mask_bit(p.mask, uint32_t(d)) || (p.removed && p.removed[d]))) {
				sc = 0;
				p.score[d] = 0;
			}
		}
		unsigned long long todo = __ballot(sc != 0);
		while (todo) {
			const int leader = __ffsll((long long)todo) - 1;
			const uint32_t v = __shfl(sc, leader, 64);
			const unsigned long long same = __ballot(sc == v);
			if (lane == leader) {
				const uint32_t c = uint32_t(__popcll(same));
				uint32_t h = (v * 2654435761u) >> 24;
				int probes = 0;
				for (; probes < 256; ++probes, h = (h + 1) & 255u) {
					const uint32_t old = atomicCAS(&keys[h], 0u, v);
					if (old == 0u || old == v) {
						atomicAdd(&cnts[h], c);
						break;
					}
				}
				if (probes == 256) atomicAdd(&p.hist[v], c);   // more than 256 distinct scores in one workgroup
			}
			todo &= ~same;
		}
	}
	__syncthreads();
	if (keys[threadIdx.x]) atomicAdd(&p.hist[keys[threadIdx.x]], cnts[threadIdx.x]);
}

// mergerimpl.h:433-446: walk the scores downwards until maxMergedDocs documents are covered
__global__ __launch_bounds__(1024) void ft_preselect_pick(FtPreselect p) {
	__shared__ unsigned long long suffix[1024];
	__shared__ uint32_t s_min;
	const int t = threadIdx.x;
	unsigned long long chunk = 0;
	for (int b = 0; b < 64; ++b) chunk += p.hist[t * 64 + b];
	suffix[t] = chunk;
	if (t == 0) s_min = 0xFFFFFFFFu;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {   // inclusive suffix sums
		const unsigned long long v = t + off < 1024 ? suffix[t + off] : 0;
		__syncthreads();
		suffix[t] += v;
		__syncthreads();
	}
	unsigned long long above = suffix[t] - chunk;   // documents with a score in a higher chunk
	// a score sc is visited iff the documents strictly above it are fewer than maxMergedDocs; minScore = lowest visited score >= 1
	uint32_t lowest = 0xFFFFFFFFu;
	unsigned long long lowest_above = 0;
	for (int b = 63; b >= 0; --b) {
		const uint32_t sc = uint32_t(t * 64 + b);
		if (sc >= 1 && above < p.max_merged) {
			lowest = sc;
			lowest_above = above;
		}
		above += p.hist[sc];
	}
	if (lowest != 0xFFFFFFFFu) atomicMin(&s_min, lowest);
	__syncthreads();
	if (s_min == 0xFFFFFFFFu) {
		if (t == 0) {
			p.pick[0] = 65535u;
			p.pick[1] = 0;
		}
	} else if (lowest == s_min) {
		p.pick[0] = lowest;
		p.pick[1] = uint32_t(p.max_merged - lowest_above);
	}
}

// mergerimpl.h:448-462: one thread per mask word; ties at minScore are kept in document order up to minScoreDocs
__global__ __launch_bounds__(256) void ft_preselect_apply(FtPreselect p, uint64_t nwords) {
	const uint32_t ticket = grab_ticket(p.ticket);
	const uint64_t w = uint64_t(ticket) * 256 + threadIdx.x;
	const uint32_t min_score = p.pick[0], min_docs = p.pick[1];
	uint32_t bits = 0, gt = 0, tie = 0;
	if (w < nwords) {
		bits = p.mask[w];
		const uint64_t d0 = w * 32;
		for (uint32_t b = 0; b < 32; ++b) {
			if (!((bits >> b) & 1u)) continue;   // only masked-in documents are inspected; d0 + b < total_docs by construction
			const uint32_t sc = p.score[d0 + b];
			gt |= uint32_t(sc > min_score) << b;
			tie |= uint32_t(sc == min_score) << b;
		}
	}
	uint32_t grand;
	const uint32_t excl = ordered_prefix(__popc(tie), ticket, p.lookback, p.error_flag, &grand);
	if (w < nwords) {
		uint32_t allowed = min_docs > excl ? min_docs - excl : 0;
		uint32_t keep = gt;
		while (tie && allowed) {
			const uint32_t low = tie & (0u - tie);
			keep |= low;
			tie ^= low;
			--allowed;
		}
		if (keep != bits) p.mask[w] = keep;
	}
}

// ---------------------------------------------------------------------------------------------- mergeTerm
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

__global__ __launch_bounds__(256) void ft_term_pass(FtTermPass p) {
	const uint32_t ticket = grab_ticket(p.ticket);
	const uint32_t base_docs = *p.num_docs_in;
	const bool full = base_docs >= p.max_merged;
	const uint64_t i0 = uint64_t(ticket) * kFtPassBlock + uint64_t(threadIdx.x) * kFtPassItems;
	float c_rank[kFtPassItems];
	uint8_t c_field[kFtPassItems];
	uint32_t c_mask = 0;
	// The gathers of one posting form a dependent chain (doc -> mask word -> slot -> removed flag); the four postings of a thread
	// are independent, so each stage is issued for all four before anything is consumed.
	uint32_t docs[kFtPassItems], slots_of[kFtPassItems];
	bool live[kFtPassItems];
	static_assert(kFtPassItems == 4, "the vector load below reads four document ids");
	if (i0 + kFtPassItems <= p.sub.n) {
		const uint4 v = *reinterpret_cast<const uint4*>(p.sub.doc + i0);   // i0 % 4 == 0 and the list is 256-byte aligned
		docs[0] = v.x;
		docs[1] = v.y;
		docs[2] = v.z;
		docs[3] = v.w;
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) live[k] = true;
	} else {
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) {
			live[k] = i0 + k < p.sub.n;
			docs[k] = live[k] ? p.sub.doc[i0 + k] : 0u;
		}
	}
	{
		uint32_t mw[kFtPassItems];
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) mw[k] = live[k] ? p.mask[docs[k] >> 5] : 0u;
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) live[k] = live[k] && ((mw[k] >> (docs[k] & 31)) & 1u);   // restrictingMask_
	}
#pragma unroll
	for (int k = 0; k < kFtPassItems; ++k) slots_of[k] = live[k] ? p.slot_of[docs[k]] : kNoSlot;
#pragma unroll
	for (int k = 0; k < kFtPassItems; ++k) live[k] = live[k] && !(slots_of[k] == kNoSlot && full);   // !docAdded && numDocs() >= maxMergedDocs_
	if (p.removed) {
		uint8_t rm[kFtPassItems];
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) rm[k] = live[k] ? p.removed[docs[k]] : uint8_t(0);
#pragma unroll
		for (int k = 0; k < kFtPassItems; ++k) live[k] = live[k] && !rm[k];
	}
#pragma unroll
	for (int k = 0; k < kFtPassItems; ++k) {
		const uint64_t i = i0 + k;
		c_rank[k] = 0.f;
		c_field[k] = 0;
		if (!live[k]) continue;
		const uint32_t d = docs[k];
		const uint32_t slot = slots_of[k];
		uint8_t field;
		const float rank = ft_term_rank(p.cfg, p.sub, p.sub.ent_off[i], p.sub.ent_off[i + 1], d, &field);
		if (rank == 0.0f) continue;
		if (slot == kNoSlot) {
			c_rank[k] = rank;
			c_field[k] = field;
			c_mask |= 1u << k;
			continue;
		}
		if (p.simple) {   // mergeSimple, mergerimpl.h:234-239: strict <, so the first maximum (and its field) wins
			if (p.slots.proc[slot] < rank) {
				p.slots.proc[slot] = rank;
				p.slots.field[slot] = field;
			}
			continue;
		}
		// ---- document already merged: mergerimpl.h:171-189
		const FtSlots& s = p.slots;
		const uint64_t* pos = p.sub.fpos + p.sub.pos_off[i];
		const uint32_t npos = p.sub.pos_off[i + 1] - p.sub.pos_off[i];
		float cur_rank = s.rank[slot];
		if (s.switched_term[slot] < p.qp_idx) {                // switchToNextWord, applied on first touch
			const uint32_t nn = s.next_cnt[slot];
			if (nn) {
				s.last_ptr[slot] = s.next_ptr[slot];
				s.last_cnt[slot] = nn;
				s.next_cnt[slot] = 0;
				cur_rank = 0.f;
				s.rank[slot] = 0.f;
			}
			s.switched_term[slot] = p.qp_idx;
		}
		if (s.last_counted[slot] < p.qp_idx) {                 // InreaseTermsCounter
			s.terms_counter[slot] = uint16_t(s.terms_counter[slot] + 1);
			s.last_counted[slot] = p.qp_idx;
		}
		unsigned dist = ft_positions_distance(s.last_ptr[slot], s.last_cnt[slot], pos, npos);
		dist = dist > 1u ? dist : 1u;
		const float norm_dist = ft_bound(float(1.0 / double(float(dist))), p.distance_weight, p.distance_boost);
		const float final_rank = norm_dist * rank;
		if (final_rank > cur_rank) {
			float pr = s.proc[slot];
			pr -= cur_rank;
			pr += final_rank;
			s.proc[slot] = pr;
			s.next_ptr[slot] = pos;
			s.next_cnt[slot] = npos;
			s.rank[slot] = final_rank;
		}
	}
	// ---- new documents: ordered slots, cut at maxMergedDocs (addDoc, merger.h:161-180)
	uint32_t grand;
	uint32_t slot = base_docs + ordered_prefix(__popc(c_mask), ticket, p.lookback, p.error_flag, &grand);
#pragma unroll
	for (int k = 0; k < kFtPassItems; ++k) {
		if (!((c_mask >> k) & 1u)) continue;
		if (slot < p.max_merged) {
			const uint64_t i = i0 + k;
			const uint32_t d = p.sub.doc[i];
			const FtSlots& s = p.slots;
			s.doc[slot] = d;
			s.proc[slot] = c_rank[k];
			s.field[slot] = c_field[k];
			p.slot_of[d] = slot;
			if (p.simple) {
				++slot;
				continue;
			}
			s.rank[slot] = c_rank[k];
			s.last_ptr[slot] = nullptr;
			s.last_cnt[slot] = 0;
			s.next_ptr[slot] = p.sub.fpos + p.sub.pos_off[i];
			s.next_cnt[slot] = p.sub.pos_off[i + 1] - p.sub.pos_off[i];
			s.switched_term[slot] = p.qp_idx;
			s.last_counted[slot] = p.qp_idx;
			s.terms_counter[slot] = 1;
		}
		++slot;
	}
	if (ticket == gridDim.x - 1 && threadIdx.x == 0) {
		const unsigned long long tot = (unsigned long long)base_docs + grand;
		*p.num_docs_out = tot < p.max_merged ? uint32_t(tot) : p.max_merged;
	}
}

// ---------------------------------------------------------------------------------------------- launchers
static inline dim3 grid_for(uint64_t n, uint32_t per_block = 256) { return dim3(uint32_t((n + per_block - 1) / per_block)); }

void launch_ft_mask_init(uint32_t* mask, const uint8_t* excluded, uint64_t total_docs, hipStream_t st) {
	hipLaunchKernelGGL(ft_mask_init, grid_for((total_docs + 31) / 32), dim3(256), 0, st, mask, excluded, total_docs);
}
void launch_ft_term_mask(const FtPosSubterm* d_subs, uint32_t nsub, uint64_t total, const float* field_boost, uint32_t num_fields, uint32_t* term_mask,
						 hipStream_t st) {
	if (!total) return;
	hipLaunchKernelGGL(ft_term_mask, grid_for(total), dim3(256), 0, st, d_subs, nsub, total, field_boost, num_fields, term_mask);
}
void launch_ft_mask_and(uint32_t* mask, const uint32_t* term_mask, uint64_t nwords, hipStream_t st) {
	hipLaunchKernelGGL(ft_mask_and, grid_for(nwords), dim3(256), 0, st, mask, term_mask, nwords);
}
void launch_ft_mask_exclude(const FtPosSubterm* d_subs, uint32_t nsub, uint64_t total, uint32_t* mask, hipStream_t st) {
	if (!total) return;
	hipLaunchKernelGGL(ft_mask_exclude, grid_for(total), dim3(256), 0, st, d_subs, nsub, total, mask);
}
void launch_ft_mask_popcount(const uint32_t* mask, uint64_t nwords, uint32_t* out, hipStream_t st) {
	const uint32_t blocks = uint32_t(std::min<uint64_t>((nwords + 255) / 256, 256));
	hipLaunchKernelGGL(ft_mask_popcount, dim3(blocks ? blocks : 1), dim3(256), 0, st, mask, nwords, out);
}
void launch_ft_prescore(const FtPosSubterm& sub, const uint32_t* mask, uint32_t* term_mask, uint16_t* score, const float* field_boost, uint32_t,
						bool same_boost, float opts_boost, hipStream_t st) {
	if (!sub.n) return;
	hipLaunchKernelGGL(ft_prescore, grid_for(sub.n), dim3(256), 0, st, sub, mask, term_mask, score, field_boost, same_boost ? 1u : 0u, opts_boost);
}
void launch_ft_preselect(const FtPreselect& p, hipStream_t st) {
	hipLaunchKernelGGL(ft_prescore_finalize, dim3(uint32_t(std::min<uint64_t>((p.total_docs + 255) / 256, 2048))), dim3(256), 0, st, p);
	hipLaunchKernelGGL(ft_preselect_pick, dim3(1), dim3(1024), 0, st, p);
	const uint64_t nwords = (p.total_docs + 31) / 32;
	hipLaunchKernelGGL(ft_preselect_apply, grid_for(nwords), dim3(256), 0, st, p, nwords);
}
void launch_ft_term_pass(const FtTermPass& p, hipStream_t st) {
	if (!p.sub.n) return;
	hipLaunchKernelGGL(ft_term_pass, dim3(ft_pass_blocks(p.sub.n)), dim3(256), 0, st, p);
}

}  // namespace rxgpu
