// Device-side calcTermRankImpl (cpp_src/core/ft/ft_fast/phrasemergerimpl.h:13-81) with the three calculators behind Bm25Calculator<BM>
// (core/ft/bm25.h:8-68: Bm25Rx, Bm25Classic, TermCount — selected per merge like the selecter does from bm25Config.bm25Type,
// selecterimpl.h:615-624) and the FTFieldConfig helpers (core/ft/config/ftconfig.h:127-148), used by the merge kernels (ft_merge.hip).
// P supplies the term / field configuration (FtTermCfg member names), S the posting arrays (FtPosSubterm; s.idf is the calculator's
// IDF, computed on the host per sub-term).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rxgpu {

constexpr int kFtBm25Rx = 0, kFtBm25Classic = 1, kFtBm25WordCount = 2;   // rxgpu_ft_config::bm25_type

__device__ __forceinline__ float ft_pos2rank(unsigned pos) {   // ftconfig.h:127-144
	if (pos <= 10) return float(1.0 - (pos / 100.0));
	if (pos <= 100) return float(0.9 - (pos / 1000.0));
	if (pos <= 1000) return float(0.8 - (pos / 10000.0));
	if (pos <= 10000) return float(0.7 - (pos / 100000.0));
	if (pos <= 100000) return float(0.6 - (pos / 1000000.0));
	return 0.5f;
}
__device__ __forceinline__ float ft_bound(float k, float weight, float boost) {   // ftconfig.h:146
	return float((1.0 - double(weight)) + double(k * boost * weight));
}

// calcTermRankImpl for one posting; returns rank, *field = field with the max rank
template <typename P, typename S>
__device__ __forceinline__ float ft_term_rank(const P& p, const S& s, uint32_t e0, uint32_t e1, uint32_t doc, uint8_t* field) {
	uint8_t best_field = 0;
	float term_rank = 0.f;
	float ranks[8];
	int nranks = 0;
	bool sum_winner = false;
	const float* w = p.words + size_t(doc) * p.num_fields;
	for (uint32_t e = e0; e < e1; ++e) {
		const unsigned f = s.ent_field[e];
		const float fb = p.field_boost[f];
		if (fb == 0.0f) continue;
		const double cnt = double(s.ent_tf[e]), wif = double(w[f]);
		double bm;
		if (p.bm25_type == kFtBm25WordCount) {   // TermCount::Get (bm25.h:58-68)
			bm = cnt;
		} else {                                  // Bm25Rx::Get / Bm25Classic::Get: same expression, TF = count (rx) or count / wordsInDoc (classic)
			const double tf = p.bm25_type == kFtBm25Classic ? cnt / wif : cnt;
			bm = s.idf * tf * (p.k1 + 1.0) / (tf + p.k1 * (1.0 - p.b + p.b * wif / double(p.avg_words[f])));
		}
		const float bm25 = float(bm);
		const float norm = ft_bound(bm25, p.bm25_weight[f], p.bm25_boost[f]);
		const float prank = ft_bound(ft_pos2rank(s.ent_first_pos[e]), p.position_weight[f], p.position_boost[f]);
		const float tlb = ft_bound(p.term_len_boost_in, p.term_len_weight[f], p.term_len_boost[f]);
		const float tmp = fb * norm * tlb * prank;
		if (tmp > term_rank) {
			best_field = uint8_t(f);
			term_rank = tmp;
			sum_winner = p.need_sum_rank[f] != 0;
		}
		if (p.need_sum_rank[f] && nranks < 8) ranks[nranks++] = tmp;
	}
	if (term_rank > 0.0f && p.summation_ratio > 0.0) {
		for (int i = 1; i < nranks; ++i) {   // descending insertion sort (<= 8 fields with needSumRank)
			const float v = ranks[i];
			int j = i;
			while (j > 0 && ranks[j - 1] < v) {
				ranks[j] = ranks[j - 1];
				--j;
			}
			ranks[j] = v;
		}
		float k = float(p.summation_ratio);
		for (int i = sum_winner ? 1 : 0; i < nranks; ++i) {
			term_rank += (k * ranks[i]);
			k = float(double(k) * p.summation_ratio);
		}
	}
	*field = best_field;
	return p.opts_boost * s.proc * term_rank;
}

}  // namespace rxgpu
