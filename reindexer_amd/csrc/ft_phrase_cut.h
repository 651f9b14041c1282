// PhraseMerger's admission cut over document-range shards (SURVEY 8e "BM25"; phrasemerger.h:341, phrasemergerimpl.h:181-183, 209-215).
//
// The reference adds a document to a phrase's mergeData_ by its first posting of the phrase's FIRST term, in (sub-term row, document)
// order, while fewer than maxMergedDocs_ = min(mergeLimit, the first term's documents) were added.  Over shards every shard runs the
// admission over its own fragments (ft_phrase_admit) and reports, row by row, how many candidates it admitted under its LOCAL bound
// min(mergeLimit, its fragment's postings).  The cut of the whole index walks the rows in order and, inside a row, the shards in order
// (a shard's documents lie before the next shard's): the first mergeLimit candidates stay.  What a shard keeps is a prefix of its own
// slots (its rows before the cut row in full, a prefix of the cut row), so the result is one number per shard.
//
// A shard's count may be truncated by its local bound; that cannot change the outcome: a truncated shard alone fills what was left of
// mergeLimit at that point, so every later candidate is cut either way (tests/test_ft_phrase_cut.py replays this against the plain rule).
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rxgpu {

// row_admitted[s]: shard s's admitted candidates per row (n_rows entries), or empty when the shard admitted nothing at all.
inline std::vector<uint64_t> ft_shard_phrase_cut(const std::vector<std::vector<uint32_t>>& row_admitted, size_t n_rows, uint64_t merge_limit) {
	std::vector<uint64_t> keep(row_admitted.size(), 0);
	uint64_t taken = 0;
	for (size_t j = 0; j < n_rows; ++j) {
		for (size_t s = 0; s < row_admitted.size(); ++s) {
			const uint64_t c = row_admitted[s].empty() ? 0 : row_admitted[s][j];
			const uint64_t take = std::min<uint64_t>(c, merge_limit > taken ? merge_limit - taken : 0);
			keep[s] += take;
			taken += take;
		}
	}
	return keep;
}

}  // namespace rxgpu
