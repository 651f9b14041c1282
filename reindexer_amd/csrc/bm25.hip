// ft_fast BM25 score accumulation on gfx950.
//
// Replaces the per-posting loop of Merger::mergeSimple (cpp_src/core/ft/ft_fast/mergerimpl.h:194-250) and its callee
// calcTermRankImpl (phrasemergerimpl.h:13-81) with Bm25Rx (bm25.h:8-36): for every posting of every sub-term
//   rank = fieldBoost * bound(bm25) * bound(termLenBoost) * bound(pos2rank(firstPos)), max over the posting's fields,
//          [+ geometric sum over needSumRank fields], * opts.boost * subterm.proc
// and per document  proc = max over postings (the FIRST maximum wins, it also decides `field`).
//
// Layout (flattened on the host at CommitFulltext time): one posting = doc id + a run of
// (field, tf, firstPos) entries, SoA.  One thread per posting: 4 B doc + 4 B entry offset + 9 B per entry streamed,
// 4 B words-in-field gathered, one 8-byte atomicMax into the dense per-document score word
// (rank bits << 32 | ~sequence position) and one 4-byte atomicMin of the first sequence position — HBM/atomic bound.
// The arithmetic keeps the reference's types (fp64 Bm25Rx, float bound()) so ranks are bit-identical to the CPU merger;
// log() for the IDF is evaluated once per sub-term on the host.
//
// mergeLimit semantics (docs are admitted in (sub-term, posting) order until maxMergedDocs): the "add events"
// (first valid posting of a doc) are compacted in sequence order by a 3-kernel block scan, cut at maxMergedDocs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rxgpu_internal.h"
#include "ft_rank.hip.h"

namespace rxgpu {

__global__ __launch_bounds__(256) void bm25_score(FtMergeParams p, FtSubterm s) {
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= s.n) return;
	const uint32_t d = s.doc[i];
	const uint32_t gp = uint32_t(s.gp_base + i);
	p.pfield[gp] = 0xFF;   // 0xFF = not a valid posting (excluded / removed / zero rank)
	if ((p.excluded && p.excluded[d]) || (p.removed && p.removed[d])) return;
	uint8_t field;
	const float rank = ft_term_rank(p, s, s.ent_off[i], s.ent_off[i + 1], d, &field);
	if (rank == 0.0f) return;
	p.pfield[gp] = field;
	// rank > 0 => its bit pattern is monotone; low word prefers the EARLIEST posting among equal ranks ("md.proc < rank" is strict)
	const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(rank)) << 32) | (0xFFFFFFFFu - gp);
	atomicMax(&p.best[d], key);
	atomicMin(&p.first[d], gp);
}

// All sub-terms in ONE launch: thread gp of the concatenated (sub-term, posting) sequence finds its sub-term by binary search over
// the sequence bases (nsub is small), then scores exactly like bm25_score.
__global__ __launch_bounds__(256) void bm25_score_fused(FtMergeParams p, const FtSubterm* subs, uint32_t nsub, uint64_t total) {
	const uint64_t gp64 = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (gp64 >= total) return;
	uint32_t lo = 0, hi = nsub - 1;
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (subs[mid].gp_base <= gp64) {
			lo = mid;
		} else {
			hi = mid - 1;
		}
	}
	const FtSubterm s = subs[lo];
	const uint64_t i = gp64 - s.gp_base;
	const uint32_t d = s.doc[i];
	const uint32_t gp = uint32_t(gp64);
	p.pfield[gp] = 0xFF;
	if ((p.excluded && p.excluded[d]) || (p.removed && p.removed[d])) return;
	uint8_t field;
	const float rank = ft_term_rank(p, s, s.ent_off[i], s.ent_off[i + 1], d, &field);
	if (rank == 0.0f) return;
	p.pfield[gp] = field;
	const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(rank)) << 32) | (0xFFFFFFFFu - gp);
	atomicMax(&p.best[d], key);
	atomicMin(&p.first[d], gp);
}

// ---- order-preserving compaction of the add events, cut at max_merged ----
constexpr int kScanBlock = 1024;

__global__ __launch_bounds__(256) void bm25_count_adds(FtMergeParams p, FtSubterm s, uint32_t* block_counts) {
	__shared__ uint32_t cnt;
	if (threadIdx.x == 0) cnt = 0;
	__syncthreads();
	const uint64_t base = uint64_t(blockIdx.x) * kScanBlock;
	uint32_t local = 0;
	for (int k = 0; k < kScanBlock / 256; ++k) {
		const uint64_t i = base + k * 256 + threadIdx.x;
		if (i < s.n) {
			const uint32_t gp = uint32_t(s.gp_base + i);
			if (p.pfield[gp] != 0xFF && p.first[s.doc[i]] == gp) ++local;
		}
	}
	atomicAdd(&cnt, local);
	__syncthreads();
	if (threadIdx.x == 0) block_counts[s.block_base + blockIdx.x] = cnt;
}

__global__ __launch_bounds__(1024) void bm25_scan_blocks(uint32_t* block_counts, uint32_t nblocks, uint32_t* total) {
	__shared__ uint32_t part[1024];
	const int t = threadIdx.x;
	const uint32_t per = (nblocks + 1023) / 1024;
	uint32_t sum = 0;
	for (uint32_t k = 0; k < per; ++k) {
		const uint32_t i = t * per + k;
		if (i < nblocks) sum += block_counts[i];
	}
	part[t] = sum;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
		const uint32_t v = t >= off ? part[t - off] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	uint32_t run = part[t] - sum;   // exclusive prefix of this thread's chunk
	for (uint32_t k = 0; k < per; ++k) {
		const uint32_t i = t * per + k;
		if (i < nblocks) {
			const uint32_t c = block_counts[i];
			block_counts[i] = run;
			run += c;
		}
	}
	if (t == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void bm25_emit(FtMergeParams p, FtSubterm s, const uint32_t* block_offsets) {
	__shared__ uint32_t wave_base[4];
	__shared__ uint32_t running;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) running = block_offsets[s.block_base + blockIdx.x];
	__syncthreads();
	const uint64_t base = uint64_t(blockIdx.x) * kScanBlock;
	for (int k = 0; k < kScanBlock / 256; ++k) {
		const uint64_t i = base + k * 256 + threadIdx.x;
		bool add = false;
		uint32_t d = 0;
		if (i < s.n) {
			const uint32_t gp = uint32_t(s.gp_base + i);
			d = s.doc[i];
			add = p.pfield[gp] != 0xFF && p.first[d] == gp;
		}
		const uint64_t m = __ballot(add);
		if (lane == 0) wave_base[wave] = uint32_t(__popcll(m));
		__syncthreads();
		uint32_t off = running;
		for (int w = 0; w < wave; ++w) off += wave_base[w];
		const uint32_t slot = off + uint32_t(__popcll(m & ((1ull << lane) - 1)));
		if (add && slot < p.max_merged) {
			const unsigned long long key = p.best[d];
			p.out_doc[slot] = d;
			p.out_proc[slot] = __uint_as_float(uint32_t(key >> 32));
			p.out_field[slot] = p.pfield[0xFFFFFFFFu - uint32_t(key & 0xFFFFFFFFu)];
		}
		__syncthreads();
		if (threadIdx.x == 0) running += wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
		__syncthreads();
	}
}

void launch_bm25_score_fused(const FtMergeParams& p, const FtSubterm* d_subs, uint32_t nsub, uint64_t total, hipStream_t st) {
	if (total == 0) return;
	hipLaunchKernelGGL(bm25_score_fused, dim3(uint32_t((total + 255) / 256)), dim3(256), 0, st, p, d_subs, nsub, total);
}
void launch_bm25_score(const FtMergeParams& p, const FtSubterm& s, hipStream_t st) {
	if (s.n == 0) return;
	hipLaunchKernelGGL(bm25_score, dim3(uint32_t((s.n + 255) / 256)), dim3(256), 0, st, p, s);
}
uint32_t bm25_scan_blocks_for(uint64_t n) { return uint32_t((n + kScanBlock - 1) / kScanBlock); }
void launch_bm25_count_adds(const FtMergeParams& p, const FtSubterm& s, uint32_t* block_counts, hipStream_t st) {
	if (s.n == 0) return;
	hipLaunchKernelGGL(bm25_count_adds, dim3(bm25_scan_blocks_for(s.n)), dim3(256), 0, st, p, s, block_counts);
}
void launch_bm25_scan_blocks(uint32_t* block_counts, uint32_t nblocks, uint32_t* total, hipStream_t st) {
	hipLaunchKernelGGL(bm25_scan_blocks, dim3(1), dim3(1024), 0, st, block_counts, nblocks, total);
}
void launch_bm25_emit(const FtMergeParams& p, const FtSubterm& s, const uint32_t* block_offsets, hipStream_t st) {
	if (s.n == 0) return;
	hipLaunchKernelGGL(bm25_emit, dim3(bm25_scan_blocks_for(s.n)), dim3(256), 0, st, p, s, block_offsets);
}

}  // namespace rxgpu
