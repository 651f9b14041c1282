// Row-subset plumbing of the pre-filtered KNN search (`WHERE cond AND KNN(...)`): an "allowed rows" bitmap
// (bit r of word r / 32 = internal row r, the restrictingMask_ analogue of ft/mergerimpl.h for vectors) is turned into the strictly
// increasing row list knn_scan_subset (knn_scan.hip) walks.  Three small kernels, deterministic and ordered by construction:
//   bitmap_count    popcount of each tile of 1024 words (32 768 rows)         -> tile_sum[tiles]
//   bitmap_offsets  exclusive prefix over the tiles (one workgroup)           -> tile_off[tiles], total
//   bitmap_expand   each tile re-reads its words (L2-resident), prefix over its popcounts in LDS, writes its rows in order
// The bitmap costs N / 8 bytes on the wire (1.25 MB at 10M rows) where a dense id list would cost 4 N.
#include <algorithm>

#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {

constexpr int kTileThreads = 256;
constexpr int kWordsPerThread = 4;
constexpr int kTileWords = kTileThreads * kWordsPerThread;   // 1024 words = 32 768 rows

// word w of the bitmap with the bits at and above row n cleared
__device__ __forceinline__ uint32_t bitmap_word(const uint32_t* __restrict__ words, uint64_t w, uint64_t nwords, uint64_t n) {
	if (w >= nwords) return 0u;
	uint32_t v = words[w];
	const uint64_t lo = w * 32;
	if (lo + 32 > n) v = lo >= n ? 0u : (v & ((1u << uint32_t(n - lo)) - 1u));
	return v;
}

__global__ __launch_bounds__(kTileThreads) void bitmap_count(const uint32_t* __restrict__ words, uint64_t nwords, uint64_t n,
															 uint32_t* __restrict__ tile_sum) {
	__shared__ uint32_t s_part[kTileThreads / kWave];
	const uint64_t w0 = uint64_t(blockIdx.x) * kTileWords + threadIdx.x * kWordsPerThread;
	uint32_t c = 0;
#pragma unroll
	for (int i = 0; i < kWordsPerThread; ++i) c += __popc(bitmap_word(words, w0 + i, nwords, n));
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (int i = 0; i < kTileThreads / kWave; ++i) t += s_part[i];
		tile_sum[blockIdx.x] = t;
	}
}

// one workgroup; tiles <= 2^32 / 32768 = 131 072, so a serial carry over 256-wide strips is a handful of iterations
__global__ __launch_bounds__(kTileThreads) void bitmap_offsets(const uint32_t* __restrict__ tile_sum, uint32_t tiles, uint32_t* __restrict__ tile_off,
															   unsigned long long* __restrict__ total) {
	__shared__ uint32_t s_scan[kTileThreads];
	__shared__ unsigned long long s_carry;
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < tiles; base += kTileThreads) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < tiles ? tile_sum[i] : 0u;
		s_scan[threadIdx.x] = v;
		__syncthreads();
		for (int o = 1; o < kTileThreads; o <<= 1) {   // Hillis-Steele inclusive scan
			const uint32_t add = threadIdx.x >= uint32_t(o) ? s_scan[threadIdx.x - o] : 0u;
			__syncthreads();
			s_scan[threadIdx.x] += add;
			__syncthreads();
		}
		const unsigned long long carry = s_carry;
		if (i < tiles) tile_off[i] = uint32_t(carry + s_scan[threadIdx.x] - v);
		__syncthreads();
		if (threadIdx.x == kTileThreads - 1) s_carry = carry + s_scan[threadIdx.x];
		__syncthreads();
	}
	if (threadIdx.x == 0) *total = s_carry;
}

__global__ __launch_bounds__(kTileThreads) void bitmap_expand(const uint32_t* __restrict__ words, uint64_t nwords, uint64_t n,
															  const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ out_rows, uint64_t cap) {
	__shared__ uint32_t s_scan[kTileThreads];
	const uint64_t w0 = uint64_t(blockIdx.x) * kTileWords + threadIdx.x * kWordsPerThread;
	uint32_t v[kWordsPerThread];
	uint32_t c = 0;
#pragma unroll
	for (int i = 0; i < kWordsPerThread; ++i) {
		v[i] = bitmap_word(words, w0 + i, nwords, n);
		c += __popc(v[i]);
	}
	s_scan[threadIdx.x] = c;
	__syncthreads();
	for (int o = 1; o < kTileThreads; o <<= 1) {
		const uint32_t add = threadIdx.x >= uint32_t(o) ? s_scan[threadIdx.x - o] : 0u;
		__syncthreads();
		s_scan[threadIdx.x] += add;
		__syncthreads();
	}
	uint64_t pos = uint64_t(tile_off[blockIdx.x]) + s_scan[threadIdx.x] - c;   // exclusive prefix of this thread inside the tile
#pragma unroll
	for (int i = 0; i < kWordsPerThread; ++i) {
		uint32_t bits = v[i];
		const uint32_t row0 = uint32_t((w0 + i) * 32);
		while (bits) {
			const int b = __builtin_ctz(bits);
			bits &= bits - 1;
			if (pos < cap) out_rows[pos] = row0 + uint32_t(b);
			++pos;
		}
	}
}

// out[i] = src[idx[i]] (positions of the radix select -> rows of the list)
__global__ __launch_bounds__(256) void gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n,
												  uint32_t* __restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = src[idx[i]];
}

// 1 when the list is strictly increasing and below `limit`, else 0 (device-resident lists cannot be checked on the host)
__global__ __launch_bounds__(256) void check_row_list(const uint32_t* __restrict__ ids, uint64_t n, uint64_t limit, uint32_t* __restrict__ bad) {
	const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
	for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
		const uint32_t v = ids[i];
		if (v >= limit || (i > 0 && ids[i - 1] >= v)) atomicOr(bad, 1u);
	}
}

uint32_t bitmap_tiles(uint64_t n_rows) {
	const uint64_t nwords = (n_rows + 31) / 32;
	return uint32_t((nwords + kTileWords - 1) / kTileWords);
}

// words: [ceil(n_rows / 32)] device; tile_scratch: 2 * bitmap_tiles(n_rows) uint32.  Phase 1 leaves the number of set bits in *total
// (device); the caller reads it back, sizes out_rows, and runs phase 2 over the same scratch.
void launch_bitmap_count(const uint32_t* words, uint64_t n_rows, uint32_t* tile_scratch, unsigned long long* total, hipStream_t s) {
	const uint64_t nwords = (n_rows + 31) / 32;
	const uint32_t tiles = bitmap_tiles(n_rows);
	hipLaunchKernelGGL(bitmap_count, dim3(tiles), dim3(kTileThreads), 0, s, words, nwords, n_rows, tile_scratch);
	hipLaunchKernelGGL(bitmap_offsets, dim3(1), dim3(kTileThreads), 0, s, tile_scratch, tiles, tile_scratch + tiles, total);
}
void launch_bitmap_expand(const uint32_t* words, uint64_t n_rows, const uint32_t* tile_scratch, uint32_t* out_rows, uint64_t cap, hipStream_t s) {
	const uint64_t nwords = (n_rows + 31) / 32;
	const uint32_t tiles = bitmap_tiles(n_rows);
	hipLaunchKernelGGL(bitmap_expand, dim3(tiles), dim3(kTileThreads), 0, s, words, nwords, n_rows, tile_scratch + tiles, out_rows, cap);
}

// IVF: the rows of the probed inverted lists marked in the allowed-rows bitmap (lists are disjoint: every bit is set once; the bitmap ->
// row-list kernels above then produce the ascending list the subset scan wants).  One workgroup per probed list; the list numbers and
// their count are on the device (the coarse search's output), nothing goes through the host.
__global__ __launch_bounds__(256) void ivf_mark_lists(const uint32_t* probe, const uint32_t* probe_cnt, const uint64_t* list_off, const uint32_t* list_rows,
													   uint32_t* bitmap) {
	if (blockIdx.x >= *probe_cnt) return;
	const uint32_t l = probe[blockIdx.x];
	for (uint64_t i = list_off[l] + threadIdx.x, e = list_off[l + 1]; i < e; i += 256) {
		const uint32_t r = list_rows[i];
		atomicOr(&bitmap[r >> 5], 1u << (r & 31));
	}
}
void launch_ivf_mark_lists(const uint32_t* probe, const uint32_t* probe_cnt, uint32_t nprobe, const uint64_t* list_off, const uint32_t* list_rows,
						   uint32_t* bitmap, hipStream_t s) {
	hipLaunchKernelGGL(ivf_mark_lists, dim3(nprobe), dim3(256), 0, s, probe, probe_cnt, list_off, list_rows, bitmap);
}

void launch_gather_u32(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* out, hipStream_t s) {
	hipLaunchKernelGGL(gather_u32, dim3((n + 255) / 256), dim3(256), 0, s, src, idx, n, out);
}

void launch_check_row_list(const uint32_t* ids, uint64_t n, uint64_t limit, uint32_t* bad, int cus, hipStream_t s) {
	const uint32_t gx = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, uint64_t(cus) * 8)));
	hipLaunchKernelGGL(check_row_list, dim3(gx), dim3(256), 0, s, ids, n, limit, bad);
}

}  // namespace rxgpu
