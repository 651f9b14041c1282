// Large-k path of the brute-force search (k+1 > 64, e.g. the reference bench's k = 1000,
// cpp_src/gtests/bench/fixtures/knn_fixture.cc:32-43): a distance pass that streams the rows once and writes
// dist[n], then an exact radix select of the kk smallest (dist,row) pairs.  Same arithmetic contract as the
// fused scan (knn_kernels.hip.h); the (dist,row) order is the eviction order of the reference's max-heap
// (std::less<pair<float,label>>, priority_queue.h + bruteforce.cc:103-127).
#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {

constexpr int kThreads = 512;
constexpr int kWavesPerBlock = kThreads / kWave;

template <int kMetric>
__global__ __launch_bounds__(kThreads) void knn_all_distances(const float* rows, const float* inv_norms, const float* query, uint64_t n,
															   uint32_t stride, uint32_t dim, float* out) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
	const uint64_t nquads = (n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t nwaves = uint64_t(gridDim.x) * kWavesPerBlock;
	for (uint64_t quad = uint64_t(blockIdx.x) * kWavesPerBlock + wave; quad < nquads; quad += nwaves) {
		const uint64_t row = quad * kRowsPerWave + g;
		const bool valid = row < n;
		const uint64_t rowc = valid ? row : n - 1;
		const float sum = group_distance_generic<kMetric>(rows + rowc * stride, query, dim, m);
		const float dist = metric_epilogue<kMetric>(sum, inv_norms, rowc);
		if (valid && m == 0) out[row] = dist;
	}
}

void launch_all_distances(int metric, const float* rows, const float* inv_norms, const float* query, uint64_t n, uint32_t stride,
						  uint32_t dim, float* out_dist, uint32_t gridx, hipStream_t s) {
	switch (metric) {
		case kL2: hipLaunchKernelGGL((knn_all_distances<kL2>), dim3(gridx), dim3(kThreads), 0, s, rows, inv_norms, query, n, stride, dim, out_dist); break;
		case kIP: hipLaunchKernelGGL((knn_all_distances<kIP>), dim3(gridx), dim3(kThreads), 0, s, rows, inv_norms, query, n, stride, dim, out_dist); break;
		default: hipLaunchKernelGGL((knn_all_distances<kCos>), dim3(gridx), dim3(kThreads), 0, s, rows, inv_norms, query, n, stride, dim, out_dist); break;
	}
}

// ---- radix select on the 64-bit composite key (sortable(dist) << 32 | row), 8 digits of 8 bits, MSB first ----

struct SelectState {
	unsigned long long prefix;      // key bits resolved so far (high digits)
	unsigned long long remaining;   // how many elements are still to be taken among keys matching prefix
	unsigned int hist[8][256];
	unsigned long long out_count;
};

__device__ __forceinline__ unsigned long long composite_key(float d, uint64_t row) {
	if (d == 0.0f) d = 0.0f;   // -0.0 and +0.0 compare equal in the reference; give them one key
	uint32_t u = __float_as_uint(d);
	u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
	return (static_cast<unsigned long long>(u) << 32) | row;
}

__global__ void select_init(SelectState* st, uint32_t kk) {
	const int t = threadIdx.x;
	for (int i = t; i < 8 * 256; i += blockDim.x) (&st->hist[0][0])[i] = 0;
	if (t == 0) {
		st->prefix = 0;
		st->remaining = kk;
		st->out_count = 0;
	}
}

// digit = 7 (most significant) .. 0
__global__ __launch_bounds__(kThreads) void select_histogram(const float* dist, uint64_t n, SelectState* st, int digit) {
	__shared__ unsigned int h[256];
	for (int i = threadIdx.x; i < 256; i += kThreads) h[i] = 0;
	__syncthreads();
	const unsigned long long prefix = st->prefix;
	const int shift = digit * 8;
	const unsigned long long himask = digit == 7 ? 0ull : (~0ull << (shift + 8));
	for (uint64_t i = uint64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kThreads) {
		const unsigned long long key = composite_key(dist[i], i);
		if ((key & himask) == prefix) atomicAdd(&h[(key >> shift) & 0xFF], 1u);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < 256; i += kThreads) {
		if (h[i]) atomicAdd(&st->hist[digit][i], h[i]);
	}
}

__global__ void select_pick_digit(SelectState* st, int digit) {
	if (threadIdx.x != 0) return;
	unsigned long long rem = st->remaining;
	unsigned int b = 0;
	for (; b < 256; ++b) {
		const unsigned int c = st->hist[digit][b];
		if (rem <= c) break;
		rem -= c;
	}
	if (b > 255) b = 255;   // kk > n: everything is taken
	st->prefix |= static_cast<unsigned long long>(b) << (digit * 8);
	st->remaining = rem;
}

// After all 8 digits st->prefix is the kk-th smallest composite key: emit everything <= it.
__global__ __launch_bounds__(kThreads) void select_emit(const float* dist, uint64_t n, SelectState* st, uint32_t kk, float* out_dist,
														  uint32_t* out_row) {
	const unsigned long long thr = st->prefix;
	const int lane = threadIdx.x & 63;
	const uint64_t span = uint64_t(gridDim.x) * kThreads;
	const uint64_t iters = (n + span - 1) / span;
	for (uint64_t it = 0; it < iters; ++it) {
		const uint64_t i = it * span + uint64_t(blockIdx.x) * kThreads + threadIdx.x;
		float d = 0.f;
		bool hit = false;
		if (i < n) {
			d = dist[i];
			hit = composite_key(d, i) <= thr;
		}
		const uint64_t hm = __ballot(hit);
		if (hm) {
			unsigned long long basePos = 0;
			if (lane == 0) basePos = atomicAdd(&st->out_count, (unsigned long long)__popcll(hm));
			basePos = __shfl(basePos, 0);
			if (hit) {
				const uint64_t pos = basePos + __popcll(hm & ((1ull << lane) - 1));
				if (pos < kk) {
					out_dist[pos] = d;
					out_row[pos] = uint32_t(i);
				}
			}
		}
	}
}

size_t select_scratch_bytes(uint64_t) { return sizeof(SelectState); }

void launch_select_smallest(const float* d_dist, uint64_t n, uint32_t kk, void* d_scratch, float* d_out_dist, uint32_t* d_out_row,
							hipStream_t s) {
	auto* st = static_cast<SelectState*>(d_scratch);
	uint64_t blocks = (n + kThreads - 1) / kThreads;
	if (blocks > 2048) blocks = 2048;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL(select_init, dim3(1), dim3(256), 0, s, st, kk);
	for (int digit = 7; digit >= 0; --digit) {
		hipLaunchKernelGGL(select_histogram, dim3(uint32_t(blocks)), dim3(kThreads), 0, s, d_dist, n, st, digit);
		hipLaunchKernelGGL(select_pick_digit, dim3(1), dim3(64), 0, s, st, digit);
	}
	hipLaunchKernelGGL(select_emit, dim3(uint32_t(blocks)), dim3(kThreads), 0, s, d_dist, n, st, kk, d_out_dist, d_out_row);
}

}  // namespace rxgpu
