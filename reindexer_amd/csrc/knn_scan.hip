// Streaming brute-force scan + fused wavefront top-k, merge, range and re-score kernels (gfx950).
// See knn_kernels.hip.h for the arithmetic contract.  Reference being replaced:
// cpp_src/core/index/float_vector/hnswlib/bruteforce.cc:103-143.
#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

#include <cstdlib>

namespace rxgpu {

constexpr int kScanThreads = 256;                    // 4 wavefronts per workgroup, one per SIMD
constexpr int kScanWaves = kScanThreads / kWave;
constexpr int kMergeThreads = 512;
constexpr int kMergeWaves = kMergeThreads / kWave;

// Workgroup epilogue shared by both scan kernels: fold the per-wave lists into one and store it.
__device__ __forceinline__ void block_merge_and_store(WaveTopK& top, const ScanParams& p, int lane, int wave) {
	__shared__ float s_d[kScanWaves][kMaxFusedK];
	__shared__ uint32_t s_i[kScanWaves][kMaxFusedK];
	s_d[wave][lane] = top.bd;
	s_i[wave][lane] = top.bi;
	__syncthreads();
	if (wave != 0) return;
	for (int w = 1; w < kScanWaves; ++w) {
		const float cd = s_d[w][lane];
		const uint32_t ci = s_i[w][lane];
		// lists are sorted: once one entry is rejected the rest of that list is too
		uint64_t pm = __ballot(ci != kInvalidRow && lane < int(top.kk));
		while (pm) {
			const int src = __builtin_ctzll(pm);
			pm &= pm - 1;
			const float d = __shfl(cd, src);
			const uint32_t i = __shfl(ci, src);
			if (!top.admits(d, i)) break;
			top.insert(d, i, lane);
		}
	}
	if (lane < int(top.kk)) {
		const size_t o = (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * top.kk + lane;
		p.part_dist[o] = top.bd;
		p.part_row[o] = top.bi;
	}
}

// Rows of this wave's step -> candidate insertion.  `dist` is replicated over each 16-lane group.
__device__ __forceinline__ void consider_quad(WaveTopK& top, float dist, uint32_t row, bool valid, int lane) {
	// Rows arrive in increasing index order within a wave, so a tie with the current worst never enters:
	// strict `<` is exactly the reference's admission test (bruteforce.cc:121).
	const bool pass = valid && (top.filled < top.kk || dist < top.thr_d);
	uint64_t pm = __ballot(pass) & 0x0001000100010001ull;   // one representative lane per group
	while (pm) {
		const int src = __builtin_ctzll(pm);
		pm &= pm - 1;
		const float d = __shfl(dist, src);
		const uint32_t i = __shfl(row, src);
		if (top.admits(d, i)) top.insert(d, i, lane);
	}
}

// dim == 64*NB.  Per step a wavefront streams 4 rows (NB 16-byte loads per lane, all issued back to back).
//   kQLds     : query fragment read from LDS each step (frees NB*4 VGPRs) instead of living in registers
//   kPrefetch : the loads of step i+1 are in flight while step i is reduced (double-buffered registers)
//   kNT       : non-temporal row loads
template <int kMetric, int NB, bool kQLds, bool kPrefetch, bool kNT>
__global__ __launch_bounds__(kScanThreads) void knn_scan_fixed(ScanParams p) {
	__shared__ float4 s_q[kQLds ? NB * 16 : 1];
	if (p.gate_cnt && p.gate_cnt[blockIdx.y] <= p.gate_cap) return;
	const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const float4* qg = reinterpret_cast<const float4*>(p.queries + size_t(blockIdx.y) * p.dim);
	float4 q[kQLds ? 1 : NB];
	if constexpr (kQLds) {
		for (int i = threadIdx.x; i < NB * 16; i += kScanThreads) s_q[i] = qg[i];
		__syncthreads();
	} else {
#pragma unroll
		for (int t = 0; t < NB; ++t) q[t] = qg[16 * t + m];
	}

	WaveTopK top;
	top.init(p.kk);

	const uint64_t nquads = (p.n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	const uint64_t first = uint64_t(blockIdx.x) * kScanWaves + wave;

	auto issue = [&](float4 (&x)[NB], uint64_t quad) {
		const uint64_t qc = quad < nquads ? quad : nquads - 1;
		uint64_t row = qc * kRowsPerWave + g;
		row = row < p.n ? row : p.n - 1;
		const float4* rp = reinterpret_cast<const float4*>(p.rows + row * p.stride) + m;
#pragma unroll
		for (int t = 0; t < NB; ++t) x[t] = load_row4<kNT>(rp + 16 * t);
	};
	auto reduce = [&](const float4 (&x)[NB], uint64_t quad) {
		float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
		for (int t = 0; t < NB; ++t) {
			if constexpr (kQLds) {
				chain_step<kMetric>(acc, s_q[16 * t + m], x[t]);
			} else {
				chain_step<kMetric>(acc, q[t], x[t]);
			}
		}
		const uint64_t row = quad * kRowsPerWave + g;
		const bool valid = quad < nquads && row < p.n;
		const float sum = fold_chains<false>(acc, nullptr, nullptr, 0, m) + 0.0f;
		const float dist = metric_epilogue<kMetric>(sum, p.inv_norms, valid ? row : p.n - 1);
		consider_quad(top, dist, uint32_t(row), valid, lane);
	};

	if (first < nquads) {
		if constexpr (kPrefetch) {
			float4 xa[NB], xb[NB];
			issue(xa, first);
			for (uint64_t quad = first; quad < nquads; quad += 2 * nwaves) {
				issue(xb, quad + nwaves);
				__builtin_amdgcn_sched_barrier(0);
				reduce(xa, quad);
				__builtin_amdgcn_sched_barrier(0);
				issue(xa, quad + 2 * nwaves);
				__builtin_amdgcn_sched_barrier(0);
				reduce(xb, quad + nwaves);
				__builtin_amdgcn_sched_barrier(0);
			}
		} else {
			float4 x[NB];
			for (uint64_t quad = first; quad < nquads; quad += nwaves) {
				issue(x, quad);
				__builtin_amdgcn_sched_barrier(0);
				reduce(x, quad);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
	block_merge_and_store(top, p, lane, wave);
}

// Any dim (tails included); query read through the caches.
template <int kMetric>
__global__ __launch_bounds__(kScanThreads) void knn_scan_generic(ScanParams p) {
	if (p.gate_cnt && p.gate_cnt[blockIdx.y] <= p.gate_cap) return;
	const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const float* q = p.queries + size_t(blockIdx.y) * p.dim;
	WaveTopK top;
	top.init(p.kk);
	const uint64_t nquads = (p.n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	for (uint64_t quad = uint64_t(blockIdx.x) * kScanWaves + wave; quad < nquads; quad += nwaves) {
		const uint64_t row = quad * kRowsPerWave + g;
		const bool valid = row < p.n;
		const uint64_t rowc = valid ? row : p.n - 1;
		const float sum = group_distance_generic<kMetric>(p.rows + rowc * p.stride, q, p.dim, m);
		const float dist = metric_epilogue<kMetric>(sum, p.inv_norms, rowc);
		consider_quad(top, dist, uint32_t(row), valid, lane);
	}
	block_merge_and_store(top, p, lane, wave);
}

// One workgroup per query: fold `total` candidate (dist,row) pairs (invalid rows skipped) into the sorted top-kk.
__global__ __launch_bounds__(kMergeThreads) void knn_merge(const float* part_dist, const uint32_t* part_row, uint32_t total,
															uint32_t kk, float* out_dist, uint32_t* out_row, uint32_t* out_count,
															const uint32_t* gate_cnt, uint32_t gate_cap) {
	__shared__ float s_d[kMergeWaves][kMaxFusedK];
	__shared__ uint32_t s_i[kMergeWaves][kMaxFusedK];
	if (gate_cnt && gate_cnt[blockIdx.x] <= gate_cap) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t base = size_t(blockIdx.x) * total;
	WaveTopK top;
	top.init(kk);
	for (uint32_t c0 = wave * kWave; c0 < total; c0 += kMergeThreads) {
		const uint32_t c = c0 + lane;
		float cd = __builtin_inff();
		uint32_t ci = kInvalidRow;
		if (c < total) {
			cd = part_dist[base + c];
			ci = part_row[base + c];
		}
		uint64_t pm = __ballot(ci != kInvalidRow && top.admits(cd, ci));
		while (pm) {
			const int src = __builtin_ctzll(pm);
			pm &= pm - 1;
			const float d = __shfl(cd, src);
			const uint32_t i = __shfl(ci, src);
			if (top.admits(d, i)) top.insert(d, i, lane);
		}
	}
	s_d[wave][lane] = top.bd;
	s_i[wave][lane] = top.bi;
	__syncthreads();
	if (wave != 0) return;
	for (int w = 1; w < kMergeWaves; ++w) {
		const float cd = s_d[w][lane];
		const uint32_t ci = s_i[w][lane];
		uint64_t pm = __ballot(ci != kInvalidRow && lane < int(kk));
		while (pm) {
			const int src = __builtin_ctzll(pm);
			pm &= pm - 1;
			const float d = __shfl(cd, src);
			const uint32_t i = __shfl(ci, src);
			if (!top.admits(d, i)) break;
			top.insert(d, i, lane);
		}
	}
	if (lane < int(kk)) {
		out_dist[size_t(blockIdx.x) * kk + lane] = top.bd;
		out_row[size_t(blockIdx.x) * kk + lane] = top.bi;
	}
	if (lane == 0 && out_count) out_count[blockIdx.x] = top.filled;
}

// Multi-GPU: fold the all-gathered per-shard lists of one query batch into the global top-kk.
// gathered: [world][2][nq][kk] 32-bit words — per shard the [nq][kk] distances followed by the [nq][kk] shard-local rows (what
// the scan writes when d_out_row == d_out_dist + nq*kk).  Global row = shard * shard_rows + local row; order = (dist, global row).
__global__ __launch_bounds__(64) void knn_merge_shards(const uint32_t* gathered, uint32_t world, uint32_t nq, uint32_t kk, uint32_t shard_rows,
														float* out_dist, uint32_t* out_row, uint32_t* out_count) {
	const int lane = threadIdx.x;
	const uint32_t q = blockIdx.x;
	WaveTopK top;
	top.init(kk);
	for (uint32_t w = 0; w < world; ++w) {
		const uint32_t* rec = gathered + size_t(w) * 2 * nq * kk + size_t(q) * kk;
		float cd = __builtin_inff();
		uint32_t ci = kInvalidRow;
		if (lane < int(kk)) {
			cd = __uint_as_float(rec[lane]);
			const uint32_t local = rec[size_t(nq) * kk + lane];
			if (local != kInvalidRow) ci = w * shard_rows + local;
		}
		uint64_t pm = __ballot(ci != kInvalidRow);
		while (pm) {
			const int src = __builtin_ctzll(pm);
			pm &= pm - 1;
			const float d = __shfl(cd, src);
			const uint32_t i = __shfl(ci, src);
			if (!top.admits(d, i)) break;   // each shard list is sorted
			top.insert(d, i, lane);
		}
	}
	if (lane < int(kk)) {
		out_dist[size_t(q) * kk + lane] = top.bd;
		out_row[size_t(q) * kk + lane] = top.bi;
	}
	if (lane == 0 && out_count) out_count[q] = top.filled;
}

// BruteforceSearch::SearchRange (bruteforce.cc:129-143): compact every row with dist < radius (or <=).
struct RangeParams {
	const float* rows;
	const float* inv_norms;
	const float* query;
	uint64_t n;
	uint32_t stride, dim;
	float radius;
	int inclusive;
	float* out_dist;
	uint32_t* out_row;
	uint64_t cap;
	unsigned long long* counter;
};

template <int kMetric>
__global__ __launch_bounds__(kScanThreads) void knn_range(RangeParams p) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
	const uint64_t nquads = (p.n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	for (uint64_t quad = uint64_t(blockIdx.x) * kScanWaves + wave; quad < nquads; quad += nwaves) {
		const uint64_t row = quad * kRowsPerWave + g;
		const bool valid = row < p.n;
		const uint64_t rowc = valid ? row : p.n - 1;
		const float sum = group_distance_generic<kMetric>(p.rows + rowc * p.stride, p.query, p.dim, m);
		const float dist = metric_epilogue<kMetric>(sum, p.inv_norms, rowc);
		const bool hit = valid && m == 0 && (p.inclusive ? dist <= p.radius : dist < p.radius);
		const uint64_t hm = __ballot(hit);
		if (hm) {
			unsigned long long basePos = 0;
			if (lane == 0) basePos = atomicAdd(p.counter, (unsigned long long)__popcll(hm));
			basePos = __shfl(basePos, 0);
			if (hit) {
				const uint64_t pos = basePos + __popcll(hm & ((1ull << lane) - 1));
				if (pos < p.cap) {
					p.out_dist[pos] = dist;
					p.out_row[pos] = uint32_t(row);
				}
			}
		}
	}
}

// DistCalculator::operator()(q,row,id) for an explicit row list: one 16-lane group per row.
template <int kMetric>
__global__ __launch_bounds__(256) void knn_distances(const float* rows, const float* inv_norms, const float* query, uint32_t stride,
													uint32_t dim, const uint32_t* ids, uint32_t n, float* out) {
	const int lane = threadIdx.x & 63, m = lane & 15;
	const uint32_t item = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	const uint32_t itemc = item < n ? item : n - 1;
	const uint64_t row = ids[itemc];
	const float sum = group_distance_generic<kMetric>(rows + row * stride, query, dim, m);
	const float dist = metric_epilogue<kMetric>(sum, inv_norms, row);
	if (item < n && m == 0) out[item] = dist;
}

// ------------------------------------------------------------------------------------------ launchers

// Tuning knobs (read once): RXGPU_SCAN_QLDS, RXGPU_SCAN_PREFETCH, RXGPU_SCAN_NT, RXGPU_SCAN_WG_PER_CU.
struct ScanTuning {
	int qlds = 1, prefetch = 1, nt = 1, wg_per_cu = 2;   // measured best on MI355X (profiles/r1_tune_ip768.jsonl)
	ScanTuning() {
		if (const char* e = getenv("RXGPU_SCAN_QLDS")) qlds = atoi(e);
		if (const char* e = getenv("RXGPU_SCAN_PREFETCH")) prefetch = atoi(e);
		if (const char* e = getenv("RXGPU_SCAN_NT")) nt = atoi(e);
		if (const char* e = getenv("RXGPU_SCAN_WG_PER_CU")) wg_per_cu = atoi(e);
		if (wg_per_cu < 1) wg_per_cu = 1;
	}
};
static const ScanTuning& tuning() {
	static ScanTuning t;
	static const bool dynamic = getenv("RXGPU_TUNE_DYNAMIC") != nullptr;   // tools/tune_scan.py: re-read on every launch
	if (dynamic) t = ScanTuning();
	return t;
}

template <int kMetric, int NB>
static void launch_scan_fixed(const ScanParams& p, dim3 grid, hipStream_t s) {
	const ScanTuning& t = tuning();
	const int v = (t.qlds ? 4 : 0) | (t.prefetch ? 2 : 0) | (t.nt ? 1 : 0);
	const dim3 blk(kScanThreads);
	switch (v) {
		case 0: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, false, false, false>), grid, blk, 0, s, p); break;
		case 1: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, false, false, true>), grid, blk, 0, s, p); break;
		case 2: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, false, true, false>), grid, blk, 0, s, p); break;
		case 3: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, false, true, true>), grid, blk, 0, s, p); break;
		case 4: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, true, false, false>), grid, blk, 0, s, p); break;
		case 5: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, true, false, true>), grid, blk, 0, s, p); break;
		case 6: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, true, true, false>), grid, blk, 0, s, p); break;
		default: hipLaunchKernelGGL((knn_scan_fixed<kMetric, NB, true, true, true>), grid, blk, 0, s, p); break;
	}
}

template <int kMetric>
static void launch_scan_metric(const ScanParams& p, dim3 grid, hipStream_t s) {
	switch (p.dim) {
		case 64: launch_scan_fixed<kMetric, 1>(p, grid, s); return;
		case 128: launch_scan_fixed<kMetric, 2>(p, grid, s); return;
		case 256: launch_scan_fixed<kMetric, 4>(p, grid, s); return;
		case 384: launch_scan_fixed<kMetric, 6>(p, grid, s); return;
		case 512: launch_scan_fixed<kMetric, 8>(p, grid, s); return;
		case 768: launch_scan_fixed<kMetric, 12>(p, grid, s); return;
		case 1024: launch_scan_fixed<kMetric, 16>(p, grid, s); return;
		default: hipLaunchKernelGGL((knn_scan_generic<kMetric>), grid, dim3(kScanThreads), 0, s, p);
	}
}

uint32_t scan_grid_x(uint64_t n, int cus) {
	const uint64_t nquads = (n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t want = (nquads + kScanWaves - 1) / kScanWaves;
	const uint64_t cap = uint64_t(cus) * tuning().wg_per_cu;
	return uint32_t(want < cap ? (want ? want : 1) : cap);
}

void launch_scan(int metric, const ScanParams& p, uint32_t nq, uint32_t gridx, hipStream_t s) {
	const dim3 grid(gridx, nq);
	switch (metric) {
		case kL2: launch_scan_metric<kL2>(p, grid, s); break;
		case kIP: launch_scan_metric<kIP>(p, grid, s); break;
		default: launch_scan_metric<kCos>(p, grid, s); break;
	}
}

void launch_merge(const float* part_dist, const uint32_t* part_row, uint32_t total_per_query, uint32_t kk, uint32_t nq, float* out_dist,
				  uint32_t* out_row, uint32_t* out_count, const uint32_t* gate_cnt, uint32_t gate_cap, hipStream_t s) {
	hipLaunchKernelGGL(knn_merge, dim3(nq), dim3(kMergeThreads), 0, s, part_dist, part_row, total_per_query, kk, out_dist, out_row, out_count,
					   gate_cnt, gate_cap);
}

void launch_merge_shards(const uint32_t* gathered, uint32_t world, uint32_t nq, uint32_t kk, uint32_t shard_rows, float* out_dist,
						 uint32_t* out_row, uint32_t* out_count, hipStream_t s) {
	hipLaunchKernelGGL(knn_merge_shards, dim3(nq), dim3(64), 0, s, gathered, world, nq, kk, shard_rows, out_dist, out_row, out_count);
}

void launch_range(int metric, const float* rows, const float* inv_norms, const float* query, uint64_t n, uint32_t stride, uint32_t dim,
				  float radius, int inclusive, float* out_dist, uint32_t* out_row, uint64_t cap, unsigned long long* counter,
				  uint32_t gridx, hipStream_t s) {
	RangeParams p{rows, inv_norms, query, n, stride, dim, radius, inclusive, out_dist, out_row, cap, counter};
	switch (metric) {
		case kL2: hipLaunchKernelGGL((knn_range<kL2>), dim3(gridx), dim3(kScanThreads), 0, s, p); break;
		case kIP: hipLaunchKernelGGL((knn_range<kIP>), dim3(gridx), dim3(kScanThreads), 0, s, p); break;
		default: hipLaunchKernelGGL((knn_range<kCos>), dim3(gridx), dim3(kScanThreads), 0, s, p); break;
	}
}

void launch_distances(int metric, const float* rows, const float* inv_norms, const float* query, uint32_t stride, uint32_t dim,
					  const uint32_t* ids, uint32_t n, float* out, hipStream_t s) {
	const uint32_t blocks = (n * kGroup + 255) / 256;
	switch (metric) {
		case kL2: hipLaunchKernelGGL((knn_distances<kL2>), dim3(blocks), dim3(256), 0, s, rows, inv_norms, query, stride, dim, ids, n, out); break;
		case kIP: hipLaunchKernelGGL((knn_distances<kIP>), dim3(blocks), dim3(256), 0, s, rows, inv_norms, query, stride, dim, ids, n, out); break;
		default: hipLaunchKernelGGL((knn_distances<kCos>), dim3(blocks), dim3(256), 0, s, rows, inv_norms, query, stride, dim, ids, n, out); break;
	}
}

}  // namespace rxgpu
