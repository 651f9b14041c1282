// Streaming brute-force scan + fused wavefront top-k, merge, range and re-score kernels (gfx950).
// See knn_kernels.hip.h for the arithmetic contract.  Reference being replaced:
// cpp_src/core/index/float_vector/hnswlib/bruteforce.cc:103-143.
#include <algorithm>
#include <type_traits>

#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

#include <cstdlib>

namespace rxgpu {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kScanThreads = 256;                    // 4 wavefronts per workgroup, one per SIMD
constexpr int kScanWaves = kScanThreads / kWave;
constexpr int kMergeThreads = 512;
constexpr int kMergeWaves = kMergeThreads / kWave;
constexpr int kMergeAhead = 8;                       // candidate chunks whose loads a merge wavefront keeps in flight

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

// the same for the two-entries-per-lane list (64 < kk <= 128)
__device__ __forceinline__ void block_merge_and_store(WaveTopK2& top, const ScanParams& p, int lane, int wave) {
	__shared__ float s_d[kScanWaves][kMaxFusedK2];
	__shared__ uint32_t s_i[kScanWaves][kMaxFusedK2];
	s_d[wave][lane] = top.d0;
	s_i[wave][lane] = top.i0;
	s_d[wave][64 + lane] = top.d1;
	s_i[wave][64 + lane] = top.i1;
	__syncthreads();
	if (wave != 0) return;
	for (int w = 1; w < kScanWaves; ++w) {
		for (uint32_t e = 0; e < top.kk; ++e) {   // sorted: once one entry is rejected the rest of that list is too
			const float d = s_d[w][e];
			const uint32_t i = s_i[w][e];
			if (i == kInvalidRow || !top.admits(d, i)) break;
			top.insert(d, i, lane);
		}
	}
	const size_t o = (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * top.kk;
	if (lane < int(top.kk)) {
		p.part_dist[o + lane] = top.d0;
		p.part_row[o + lane] = top.i0;
	}
	if (64 + lane < int(top.kk)) {
		p.part_dist[o + 64 + lane] = top.d1;
		p.part_row[o + 64 + lane] = top.i1;
	}
}

// Rows of this wave's step -> candidate insertion.  `dist` is replicated over each 16-lane group.
template <typename TK>
__device__ __forceinline__ void consider_quad(TK& top, float dist, uint32_t row, bool valid, int lane) {
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
template <int kMetric, int NB, bool kQLds, bool kPrefetch, bool kNT, typename TK = WaveTopK>
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

	TK top;
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
template <int kMetric, typename TK = WaveTopK>
__global__ __launch_bounds__(kScanThreads) void knn_scan_generic(ScanParams p) {
	if (p.gate_cnt && p.gate_cnt[blockIdx.y] <= p.gate_cap) return;
	const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const float* q = p.queries + size_t(blockIdx.y) * p.dim;
	TK top;
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Pre-filtered scan: only the rows listed in `ids` (strictly increasing internal row numbers, p.n of them) compete.  This is the
// caller side of `WHERE cond AND KNN(...)` (the reference post-filters on the host, nsselecter.cc:841-875): HBM traffic is
// ids (4 B) + one row per ALLOWED row, so a 5 % filter reads 5 % of the corpus.  Distances are the same per-row arithmetic as
// knn_scan_fixed and the (dist,row) list order is unchanged, hence the result is the unfiltered engine's result over the sub-corpus.
//
// A wavefront owns chunks of 64 consecutive list entries (chunk c of wave w = first + c * nwaves): ONE coalesced 256-byte load brings the
// ids of 16 steps (lane l: step l / 4, group l % 4; a step's id is fetched with a cross-lane read), so the dependent id -> row address
// hop is off the per-step path and the row loads keep the double-buffered issue order of knn_scan_fixed across chunk boundaries.
template <int kMetric, int NB, typename TK = WaveTopK>
__global__ __launch_bounds__(kScanThreads) void knn_scan_subset(ScanParams p, const uint32_t* __restrict__ ids) {
	__shared__ float4 s_q[NB * 16];
	const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const float4* qg = reinterpret_cast<const float4*>(p.queries + size_t(blockIdx.y) * p.dim);
	for (int i = threadIdx.x; i < NB * 16; i += kScanThreads) s_q[i] = qg[i];
	__syncthreads();

	TK top;
	top.init(p.kk);

	constexpr int kChunk = 64, kSteps = kChunk / kRowsPerWave;
	const uint64_t nchunks = (p.n + kChunk - 1) / kChunk;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	const uint64_t first = uint64_t(blockIdx.x) * kScanWaves + wave;

	auto load_ids = [&](uint64_t chunk) -> uint32_t {   // clamped: reading past the list re-reads its last entry
		const uint64_t cc = chunk < nchunks ? chunk : nchunks - 1;
		const uint64_t item = cc * kChunk + lane;
		return ids[item < p.n ? item : p.n - 1];
	};
	auto issue = [&](float4 (&x)[NB], uint32_t row) {
		const float4* rp = reinterpret_cast<const float4*>(p.rows + uint64_t(row) * p.stride) + m;
#pragma unroll
		for (int t = 0; t < NB; ++t) x[t] = load_row4<true>(rp + 16 * t);
	};
	auto reduce = [&](const float4 (&x)[NB], uint64_t item, uint32_t row) {
		float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
		for (int t = 0; t < NB; ++t) chain_step<kMetric>(acc, s_q[16 * t + m], x[t]);
		const float sum = fold_chains<false>(acc, nullptr, nullptr, 0, m) + 0.0f;
		const float dist = metric_epilogue<kMetric>(sum, p.inv_norms, row);
		consider_quad(top, dist, row, item < p.n, lane);
	};

	if (first < nchunks) {
		float4 xa[NB], xb[NB];
		uint32_t idc = load_ids(first);
		issue(xa, __shfl(idc, g));
		for (uint64_t chunk = first; chunk < nchunks; chunk += nwaves) {
			const uint32_t idn = load_ids(chunk + nwaves);   // in flight while this chunk's 16 steps run
			const uint64_t base = chunk * kChunk + g;
#pragma unroll 1
			for (int t = 0; t < kSteps; t += 2) {
				const uint32_t ra = __shfl(idc, t * kRowsPerWave + g);
				const uint32_t rb = __shfl(idc, (t + 1) * kRowsPerWave + g);
				issue(xb, rb);
				__builtin_amdgcn_sched_barrier(0);
				reduce(xa, base + uint64_t(t) * kRowsPerWave, ra);
				__builtin_amdgcn_sched_barrier(0);
				// step t + 2 of this chunk, or step 0 of this wave's next chunk (harmless re-read after the last one)
				const uint32_t rn = t + 2 < kSteps ? __shfl(idc, (t + 2) * kRowsPerWave + g) : __shfl(idn, g);
				issue(xa, rn);
				__builtin_amdgcn_sched_barrier(0);
				reduce(xb, base + uint64_t(t + 1) * kRowsPerWave, rb);
				__builtin_amdgcn_sched_barrier(0);
			}
			idc = idn;
		}
	}
	block_merge_and_store(top, p, lane, wave);
}

// Any dim: one list entry per 16-lane group and step.
template <int kMetric, typename TK = WaveTopK>
__global__ __launch_bounds__(kScanThreads) void knn_scan_subset_generic(ScanParams p, const uint32_t* __restrict__ ids) {
	const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const float* q = p.queries + size_t(blockIdx.y) * p.dim;
	TK top;
	top.init(p.kk);
	const uint64_t nquads = (p.n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	for (uint64_t quad = uint64_t(blockIdx.x) * kScanWaves + wave; quad < nquads; quad += nwaves) {
		const uint64_t item = quad * kRowsPerWave + g;
		const bool valid = item < p.n;
		const uint32_t row = ids[valid ? item : p.n - 1];
		const float sum = group_distance_generic<kMetric>(p.rows + uint64_t(row) * p.stride, q, p.dim, m);
		const float dist = metric_epilogue<kMetric>(sum, p.inv_norms, row);
		consider_quad(top, dist, row, valid, lane);
	}
	block_merge_and_store(top, p, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// bf16-pruned scan (opt-in, RXGPU_SCAN_BF16=1): half the HBM bytes per query, the SAME result bits.
//   1. knn_scan_bf16      approximate distance d~ of every row from the bf16 shadow (2 bytes per element), stored ([n] floats) and folded
//                         into the per-wave top-kk exactly like the f32 scan
//   2. knn_merge          -> d~_(kk), the kk-th best approximate distance
//   3. knn_filter_approx  rows with d~ <= d~_(kk) + 2 eps  (eps = the rigorous bf16 bound of knn_query_stats<.., true>)
//   4. knn_rescore + knn_merge   EXACT distances of those few dozen rows, exact top-kk by (dist, row)
// Soundness: a row r of the true top-kk has d~_r <= d_r + eps <= D_kk + eps, and D_kk <= kk-th smallest of (d~ + eps) = d~_(kk) + eps.
// One 16-lane group owns a row: lane m loads the 16-byte chunks m, m + 16, ... (8 bf16 each; 256 contiguous bytes per group per load),
// widens them to f32 (exact) and runs four fmaf chains against the f32 query fragment it keeps in registers.
template <int NC>
__device__ __forceinline__ float bf16_group_dot(const u32x4 (&x)[NC], const float (&q)[NC * 8]) {
	float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
	for (int t = 0; t < NC; ++t) {
		a0 = __builtin_fmaf(q[8 * t + 0], __uint_as_float(x[t].x << 16), a0);
		a1 = __builtin_fmaf(q[8 * t + 1], __uint_as_float(x[t].x & 0xFFFF0000u), a1);
		a2 = __builtin_fmaf(q[8 * t + 2], __uint_as_float(x[t].y << 16), a2);
		a3 = __builtin_fmaf(q[8 * t + 3], __uint_as_float(x[t].y & 0xFFFF0000u), a3);
		a0 = __builtin_fmaf(q[8 * t + 4], __uint_as_float(x[t].z << 16), a0);
		a1 = __builtin_fmaf(q[8 * t + 5], __uint_as_float(x[t].z & 0xFFFF0000u), a1);
		a2 = __builtin_fmaf(q[8 * t + 6], __uint_as_float(x[t].w << 16), a2);
		a3 = __builtin_fmaf(q[8 * t + 7], __uint_as_float(x[t].w & 0xFFFF0000u), a3);
	}
	float s = (a0 + a2) + (a1 + a3);
	s += __shfl_xor(s, 8);
	s += __shfl_xor(s, 4);
	s += __shfl_xor(s, 2);
	s += __shfl_xor(s, 1);
	return s;
}

// NC = ld / 128: chunks per lane per row (768 -> 6)
template <int kMetric, int NC>
__global__ __launch_bounds__(kScanThreads) void knn_scan_bf16(ScanBf16Params p) {
	const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t qi = blockIdx.y;
	float q[NC * 8];
	{
		const float4* qg = reinterpret_cast<const float4*>(p.queries32 + size_t(qi) * p.ld);
#pragma unroll
		for (int t = 0; t < NC; ++t) {
			const float4 lo = qg[2 * (m + 16 * t)], hi = qg[2 * (m + 16 * t) + 1];
			q[8 * t + 0] = lo.x; q[8 * t + 1] = lo.y; q[8 * t + 2] = lo.z; q[8 * t + 3] = lo.w;
			q[8 * t + 4] = hi.x; q[8 * t + 5] = hi.y; q[8 * t + 6] = hi.z; q[8 * t + 7] = hi.w;
		}
	}
	float q_term = 0.f;
	if constexpr (kMetric == kL2) q_term = p.q_sq[qi];
	WaveTopK top;
	top.init(p.sp.kk);
	float* approx = p.approx + size_t(qi) * p.sp.n;
	const uint64_t n = p.sp.n;
	const uint64_t nquads = (n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	const uint64_t first = uint64_t(blockIdx.x) * kScanWaves + wave;
	auto issue = [&](u32x4 (&x)[NC], uint64_t quad) {
		uint64_t row = quad * kRowsPerWave + g;
		if (row >= n) row = n - 1;
		// chunk m + 16 t of the row = k-block (m >> 2) + 4 t, 16-byte piece m & 3 of it; in the tile-blocked shadow the four 16-lane groups
		// of a wavefront (four consecutive rows) read 256 contiguous bytes per k-block, as they read 256 contiguous bytes per row otherwise
		const bool blocked = p.blocked != 0;
		const u32x4* src = reinterpret_cast<const u32x4*>(p.rows16 + shadow_elem_base(row, p.ld, blocked) + uint64_t(m >> 2) * shadow_stage_step(blocked) + (m & 3) * 8);
		const uint32_t tstep = shadow_stage_step(blocked) / 2;   // four k-blocks further on, in 16-byte units
#pragma unroll
		for (int t = 0; t < NC; ++t) x[t] = __builtin_nontemporal_load(src + size_t(tstep) * t);
	};
	auto reduce = [&](const u32x4 (&x)[NC], uint64_t quad) {
		if (quad >= nquads) return;
		const uint64_t row = quad * kRowsPerWave + g;
		const bool valid = row < n;
		const uint64_t rowc = valid ? row : n - 1;
		const float sum = bf16_group_dot<NC>(x, q);
		float dist;
		if constexpr (kMetric == kL2) {
			dist = (q_term + p.row_sq[rowc]) - 2.0f * sum;
		} else if constexpr (kMetric == kIP) {
			dist = -sum;
		} else {
			dist = -sum * p.sp.inv_norms[rowc];
		}
		if (valid && m == 0) approx[row] = dist;
		consider_quad(top, dist, uint32_t(row), valid, lane);
	};
	if (first < nquads) {
		u32x4 xa[NC], xb[NC];
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
	}
	block_merge_and_store(top, p.sp, lane, wave);
}

// The same pass over the TILE-BLOCKED shadow ([tile of 256 rows][32-element k-block][row][32], knn_kernels.hip.h).  There a row's k-blocks lie
// 16 KB apart, and what is contiguous is a k-block of CONSECUTIVE rows: a wavefront therefore takes 16 rows at a time — lane 4 r + c owns
// 16-byte piece c of every k-block of row r — so that one load instruction reads 16 rows x 64 B = 1 KB in one piece (the 16-lane-per-row
// mapping above reads 256 B pieces from this layout: 2.54 ms against 2.46 on the row-major shadow at 10M x 768).  The query fragment a lane
// needs changes with the k-block: it comes from LDS (four distinct 32-byte addresses per wavefront and k-block: broadcast reads).  The
// k-blocks of a row travel in four groups of NC (the register budget of the double-buffered loads), partial sums carried between them.
template <int kMetric, int NC>
__global__ __launch_bounds__(kScanThreads) void knn_scan_bf16_blk(ScanBf16Params p) {
	__shared__ __attribute__((aligned(16))) float s_q[NC * 128];   // [k-block][piece][8]
	const int lane = threadIdx.x & 63, c = lane & 3, r = lane >> 2;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t qi = blockIdx.y;
	for (int i = threadIdx.x; i < NC * 128 / 4; i += kScanThreads) {
		reinterpret_cast<float4*>(s_q)[i] = reinterpret_cast<const float4*>(p.queries32 + size_t(qi) * p.ld)[i];   // ld = NC * 128 floats, k-major already
	}
	__syncthreads();
	float q_term = 0.f;
	if constexpr (kMetric == kL2) q_term = p.q_sq[qi];
	WaveTopK top;
	top.init(p.sp.kk);
	float* approx = p.approx + size_t(qi) * p.sp.n;
	const uint64_t n = p.sp.n;
	constexpr uint64_t kRows = 16;
	const uint64_t nsets = (n + kRows - 1) / kRows;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	const uint64_t first = uint64_t(blockIdx.x) * kScanWaves + wave;
	// unit u = (row set, k-block group): set = first + (u / 4) * nwaves, group = u % 4
	const uint64_t my_sets = first < nsets ? (nsets - first + nwaves - 1) / nwaves : 0;
	const uint64_t units = my_sets * 4;
	auto issue = [&](u32x4 (&x)[NC], uint64_t u) {
		if (u >= units) return;
		const uint64_t set = first + (u >> 2) * nwaves;
		const uint32_t grp = uint32_t(u & 3);
		uint64_t row = set * kRows + r;
		if (row >= n) row = n - 1;
		const u32x4* src = reinterpret_cast<const u32x4*>(p.rows16 + shadow_elem_base(row, p.ld, true) + uint64_t(grp) * NC * kShadowStageElems + c * 8);
#pragma unroll
		for (int t = 0; t < NC; ++t) x[t] = __builtin_nontemporal_load(src + size_t(t) * (kShadowStageElems / 8));
	};
	float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
	auto reduce = [&](const u32x4 (&x)[NC], uint64_t u) {
		if (u >= units) return;
		const uint32_t grp = uint32_t(u & 3);
#pragma unroll
		for (int t = 0; t < NC; ++t) {
			const float4 qa = *reinterpret_cast<const float4*>(&s_q[((grp * NC + t) * 4 + c) * 8]);
			const float4 qb = *reinterpret_cast<const float4*>(&s_q[((grp * NC + t) * 4 + c) * 8 + 4]);
			a0 = __builtin_fmaf(qa.x, __uint_as_float(x[t].x << 16), a0);
			a1 = __builtin_fmaf(qa.y, __uint_as_float(x[t].x & 0xFFFF0000u), a1);
			a2 = __builtin_fmaf(qa.z, __uint_as_float(x[t].y << 16), a2);
			a3 = __builtin_fmaf(qa.w, __uint_as_float(x[t].y & 0xFFFF0000u), a3);
			a0 = __builtin_fmaf(qb.x, __uint_as_float(x[t].z << 16), a0);
			a1 = __builtin_fmaf(qb.y, __uint_as_float(x[t].z & 0xFFFF0000u), a1);
			a2 = __builtin_fmaf(qb.z, __uint_as_float(x[t].w << 16), a2);
			a3 = __builtin_fmaf(qb.w, __uint_as_float(x[t].w & 0xFFFF0000u), a3);
			if (t & 1) __builtin_amdgcn_sched_barrier(0);   // two k-blocks of query fragment in registers at a time, not all NC (142 VGPRs: 3 waves per SIMD)
		}
		if (grp != 3) return;   // uniform
		float sum = (a0 + a2) + (a1 + a3);
		a0 = a1 = a2 = a3 = 0.f;
		sum += __shfl_xor(sum, 2);
		sum += __shfl_xor(sum, 1);
		const uint64_t set = first + (u >> 2) * nwaves;
		const uint64_t row = set * kRows + r;
		const bool valid = row < n;
		const uint64_t rowc = valid ? row : n - 1;
		float dist;
		if constexpr (kMetric == kL2) {
			dist = (q_term + p.row_sq[rowc]) - 2.0f * sum;
		} else if constexpr (kMetric == kIP) {
			dist = -sum;
		} else {
			dist = -sum * p.sp.inv_norms[rowc];
		}
		if (valid && c == 0) approx[row] = dist;
		// rows in increasing order within the wavefront: lane group r holds row set * 16 + r (consider_quad's rule, one lane per 4-lane group)
		const bool pass = valid && (top.filled < top.kk || dist < top.thr_d);
		uint64_t pm = __ballot(pass) & 0x1111111111111111ull;
		while (pm) {
			const int src = __builtin_ctzll(pm);
			pm &= pm - 1;
			const float d = __shfl(dist, src);
			const uint32_t i = __shfl(uint32_t(row), src);
			if (top.admits(d, i)) top.insert(d, i, lane);
		}
	};
	if (units) {
		u32x4 xa[NC], xb[NC];
		issue(xa, 0);
		for (uint64_t u = 0; u < units; u += 2) {
			issue(xb, u + 1);
			__builtin_amdgcn_sched_barrier(0);
			reduce(xa, u);
			__builtin_amdgcn_sched_barrier(0);
			issue(xa, u + 2);
			__builtin_amdgcn_sched_barrier(0);
			reduce(xb, u + 1);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	block_merge_and_store(top, p.sp, lane, wave);
}

// rows whose approximate distance is within the bound of the kk-th best approximate distance -> candidate list of the query
__global__ __launch_bounds__(256) void knn_filter_approx(const float* approx, uint64_t n, const float* top_dist, const uint32_t* top_count, uint32_t kk,
														 const float* margin, uint32_t* cand_row, uint32_t* cand_cnt, uint32_t cap) {
	const uint32_t qi = blockIdx.y;
	const int lane = threadIdx.x & 63;
	const float thr = (top_count[qi] >= kk ? top_dist[size_t(qi) * kk + kk - 1] : __builtin_inff()) + margin[qi];
	const float* a = approx + size_t(qi) * n;
	const uint64_t span = uint64_t(gridDim.x) * blockDim.x;
	for (uint64_t base = uint64_t(blockIdx.x) * blockDim.x; base < n; base += span) {   // wave-uniform trip count
		const uint64_t row = base + threadIdx.x;
		const bool pass = row < n && a[row] <= thr;
		const uint64_t pm = __ballot(pass);
		if (!pm) continue;
		uint32_t pos0 = 0;
		if (lane == __builtin_ctzll(pm)) pos0 = atomicAdd(&cand_cnt[qi], uint32_t(__popcll(pm)));
		pos0 = __shfl(pos0, __builtin_ctzll(pm));
		const uint32_t pos = pos0 + uint32_t(__popcll(pm & ((1ull << lane) - 1)));
		if (pass && pos < cap) cand_row[size_t(qi) * cap + pos] = uint32_t(row);
	}
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
	// kMergeAhead chunks of 64 candidates per trip, all their loads issued together: one chunk per trip left every trip behind its own
	// memory round trip (~2 us each: 0.27 ms for the 512 x 101 partial entries of a k = 100 search, against 0.03 ms of list work)
	for (uint32_t c0 = wave * kWave; c0 < total; c0 += kMergeThreads * kMergeAhead) {
		float cd[kMergeAhead];
		uint32_t ci[kMergeAhead];
#pragma unroll
		for (int u = 0; u < kMergeAhead; ++u) {
			const uint32_t c = c0 + uint32_t(u) * kMergeThreads + lane;
			cd[u] = __builtin_inff();
			ci[u] = kInvalidRow;
			if (c < total) {
				cd[u] = part_dist[base + c];
				ci[u] = part_row[base + c];
			}
		}
#pragma unroll
		for (int u = 0; u < kMergeAhead; ++u) {
			uint64_t pm = __ballot(ci[u] != kInvalidRow && top.admits(cd[u], ci[u]));
			while (pm) {
				const int src = __builtin_ctzll(pm);
				pm &= pm - 1;
				const float d = __shfl(cd[u], src);
				const uint32_t i = __shfl(ci[u], src);
				if (top.admits(d, i)) top.insert(d, i, lane);
			}
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

// knn_merge for 64 < kk <= 128 (two entries per lane)
__global__ __launch_bounds__(kMergeThreads) void knn_merge_wide(const float* part_dist, const uint32_t* part_row, uint32_t total, uint32_t kk,
																 float* out_dist, uint32_t* out_row, uint32_t* out_count) {
	__shared__ float s_d[kMergeWaves][kMaxFusedK2];
	__shared__ uint32_t s_i[kMergeWaves][kMaxFusedK2];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t base = size_t(blockIdx.x) * total;
	WaveTopK2 top;
	top.init(kk);
	// kMergeAhead chunks of 64 candidates per trip, all their loads issued together: one chunk per trip left every trip behind its own
	// memory round trip (~2 us each: 0.27 ms for the 512 x 101 partial entries of a k = 100 search, against 0.03 ms of list work)
	for (uint32_t c0 = wave * kWave; c0 < total; c0 += kMergeThreads * kMergeAhead) {
		float cd[kMergeAhead];
		uint32_t ci[kMergeAhead];
#pragma unroll
		for (int u = 0; u < kMergeAhead; ++u) {
			const uint32_t c = c0 + uint32_t(u) * kMergeThreads + lane;
			cd[u] = __builtin_inff();
			ci[u] = kInvalidRow;
			if (c < total) {
				cd[u] = part_dist[base + c];
				ci[u] = part_row[base + c];
			}
		}
#pragma unroll
		for (int u = 0; u < kMergeAhead; ++u) {
			uint64_t pm = __ballot(ci[u] != kInvalidRow && top.admits(cd[u], ci[u]));
			while (pm) {
				const int src = __builtin_ctzll(pm);
				pm &= pm - 1;
				const float d = __shfl(cd[u], src);
				const uint32_t i = __shfl(ci[u], src);
				if (top.admits(d, i)) top.insert(d, i, lane);
			}
		}
	}
	s_d[wave][lane] = top.d0;
	s_i[wave][lane] = top.i0;
	s_d[wave][64 + lane] = top.d1;
	s_i[wave][64 + lane] = top.i1;
	__syncthreads();
	if (wave != 0) return;
	for (int w = 1; w < kMergeWaves; ++w) {
		for (uint32_t e = 0; e < kk; ++e) {
			const float d = s_d[w][e];
			const uint32_t i = s_i[w][e];
			if (i == kInvalidRow || !top.admits(d, i)) break;
			top.insert(d, i, lane);
		}
	}
	const size_t o = size_t(blockIdx.x) * kk;
	if (lane < int(kk)) {
		out_dist[o + lane] = top.d0;
		out_row[o + lane] = top.i0;
	}
	if (64 + lane < int(kk)) {
		out_dist[o + 64 + lane] = top.d1;
		out_row[o + 64 + lane] = top.i1;
	}
	if (lane == 0 && out_count) out_count[blockIdx.x] = top.filled;
}

// ---- merge of SORTED partial lists (what the scans leave: gridDim.x lists of kk entries per query), without serial insertions.
// Folding ~500 lists through a wavefront-wide sorted list costs an insertion (a chain of cross-lane operations, ~0.4 us) for every
// candidate that beats the running kk-th — ~650 of them at k = 100: 0.25 ms behind a 1.5 ms scan.  The lists are sorted, so:
//   1. the kk-th smallest list HEAD bounds the kk-th best overall (the heads are kk real candidates at or below it);
//   2. only entries at or below that bound can be in the result: every list is walked from its head while it stays below — for rows
//      spread evenly over the partitions that is ~1.1 kk entries in total;
//   3. those candidates are sorted in LDS (bitonic, (distance, row) keys) and the first kk written.
// Exact for any data; when the bound lets more than kMergeCandMax entries through (masses of equal distances) or there are fewer lists
// than kk, the insertion merge runs instead (same kernel, workgroup-uniform branch).
constexpr uint32_t kMergeHeadsMax = 4096, kMergeCandMax = 2048;
__device__ __forceinline__ unsigned long long merge_pair_key(float d, uint32_t row) {
	uint32_t b = __float_as_uint(d);
	if (b == 0x80000000u) b = 0;   // -0.0 == +0.0 for pair_lt: the row decides
	b ^= (b >> 31) ? 0xFFFFFFFFu : 0x80000000u;
	return (static_cast<unsigned long long>(b) << 32) | row;
}
// ascending bitonic sort of n = 2^m (key, payload) pairs in LDS by all kMergeThreads threads
__device__ __forceinline__ void merge_bitonic(unsigned long long* key, float* val, uint32_t n) {
	for (uint32_t k = 2; k <= n; k <<= 1) {
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			for (uint32_t t = threadIdx.x; t < n / 2; t += kMergeThreads) {
				const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
				const unsigned long long x = key[i], y = key[l];
				if ((x > y) == ((i & k) == 0)) {
					key[i] = y;
					key[l] = x;
					const float vx = val[i];
					val[i] = val[l];
					val[l] = vx;
				}
			}
			__syncthreads();
		}
	}
}
template <typename TK>
__device__ void merge_by_insertion(const float* part_dist, const uint32_t* part_row, size_t base, uint32_t total, uint32_t kk, float* out_dist,
								   uint32_t* out_row, uint32_t* out_count, float* s_d, uint32_t* s_i);

template <typename TK>
__global__ __launch_bounds__(kMergeThreads) void knn_merge_lists(const float* part_dist, const uint32_t* part_row, uint32_t nlists, uint32_t kk,
																  float* out_dist, uint32_t* out_row, uint32_t* out_count) {
	__shared__ unsigned long long s_key[kMergeHeadsMax];
	__shared__ float s_val[kMergeHeadsMax];
	__shared__ uint32_t s_n;
	const uint32_t tid = threadIdx.x, total = nlists * kk;
	const size_t base = size_t(blockIdx.x) * total;
	bool serial = nlists < kk || nlists > kMergeHeadsMax;
	if (!serial) {
		uint32_t n1 = 64;
		while (n1 < nlists) n1 <<= 1;
		for (uint32_t l = tid; l < n1; l += kMergeThreads) {
			unsigned long long k = ~0ull;
			if (l < nlists) {
				const uint32_t r = part_row[base + size_t(l) * kk];
				if (r != kInvalidRow) k = merge_pair_key(part_dist[base + size_t(l) * kk], r);
			}
			s_key[l] = k;
			s_val[l] = 0.f;
		}
		if (tid == 0) s_n = 0;
		__syncthreads();
		merge_bitonic(s_key, s_val, n1);
		const unsigned long long bound = s_key[kk - 1];   // ~0: fewer than kk non-empty lists — everything valid is a candidate
		__syncthreads();
		// (the candidates overwrite the heads: every thread has read the bound)
		for (uint32_t l = tid; l < nlists; l += kMergeThreads) {
			const size_t at = base + size_t(l) * kk;
			for (uint32_t e = 0; e < kk; ++e) {
				const uint32_t r = part_row[at + e];
				if (r == kInvalidRow) break;
				const float d = part_dist[at + e];
				const unsigned long long k = merge_pair_key(d, r);
				if (k > bound) break;
				const uint32_t pos = atomicAdd(&s_n, 1u);
				if (pos < kMergeCandMax) {
					s_key[pos] = k;
					s_val[pos] = d;
				}
			}
		}
		__syncthreads();
		const uint32_t n = s_n;
		serial = n > kMergeCandMax;
		if (!serial) {
			uint32_t n2 = 64;
			while (n2 < n) n2 <<= 1;
			for (uint32_t q = n + tid; q < n2; q += kMergeThreads) {
				s_key[q] = ~0ull;
				s_val[q] = 0.f;
			}
			__syncthreads();
			merge_bitonic(s_key, s_val, n2);
			const uint32_t cnt = n < kk ? n : kk;
			for (uint32_t q = tid; q < kk; q += kMergeThreads) {
				out_dist[size_t(blockIdx.x) * kk + q] = q < cnt ? s_val[q] : __builtin_inff();
				out_row[size_t(blockIdx.x) * kk + q] = q < cnt ? uint32_t(s_key[q]) : kInvalidRow;
			}
			if (tid == 0 && out_count) out_count[blockIdx.x] = cnt;
			return;
		}
		__syncthreads();
	}
	// the insertion merge over the same lists (LDS reused for the wavefronts' lists)
	merge_by_insertion<TK>(part_dist, part_row, base, total, kk, out_dist, out_row, out_count, s_val, reinterpret_cast<uint32_t*>(s_key));
}

__device__ __forceinline__ void merge_store_list(const WaveTopK& t, float* d, uint32_t* i, int lane) {
	d[lane] = t.bd;
	i[lane] = t.bi;
	d[64 + lane] = __builtin_inff();
	i[64 + lane] = kInvalidRow;
}
__device__ __forceinline__ void merge_store_list(const WaveTopK2& t, float* d, uint32_t* i, int lane) {
	d[lane] = t.d0;
	i[lane] = t.i0;
	d[64 + lane] = t.d1;
	i[64 + lane] = t.i1;
}
template <typename TK>
__device__ void merge_by_insertion(const float* part_dist, const uint32_t* part_row, size_t base, uint32_t total, uint32_t kk, float* out_dist,
								   uint32_t* out_row, uint32_t* out_count, float* s_d, uint32_t* s_i) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	TK top;
	top.init(kk);
	for (uint32_t c0 = wave * kWave; c0 < total; c0 += kMergeThreads * kMergeAhead) {
		float cd[kMergeAhead];
		uint32_t ci[kMergeAhead];
#pragma unroll
		for (int u = 0; u < kMergeAhead; ++u) {
			const uint32_t c = c0 + uint32_t(u) * kMergeThreads + lane;
			cd[u] = __builtin_inff();
			ci[u] = kInvalidRow;
			if (c < total) {
				cd[u] = part_dist[base + c];
				ci[u] = part_row[base + c];
			}
		}
#pragma unroll
		for (int u = 0; u < kMergeAhead; ++u) {
			uint64_t pm = __ballot(ci[u] != kInvalidRow && top.admits(cd[u], ci[u]));
			while (pm) {
				const int src = __builtin_ctzll(pm);
				pm &= pm - 1;
				const float d = __shfl(cd[u], src);
				const uint32_t i = __shfl(ci[u], src);
				if (top.admits(d, i)) top.insert(d, i, lane);
			}
		}
	}
	merge_store_list(top, s_d + wave * kMaxFusedK2, s_i + wave * kMaxFusedK2, lane);
	__syncthreads();
	if (wave != 0) return;
	for (int w = 1; w < kMergeWaves; ++w) {
		for (uint32_t e = 0; e < kk; ++e) {   // sorted: once one entry is rejected the rest of that list is too
			const float d = s_d[w * kMaxFusedK2 + e];
			const uint32_t i = s_i[w * kMaxFusedK2 + e];
			if (i == kInvalidRow || !top.admits(d, i)) break;
			top.insert(d, i, lane);
		}
	}
	merge_store_list(top, s_d, s_i, lane);   // wave 0 only, its own slice
	const size_t o = size_t(blockIdx.x) * kk;
	if (lane < int(kk)) {
		out_dist[o + lane] = s_d[lane];
		out_row[o + lane] = s_i[lane];
	}
	if (64 + lane < int(kk)) {
		out_dist[o + 64 + lane] = s_d[64 + lane];
		out_row[o + 64 + lane] = s_i[64 + lane];
	}
	if (lane == 0 && out_count) out_count[blockIdx.x] = top.filled;
}

// Multi-GPU: fold the all-gathered per-shard lists of one query batch into the global top-kk.
// gathered: [world][2][nq][kk] 32-bit words — per shard the [nq][kk] distances followed by the [nq][kk] shard-local rows (what
// the scan writes when d_out_row == d_out_dist + nq*kk).  Global row = shard * shard_rows + local row; order = (dist, global row).
// slot_base (in-process RCCL path, rxgpu_sharded.hip): the global row base of every gathered position (rank-major, several shards per
// device, padded positions = kInvalidRow and skipped); null: position w is shard w.
// sorted = 0: the lists are UNORDERED sets (the per-shard HNSW results, members of top_candidates): every entry is offered.
__global__ __launch_bounds__(64) void knn_merge_shards(const uint32_t* gathered, uint32_t world, uint32_t nq, uint32_t kk, uint32_t shard_rows,
														const uint32_t* slot_base, float* out_dist, uint32_t* out_row, uint32_t* out_count, uint32_t sorted) {
	const int lane = threadIdx.x;
	const uint32_t q = blockIdx.x;
	WaveTopK top;
	top.init(kk);
	for (uint32_t w = 0; w < world; ++w) {
		const uint32_t* rec = gathered + size_t(w) * 2 * nq * kk + size_t(q) * kk;
		const uint32_t base = slot_base ? slot_base[w] : w * shard_rows;
		if (base == kInvalidRow) continue;
		float cd = __builtin_inff();
		uint32_t ci = kInvalidRow;
		if (lane < int(kk)) {
			cd = __uint_as_float(rec[lane]);
			const uint32_t local = rec[size_t(nq) * kk + lane];
			if (local != kInvalidRow) ci = base + local;
		}
		uint64_t pm = __ballot(ci != kInvalidRow);
		while (pm) {
			const int src = __builtin_ctzll(pm);
			pm &= pm - 1;
			const float d = __shfl(cd, src);
			const uint32_t i = __shfl(ci, src);
			if (!top.admits(d, i)) {
				if (sorted) break;   // each shard list is sorted
				continue;
			}
			top.insert(d, i, lane);
		}
	}
	if (lane < int(kk)) {
		out_dist[size_t(q) * kk + lane] = top.bd;
		out_row[size_t(q) * kk + lane] = top.bi;
	}
	if (lane == 0 && out_count) out_count[q] = top.filled;
}

// Sharded HNSW: one shard's search result ([nq][k] distances, [nq][k] local rows, [nq] counts <= k) into its slot of the exchange's send
// buffer — [nq][kk] distances | [nq][kk] rows, entries past a query's count = (+inf, kInvalidRow), which knn_merge_shards skips.
__global__ __launch_bounds__(256) void knn_pack_lists(const float* dist, const uint32_t* row, const uint32_t* count, uint32_t nq, uint32_t k, uint32_t kk,
													   uint32_t* dst_dist, uint32_t* dst_row) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq * kk) return;
	const uint32_t q = i / kk, j = i % kk;
	const bool have = j < k && j < count[q];
	dst_dist[i] = have ? __float_as_uint(dist[size_t(q) * k + j]) : 0x7F800000u;
	dst_row[i] = have ? row[size_t(q) * k + j] : kInvalidRow;
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

// SearchRange restricted to a row list (the list scan of an IVF range query, ivf_index.cc:212-272; p.n = list entries)
template <int kMetric>
__global__ __launch_bounds__(kScanThreads) void knn_range_subset(RangeParams p, const uint32_t* __restrict__ ids) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
	const uint64_t nquads = (p.n + kRowsPerWave - 1) / kRowsPerWave;
	const uint64_t nwaves = uint64_t(gridDim.x) * kScanWaves;
	for (uint64_t quad = uint64_t(blockIdx.x) * kScanWaves + wave; quad < nquads; quad += nwaves) {
		const uint64_t item = quad * kRowsPerWave + g;
		const bool valid = item < p.n;
		const uint64_t row = ids[valid ? item : p.n - 1];
		const float sum = group_distance_generic<kMetric>(p.rows + row * p.stride, p.query, p.dim, m);
		const float dist = metric_epilogue<kMetric>(sum, p.inv_norms, row);
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
	const uint32_t item = blockIdx.x * (256u / kGroup) + (threadIdx.x >> 4);   // 16 rows a block: no 32-bit thread-index product (n up to 2^32 - 1)
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
static void launch_scan_metric_wide(const ScanParams& p, dim3 grid, hipStream_t s) {
	const dim3 blk(kScanThreads);
	switch (p.dim) {
		case 128: hipLaunchKernelGGL((knn_scan_fixed<kMetric, 2, true, true, true, WaveTopK2>), grid, blk, 0, s, p); return;
		case 256: hipLaunchKernelGGL((knn_scan_fixed<kMetric, 4, true, true, true, WaveTopK2>), grid, blk, 0, s, p); return;
		case 512: hipLaunchKernelGGL((knn_scan_fixed<kMetric, 8, true, true, true, WaveTopK2>), grid, blk, 0, s, p); return;
		case 768: hipLaunchKernelGGL((knn_scan_fixed<kMetric, 12, true, true, true, WaveTopK2>), grid, blk, 0, s, p); return;
		case 1024: hipLaunchKernelGGL((knn_scan_fixed<kMetric, 16, true, true, true, WaveTopK2>), grid, blk, 0, s, p); return;
		default: hipLaunchKernelGGL((knn_scan_generic<kMetric, WaveTopK2>), grid, blk, 0, s, p);
	}
}

template <int kMetric>
static void launch_scan_metric(const ScanParams& p, dim3 grid, hipStream_t s) {
	if (p.kk > uint32_t(kMaxFusedK)) {
		launch_scan_metric_wide<kMetric>(p, grid, s);
		return;
	}
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

// ---- pre-filtered scan ----
constexpr uint64_t kSubsetChunk = 64;   // list entries per wavefront chunk of knn_scan_subset

static bool subset_fixed_dim(uint32_t dim, uint32_t kk) {
	if (kk > uint32_t(kMaxFusedK)) return dim == 512 || dim == 768;
	return dim == 128 || dim == 256 || dim == 512 || dim == 768 || dim == 1024;
}
// short lists go to the one-quad-per-step kernel (16x more wavefronts in flight); long ones to the chunked, double-buffered one
static bool subset_use_chunked(uint64_t n_ids, uint32_t dim, uint32_t kk, int cus) {
	const uint64_t waves = uint64_t(cus) * tuning().wg_per_cu * kScanWaves;
	return subset_fixed_dim(dim, kk) && n_ids >= 2 * waves * kSubsetChunk;
}
uint32_t subset_grid_x(uint64_t n_ids, uint32_t dim, uint32_t kk, int cus) {
	if (!subset_use_chunked(n_ids, dim, kk, cus)) return scan_grid_x(n_ids, cus);
	const uint64_t nchunks = (n_ids + kSubsetChunk - 1) / kSubsetChunk;
	const uint64_t want = (nchunks + kScanWaves - 1) / kScanWaves;
	const uint64_t cap = uint64_t(cus) * tuning().wg_per_cu;
	return uint32_t(want < cap ? want : cap);
}

template <int kMetric, typename TK>
static void launch_scan_subset_tk(const ScanParams& p, const uint32_t* ids, bool chunked, dim3 grid, hipStream_t s) {
	const dim3 blk(kScanThreads);
	if (chunked) {
		switch (p.dim) {
			case 128: if constexpr (std::is_same_v<TK, WaveTopK>) { hipLaunchKernelGGL((knn_scan_subset<kMetric, 2, TK>), grid, blk, 0, s, p, ids); return; } break;
			case 256: if constexpr (std::is_same_v<TK, WaveTopK>) { hipLaunchKernelGGL((knn_scan_subset<kMetric, 4, TK>), grid, blk, 0, s, p, ids); return; } break;
			case 512: hipLaunchKernelGGL((knn_scan_subset<kMetric, 8, TK>), grid, blk, 0, s, p, ids); return;
			case 768: hipLaunchKernelGGL((knn_scan_subset<kMetric, 12, TK>), grid, blk, 0, s, p, ids); return;
			case 1024: if constexpr (std::is_same_v<TK, WaveTopK>) { hipLaunchKernelGGL((knn_scan_subset<kMetric, 16, TK>), grid, blk, 0, s, p, ids); return; } break;
			default: break;
		}
	}
	hipLaunchKernelGGL((knn_scan_subset_generic<kMetric, TK>), grid, blk, 0, s, p, ids);
}
template <int kMetric>
static void launch_scan_subset_metric(const ScanParams& p, const uint32_t* ids, bool chunked, dim3 grid, hipStream_t s) {
	if (p.kk > uint32_t(kMaxFusedK)) {
		launch_scan_subset_tk<kMetric, WaveTopK2>(p, ids, chunked, grid, s);
	} else {
		launch_scan_subset_tk<kMetric, WaveTopK>(p, ids, chunked, grid, s);
	}
}
// p.n = number of list entries; gridx MUST come from subset_grid_x (it selects the kernel together with this function)
void launch_scan_subset(int metric, const ScanParams& p, const uint32_t* ids, uint32_t nq, uint32_t gridx, int cus, hipStream_t s) {
	const dim3 grid(gridx, nq);
	const bool chunked = subset_use_chunked(p.n, p.dim, p.kk, cus);
	switch (metric) {
		case kL2: launch_scan_subset_metric<kL2>(p, ids, chunked, grid, s); break;
		case kIP: launch_scan_subset_metric<kIP>(p, ids, chunked, grid, s); break;
		default: launch_scan_subset_metric<kCos>(p, ids, chunked, grid, s); break;
	}
}

template <int kMetric>
static bool launch_scan_bf16_metric(const ScanBf16Params& p, dim3 grid, hipStream_t s) {
	if (p.blocked) {   // the tile-blocked shadow: 16 rows per wavefront
		switch (p.ld / 128) {
			case 1: hipLaunchKernelGGL((knn_scan_bf16_blk<kMetric, 1>), grid, dim3(kScanThreads), 0, s, p); return true;
			case 2: hipLaunchKernelGGL((knn_scan_bf16_blk<kMetric, 2>), grid, dim3(kScanThreads), 0, s, p); return true;
			case 3: hipLaunchKernelGGL((knn_scan_bf16_blk<kMetric, 3>), grid, dim3(kScanThreads), 0, s, p); return true;
			case 4: hipLaunchKernelGGL((knn_scan_bf16_blk<kMetric, 4>), grid, dim3(kScanThreads), 0, s, p); return true;
			case 6: hipLaunchKernelGGL((knn_scan_bf16_blk<kMetric, 6>), grid, dim3(kScanThreads), 0, s, p); return true;
			case 8: hipLaunchKernelGGL((knn_scan_bf16_blk<kMetric, 8>), grid, dim3(kScanThreads), 0, s, p); return true;
			default: return false;
		}
	}
	switch (p.ld / 128) {   // ld is a multiple of 64; the per-lane chunk count must be whole (ld % 128 == 0)
		case 1: hipLaunchKernelGGL((knn_scan_bf16<kMetric, 1>), grid, dim3(kScanThreads), 0, s, p); return true;
		case 2: hipLaunchKernelGGL((knn_scan_bf16<kMetric, 2>), grid, dim3(kScanThreads), 0, s, p); return true;
		case 3: hipLaunchKernelGGL((knn_scan_bf16<kMetric, 3>), grid, dim3(kScanThreads), 0, s, p); return true;
		case 4: hipLaunchKernelGGL((knn_scan_bf16<kMetric, 4>), grid, dim3(kScanThreads), 0, s, p); return true;
		case 6: hipLaunchKernelGGL((knn_scan_bf16<kMetric, 6>), grid, dim3(kScanThreads), 0, s, p); return true;
		case 8: hipLaunchKernelGGL((knn_scan_bf16<kMetric, 8>), grid, dim3(kScanThreads), 0, s, p); return true;
		default: return false;
	}
}
bool scan_bf16_supported(uint32_t ld) {
	const uint32_t nc = ld / 128;
	return ld % 128 == 0 && (nc == 1 || nc == 2 || nc == 3 || nc == 4 || nc == 6 || nc == 8);
}
void launch_scan_bf16(int metric, const ScanBf16Params& p, uint32_t nq, uint32_t gridx, hipStream_t s) {
	const dim3 grid(gridx, nq);
	switch (metric) {
		case kL2: launch_scan_bf16_metric<kL2>(p, grid, s); break;
		case kIP: launch_scan_bf16_metric<kIP>(p, grid, s); break;
		default: launch_scan_bf16_metric<kCos>(p, grid, s); break;
	}
}
void launch_filter_approx(const float* approx, uint64_t n, const float* top_dist, const uint32_t* top_count, uint32_t kk, const float* margin,
						  uint32_t* cand_row, uint32_t* cand_cnt, uint32_t cap, uint32_t nq, int cus, hipStream_t s) {
	const uint32_t gx = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, uint64_t(cus) * 8)));
	hipLaunchKernelGGL(knn_filter_approx, dim3(gx, nq), dim3(256), 0, s, approx, n, top_dist, top_count, kk, margin, cand_row, cand_cnt, cap);
}

void launch_merge(const float* part_dist, const uint32_t* part_row, uint32_t total_per_query, uint32_t kk, uint32_t nq, float* out_dist,
				  uint32_t* out_row, uint32_t* out_count, const uint32_t* gate_cnt, uint32_t gate_cap, hipStream_t s) {
	if (kk > uint32_t(kMaxFusedK)) {   // 64 < kk <= 128: never gated (the batched / pruned paths keep kk <= 64)
		hipLaunchKernelGGL(knn_merge_wide, dim3(nq), dim3(kMergeThreads), 0, s, part_dist, part_row, total_per_query, kk, out_dist, out_row, out_count);
		return;
	}
	hipLaunchKernelGGL(knn_merge, dim3(nq), dim3(kMergeThreads), 0, s, part_dist, part_row, total_per_query, kk, out_dist, out_row, out_count,
					   gate_cnt, gate_cap);
}

// the partial results are SORTED lists of kk entries (nlists per query): knn_merge_lists
void launch_merge_lists(const float* part_dist, const uint32_t* part_row, uint32_t nlists, uint32_t kk, uint32_t nq, float* out_dist, uint32_t* out_row,
						uint32_t* out_count, hipStream_t s) {
	if (kk > uint32_t(kMaxFusedK)) {
		hipLaunchKernelGGL((knn_merge_lists<WaveTopK2>), dim3(nq), dim3(kMergeThreads), 0, s, part_dist, part_row, nlists, kk, out_dist, out_row, out_count);
	} else {
		hipLaunchKernelGGL((knn_merge_lists<WaveTopK>), dim3(nq), dim3(kMergeThreads), 0, s, part_dist, part_row, nlists, kk, out_dist, out_row, out_count);
	}
}

void launch_merge_shards(const uint32_t* gathered, uint32_t world, uint32_t nq, uint32_t kk, uint32_t shard_rows, float* out_dist,
						 uint32_t* out_row, uint32_t* out_count, hipStream_t s, const uint32_t* slot_base, bool sorted) {
	hipLaunchKernelGGL(knn_merge_shards, dim3(nq), dim3(64), 0, s, gathered, world, nq, kk, shard_rows, slot_base, out_dist, out_row, out_count, sorted ? 1u : 0u);
}

void launch_pack_lists(const float* dist, const uint32_t* row, const uint32_t* count, uint32_t nq, uint32_t k, uint32_t kk, uint32_t* dst_dist, uint32_t* dst_row,
					   hipStream_t s) {
	const uint32_t total = nq * kk;
	if (!total) return;
	hipLaunchKernelGGL(knn_pack_lists, dim3((total + 255) / 256), dim3(256), 0, s, dist, row, count, nq, k, kk, dst_dist, dst_row);
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

void launch_range_subset(int metric, const float* rows, const float* inv_norms, const float* query, const uint32_t* ids, uint64_t n_ids,
						 uint32_t stride, uint32_t dim, float radius, int inclusive, float* out_dist, uint32_t* out_row, uint64_t cap,
						 unsigned long long* counter, uint32_t gridx, hipStream_t s) {
	RangeParams p{rows, inv_norms, query, n_ids, stride, dim, radius, inclusive, out_dist, out_row, cap, counter};
	switch (metric) {
		case kL2: hipLaunchKernelGGL((knn_range_subset<kL2>), dim3(gridx), dim3(kScanThreads), 0, s, p, ids); break;
		case kIP: hipLaunchKernelGGL((knn_range_subset<kIP>), dim3(gridx), dim3(kScanThreads), 0, s, p, ids); break;
		default: hipLaunchKernelGGL((knn_range_subset<kCos>), dim3(gridx), dim3(kScanThreads), 0, s, p, ids); break;
	}
}

void launch_distances(int metric, const float* rows, const float* inv_norms, const float* query, uint32_t stride, uint32_t dim,
					  const uint32_t* ids, uint32_t n, float* out, hipStream_t s) {
	const uint32_t blocks = uint32_t((uint64_t(n) * kGroup + 255) / 256);   // 64-bit product: n * 16 wraps from 2^28 rows on
	switch (metric) {
		case kL2: hipLaunchKernelGGL((knn_distances<kL2>), dim3(blocks), dim3(256), 0, s, rows, inv_norms, query, stride, dim, ids, n, out); break;
		case kIP: hipLaunchKernelGGL((knn_distances<kIP>), dim3(blocks), dim3(256), 0, s, rows, inv_norms, query, stride, dim, ids, n, out); break;
		default: hipLaunchKernelGGL((knn_distances<kCos>), dim3(blocks), dim3(256), 0, s, rows, inv_norms, query, stride, dim, ids, n, out); break;
	}
}

}  // namespace rxgpu
