// Batched queries x corpus on the matrix cores (BASELINE configs[1], batch = 256 "MFMA path").
//
// The reference has no batched API: a batch is B sequential BruteforceSearch::SearchKnn calls
// (cpp_src/core/index/float_vector/hnswlib/bruteforce.cc:103-127).  Here the B x N score matrix is an fp32 GEMM on
// v_mfma_f32_32x32x2_f32 (exact f32, but ONE k-ordered fmaf chain per element — not the reference's 64-chain order),
// so the GEMM only NOMINATES candidates and the exact kernels decide:
//   1. knn_row_stats        max |x| (and per-row |x|^2 for L2) -> rigorous rounding bound eps_q = gamma_D * |q| * max|x|
//   2. knn_gemm<DENSE>      approximate distances of every query against a row sample
//   3. knn_sample_threshold thr_q = kk-th best sample distance + 2 eps_q   (any row of the true top-kk must pass it)
//   4. knn_gemm<FILTER>     stream the whole corpus once; append rows with approx dist <= thr_q to per-query lists
//   5. knn_rescore          EXACT distances (knn_kernels.hip.h, bit-identical to the reference) of the nominated rows
//   6. knn_merge            exact top-kk by (dist,row)
// A query whose list overflows (adversarial data: massive ties) is redone by the exact fused scan, gated on device.
// Result: the same ids and distance bits as B sequential reference calls, at one corpus read per <=256 queries.
#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmThreads = 256;     // per query split: 4 wavefronts, wave w owns rows [32(w&3), +32) of the tile and 1/QS of the MT queries
constexpr int kGemmRows = 128;        // corpus rows per tile
constexpr int kGemmKS = 32;           // floats of the dimension staged per step
constexpr int kLdsStride = kGemmKS + 1;   // +1 pad: (row + k) mod 32 banks, conflict-free fragment reads

// QS = 2 runs 8 wavefronts per workgroup (two per SIMD): while one waits on LDS staging / the barrier the other keeps the
// matrix pipe busy, and each wave carries half the accumulators.
template <int kMetric, int MT, int kMode, int QS>
__global__ __launch_bounds__(kGemmThreads * QS) void knn_gemm(GemmParams p) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	constexpr int kThreads = kGemmThreads * QS;
	constexpr int QB = MT / 32 / QS;                  // 32-query blocks per wave
	constexpr int kQTile = MT * kLdsStride;           // floats per Q buffer
	constexpr int kXTile = kGemmRows * kLdsStride;
	float* q_s = lds;                                 // [2][MT][33]
	float* x_s = lds + 2 * kQTile;                    // [2][128][33]
	float* thr_s = x_s + 2 * kXTile;                  // [MT] thresholds (FILTER)
	float* aux_s = thr_s + MT;                        // [MT] |q|^2 (L2)
	constexpr int kQLoads = MT * (kGemmKS / 4) / kThreads;          // float4 per thread per step
	constexpr int kXLoads = kGemmRows * (kGemmKS / 4) / kThreads;
	static_assert(kQLoads >= 1 && kXLoads >= 1, "tile too small for the thread count");

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wave = wave_all & 3;                    // row block
	const int qs = wave_all >> 2;                     // query split
	const uint32_t ksteps = (p.dim + kGemmKS - 1) / kGemmKS;
	const uint64_t ntiles = (p.n + kGemmRows - 1) / kGemmRows;
	for (int i = tid; i < MT; i += kThreads) {
		thr_s[i] = kMode == kGemmFilter ? p.thr[i] : 0.f;
		aux_s[i] = kMetric == kL2 ? p.q_sq[i] : 0.f;
	}

	for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		const uint64_t row0 = tile * kGemmRows;
		f32x16 acc[QB];
#pragma unroll
		for (int b = 0; b < QB; ++b) {
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
		}
		float4 qreg[kQLoads], xreg[kXLoads];

		auto load_step = [&](uint32_t ks) {
			const uint32_t k0 = ks * kGemmKS;
#pragma unroll
			for (int i = 0; i < kQLoads; ++i) {
				const int idx = tid + i * kThreads;
				const uint32_t qi = idx >> 3, k = k0 + ((idx & 7) << 2);
				qreg[i] = *reinterpret_cast<const float4*>(p.queries + size_t(qi) * p.q_stride + k);   // padded: always in range
			}
#pragma unroll
			for (int i = 0; i < kXLoads; ++i) {
				const int idx = tid + i * kThreads;
				const uint64_t r = row0 + (idx >> 3);
				const uint32_t k = k0 + ((idx & 7) << 2);
				float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
				if (r < p.n) {
					const float* src = p.rows + r * p.row_step * p.stride + k;
					if (k + 3 < p.dim) {
						v = load_row4<true>(reinterpret_cast<const float4*>(src));
					} else {
						if (k < p.dim) v.x = src[0];
						if (k + 1 < p.dim) v.y = src[1];
						if (k + 2 < p.dim) v.z = src[2];
					}
				}
				xreg[i] = v;
			}
		};
		auto store_step = [&](int buf) {
#pragma unroll
			for (int i = 0; i < kQLoads; ++i) {
				const int idx = tid + i * kThreads;
				float* d = q_s + buf * kQTile + (idx >> 3) * kLdsStride + ((idx & 7) << 2);
				d[0] = qreg[i].x;
				d[1] = qreg[i].y;
				d[2] = qreg[i].z;
				d[3] = qreg[i].w;
			}
#pragma unroll
			for (int i = 0; i < kXLoads; ++i) {
				const int idx = tid + i * kThreads;
				float* d = x_s + buf * kXTile + (idx >> 3) * kLdsStride + ((idx & 7) << 2);
				d[0] = xreg[i].x;
				d[1] = xreg[i].y;
				d[2] = xreg[i].z;
				d[3] = xreg[i].w;
			}
		};

		__syncthreads();   // previous tile's last reads are done before buffer 0 is overwritten
		load_step(0);
		store_step(0);
		__syncthreads();
		for (uint32_t ks = 0; ks < ksteps; ++ks) {
			const int buf = ks & 1;
			if (ks + 1 < ksteps) load_step(ks + 1);
			const float* xb = x_s + buf * kXTile + (32 * wave + (lane & 31)) * kLdsStride + (lane >> 5);
			const float* qb = q_s + buf * kQTile + (qs * QB * 32 + (lane & 31)) * kLdsStride + (lane >> 5);
#pragma unroll 4
			for (int kk = 0; kk < kGemmKS / 2; ++kk) {
				const float bfrag = xb[2 * kk];
#pragma unroll
				for (int b = 0; b < QB; ++b) {
					const float afrag = qb[b * 32 * kLdsStride + 2 * kk];
					acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag, bfrag, acc[b], 0, 0, 0);
				}
			}
			if (ks + 1 < ksteps) store_step(buf ^ 1);
			__syncthreads();
		}

		// epilogue: element (query i, row j): j = lane&31, i = 32b + (r&3) + 8(r>>2) + 4(lane>>5)
		const uint64_t row = row0 + 32 * wave + (lane & 31);
		const bool row_ok = row < p.n;
		const uint64_t rowc = (row_ok ? row : p.n - 1) * p.row_step;
		const int qlane = 4 * (lane >> 5) + qs * QB * 32;
		float row_term = 0.f;   // per-row factor of the approximate distance
		if constexpr (kMetric == kL2) row_term = p.row_sq[rowc];
		if constexpr (kMetric == kCos) row_term = p.inv_norms[rowc];
		if constexpr (kMode == kGemmDense) {
			float* dp = p.dense + size_t(qlane) * p.n + row;   // running pointer, kept opaque so that 16*QB addresses are not precomputed
			const size_t n1 = p.n, n5 = 5 * p.n;
#pragma unroll
			for (int b = 0; b < QB; ++b) {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const int qo = 32 * b + (r & 3) + 8 * (r >> 2);
					float d;
					if constexpr (kMetric == kL2) {
						d = (aux_s[qo + qlane] + row_term) - 2.0f * acc[b][r];
					} else if constexpr (kMetric == kIP) {
						d = -acc[b][r];
					} else {
						d = -acc[b][r] * row_term;
					}
					if (row_ok) *dp = d;
					dp += ((r & 3) == 3) ? n5 : n1;
					asm volatile("" : "+v"(dp));
				}
			}
		} else {
#pragma unroll
			for (int b = 0; b < QB; ++b) {
				uint32_t mask = 0;   // bit r: element r of this 32-query block passes its query's threshold
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const int qo = 32 * b + (r & 3) + 8 * (r >> 2);
					float d;
					if constexpr (kMetric == kL2) {
						d = (aux_s[qo + qlane] + row_term) - 2.0f * acc[b][r];
					} else if constexpr (kMetric == kIP) {
						d = -acc[b][r];
					} else {
						d = -acc[b][r] * row_term;
					}
					mask |= (d <= thr_s[qo + qlane]) ? (1u << r) : 0u;   // padded queries carry thr = -inf
				}
				if (!row_ok) mask = 0;
				if (__ballot(mask != 0)) {   // rare once the thresholds are tight
					while (mask) {
						const int r = __builtin_ctz(mask);
						mask &= mask - 1;
						const uint32_t qi = 32 * b + (r & 3) + 8 * (r >> 2) + qlane;
						const uint32_t pos = atomicAdd(&p.cand_cnt[qi], 1u);
						if (pos < p.cap) p.cand_row[size_t(qi) * p.cap + pos] = uint32_t(row);
					}
				}
			}
		}
	}
}

// ---- per-row statistics: |x|^2 (L2 needs it per row) and the maxima entering the rounding bound ----
// stats[0] = max |x|^2, stats[1] = max (|x| * inv_norm)^2 (cosine)
__global__ __launch_bounds__(256) void knn_row_stats(const float* rows, const float* inv_norms, uint64_t n, uint32_t stride, uint32_t dim,
													  float* row_sq, unsigned int* stats) {
	const int lane = threadIdx.x & 63, m = lane & 15;
	const uint64_t ngroups = uint64_t(gridDim.x) * (blockDim.x / kGroup);
	const uint64_t gid = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
	const uint64_t rounds = (n + ngroups - 1) / ngroups;
	float mx = 0.f, mxc = 0.f;
	for (uint64_t it = 0; it < rounds; ++it) {   // uniform trip count: every lane takes part in the shuffles
		const uint64_t row = it * ngroups + gid;
		const bool ok = row < n;
		const float* r = rows + (ok ? row : n - 1) * stride;
		float s = 0.f;
		for (uint32_t i = m; i < dim; i += kGroup) s = __builtin_fmaf(r[i], r[i], s);
		s += __shfl_xor(s, 1);
		s += __shfl_xor(s, 2);
		s += __shfl_xor(s, 4);
		s += __shfl_xor(s, 8);
		if (ok) {
			if (row_sq && m == 0) row_sq[row] = s;
			mx = fmaxf(mx, s);
			if (inv_norms) {
				const float inv = inv_norms[row];
				mxc = fmaxf(mxc, s * inv * inv);
			}
		}
	}
	for (int o = 32; o; o >>= 1) {
		mx = fmaxf(mx, __shfl_xor(mx, o));
		mxc = fmaxf(mxc, __shfl_xor(mxc, o));
	}
	if (lane == 0) {   // non-negative floats order like their bit patterns
		atomicMax(&stats[0], __float_as_uint(mx));
		atomicMax(&stats[1], __float_as_uint(mxc));
	}
}

// |q|^2 and the per-query margin 2*eps_q (see file header); one 64-lane wave per query
// kBf16: the nomination runs on bf16-rounded operands (knn_batched_bf16.hip): + (2^-8 + 2^-18)|q||x| for the two roundings
template <int kMetric, bool kBf16>
__global__ __launch_bounds__(64) void knn_query_stats(const float* queries, uint32_t nq, uint32_t q_stride, uint32_t dim,
													   const unsigned int* stats, float* q_sq, float* margin) {
	const uint32_t qi = blockIdx.x;
	const int lane = threadIdx.x;
	float s = 0.f;
	if (qi < nq) {
		const float* q = queries + size_t(qi) * q_stride;
		for (uint32_t i = lane; i < dim; i += 64) s = __builtin_fmaf(q[i], q[i], s);
	}
	for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
	if (lane == 0) {
		q_sq[qi] = s;
		const float u = 5.9604645e-08f;   // 2^-24
		// f32 accumulation: covers both summation trees (D-chain vs 64-chain + fold), 10% slack.  The bf16 MFMA adds 16 products per instruction in an
		// adder tree whose internal rounding mode is not documented: allow 2 ulp-halves per addition and the tree depth on top of the chain (4x)
		const float gamma = (kBf16 ? 4.4f : 1.1f) * float(dim + 64) * u;
		// rne_bf16 on q and x: |q~.x~ - q.x| <= ((1+2^-9)^2 - 1) sum|q_i x_i| <= 2^-8 (1 + 2^-10) |q||x|   (only the inner product is affected:
		// |q|^2 and |x|^2 of the L2 form come from the f32 data)
		const float gb = kBf16 ? 1.01f * 0.00390625f : 0.0f;
		const float xmax2 = __uint_as_float(stats[0]);
		float eps;
		if constexpr (kMetric == kL2) {
			// d = (qq + xx) - 2 ip: 2*gamma*|q||x| <= gamma*(qq+xx), plus the roundings of qq, xx and of the reference's own sum; bf16 adds 2*gb*|q||x|
			eps = 2.0f * gamma * (s + xmax2) + 2.0f * gb * sqrtf(s) * sqrtf(xmax2);
		} else if constexpr (kMetric == kIP) {
			eps = (gamma + gb) * sqrtf(s) * sqrtf(xmax2);
		} else {
			eps = (gamma + gb + 4.0f * u) * sqrtf(s) * sqrtf(__uint_as_float(stats[1]));
		}
		// bf16 MFMA may flush subnormal inputs: at most dim * 2^-126 * (|q| + max|x|), far below the 1e-30 floor added here
		margin[qi] = 2.0f * eps * 1.01f + (kBf16 ? 1e-30f : 1e-37f);
	}
}

// thr[q] = kk-th smallest of dense[q][0..ns) + margin[q]; one workgroup per query (fewer than kk samples: +inf)
__global__ __launch_bounds__(256) void knn_sample_threshold(const float* dense, uint64_t ns, uint32_t nq, uint32_t kk, const float* margin,
															 float* thr) {
	__shared__ float s_d[4][kMaxFusedK];
	__shared__ uint32_t s_i[4][kMaxFusedK];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t qi = blockIdx.x;
	WaveTopK top;
	top.init(kk);
	if (qi < nq) {
		const float* d = dense + size_t(qi) * ns;
		for (uint64_t c0 = uint64_t(wave) * 64; c0 < ns; c0 += 256) {
			const uint64_t c = c0 + lane;
			const float cd = c < ns ? d[c] : __builtin_inff();
			uint64_t pm = __ballot(c < ns && top.admits(cd, uint32_t(c)));
			while (pm) {
				const int src = __builtin_ctzll(pm);
				pm &= pm - 1;
				const float dd = __shfl(cd, src);
				const uint32_t ii = uint32_t(c0) + src;
				if (top.admits(dd, ii)) top.insert(dd, ii, lane);
			}
		}
	}
	s_d[wave][lane] = top.bd;
	s_i[wave][lane] = top.bi;
	__syncthreads();
	if (wave != 0) return;
	for (int w = 1; w < 4; ++w) {
		const float cd = s_d[w][lane];
		const uint32_t ci = s_i[w][lane];
		uint64_t pm = __ballot(ci != kInvalidRow && lane < int(kk));
		while (pm) {
			const int src = __builtin_ctzll(pm);
			pm &= pm - 1;
			const float dd = __shfl(cd, src);
			const uint32_t ii = __shfl(ci, src);
			if (!top.admits(dd, ii)) break;
			top.insert(dd, ii, lane);
		}
	}
	if (lane == 0) thr[qi] = (qi < nq && top.filled == kk) ? top.thr_d + margin[qi] : (qi < nq ? __builtin_inff() : -__builtin_inff());
}

// EXACT distances of the nominated rows: one 16-lane group per candidate slot; unused slots are invalidated
template <int kMetric>
__global__ __launch_bounds__(256) void knn_rescore(const float* rows, const float* inv_norms, const float* queries, uint32_t q_stride,
													uint32_t stride, uint32_t dim, uint32_t cap, const uint32_t* cand_cnt, uint32_t* cand_row,
													float* cand_dist) {
	const int lane = threadIdx.x & 63, m = lane & 15;
	const uint32_t qi = blockIdx.y;
	const uint32_t cnt = min(cand_cnt[qi], cap);
	const uint32_t slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	if (slot >= cap) return;
	const size_t o = size_t(qi) * cap + slot;
	if (slot >= cnt) {
		if (m == 0) {
			cand_row[o] = kInvalidRow;
			cand_dist[o] = __builtin_inff();
		}
		return;
	}
	const uint64_t row = cand_row[o];
	const float sum = group_distance_generic<kMetric>(rows + row * stride, queries + size_t(qi) * q_stride, dim, m);
	const float dist = metric_epilogue<kMetric>(sum, inv_norms, row);
	if (m == 0) cand_dist[o] = dist;
}

// ------------------------------------------------------------------------------------------ launchers

size_t gemm_lds_bytes(int mt) { return (size_t(2) * (mt + kGemmRows) * kLdsStride + 2 * size_t(mt)) * sizeof(float); }

template <int kMetric, int MT, int kMode>
static hipError_t launch_gemm_one(const GemmParams& p, uint32_t grid, hipStream_t s) {
	constexpr int QS = MT >= 128 ? 2 : 1;
	const size_t lds = gemm_lds_bytes(MT);
	static std::atomic<uint64_t> raised{0};
	if (hipError_t e = raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&knn_gemm<kMetric, MT, kMode, QS>), lds); e != hipSuccess) return e;
	hipLaunchKernelGGL((knn_gemm<kMetric, MT, kMode, QS>), dim3(grid), dim3(kGemmThreads * QS), lds, s, p);
	return hipGetLastError();
}

template <int kMetric, int kMode>
static hipError_t launch_gemm_mt(int mt, const GemmParams& p, uint32_t grid, hipStream_t s) {
	switch (mt) {
		case 32: return launch_gemm_one<kMetric, 32, kMode>(p, grid, s);
		case 64: return launch_gemm_one<kMetric, 64, kMode>(p, grid, s);
		case 128: return launch_gemm_one<kMetric, 128, kMode>(p, grid, s);
		default: return launch_gemm_one<kMetric, 256, kMode>(p, grid, s);
	}
}

hipError_t launch_gemm(int metric, int mt, int mode, const GemmParams& p, uint32_t grid, hipStream_t s) {
	if (mode == kGemmDense) {
		switch (metric) {
			case kL2: return launch_gemm_mt<kL2, kGemmDense>(mt, p, grid, s);
			case kIP: return launch_gemm_mt<kIP, kGemmDense>(mt, p, grid, s);
			default: return launch_gemm_mt<kCos, kGemmDense>(mt, p, grid, s);
		}
	}
	switch (metric) {
		case kL2: return launch_gemm_mt<kL2, kGemmFilter>(mt, p, grid, s);
		case kIP: return launch_gemm_mt<kIP, kGemmFilter>(mt, p, grid, s);
		default: return launch_gemm_mt<kCos, kGemmFilter>(mt, p, grid, s);
	}
}

void launch_row_stats(const float* rows, const float* inv_norms, uint64_t n, uint32_t stride, uint32_t dim, float* row_sq,
					  unsigned int* stats, int cus, hipStream_t s) {
	uint64_t blocks = (n * kGroup + 255) / 256;
	const uint64_t cap = uint64_t(cus) * 8;
	if (blocks > cap) blocks = cap;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL(knn_row_stats, dim3(uint32_t(blocks)), dim3(256), 0, s, rows, inv_norms, n, stride, dim, row_sq, stats);
}

void launch_query_stats(int metric, const float* queries, uint32_t nq, uint32_t mt, uint32_t q_stride, uint32_t dim, const unsigned int* stats,
						float* q_sq, float* margin, bool bf16, hipStream_t s) {
#define RX_QS(M, B) hipLaunchKernelGGL((knn_query_stats<M, B>), dim3(mt), dim3(64), 0, s, queries, nq, q_stride, dim, stats, q_sq, margin)
	if (bf16) {
		switch (metric) {
			case kL2: RX_QS(kL2, true); break;
			case kIP: RX_QS(kIP, true); break;
			default: RX_QS(kCos, true); break;
		}
	} else {
		switch (metric) {
			case kL2: RX_QS(kL2, false); break;
			case kIP: RX_QS(kIP, false); break;
			default: RX_QS(kCos, false); break;
		}
	}
#undef RX_QS
}

void launch_sample_threshold(const float* dense, uint64_t ns, uint32_t nq, uint32_t mt, uint32_t kk, const float* margin, float* thr,
							 hipStream_t s) {
	hipLaunchKernelGGL(knn_sample_threshold, dim3(mt), dim3(256), 0, s, dense, ns, nq, kk, margin, thr);
}

void launch_rescore(int metric, const float* rows, const float* inv_norms, const float* queries, uint32_t q_stride, uint32_t stride,
					uint32_t dim, uint32_t nq, uint32_t cap, const uint32_t* cand_cnt, uint32_t* cand_row, float* cand_dist, hipStream_t s) {
	const dim3 grid((cap * kGroup + 255) / 256, nq);
	switch (metric) {
		case kL2: hipLaunchKernelGGL((knn_rescore<kL2>), grid, dim3(256), 0, s, rows, inv_norms, queries, q_stride, stride, dim, cap, cand_cnt, cand_row, cand_dist); break;
		case kIP: hipLaunchKernelGGL((knn_rescore<kIP>), grid, dim3(256), 0, s, rows, inv_norms, queries, q_stride, stride, dim, cap, cand_cnt, cand_row, cand_dist); break;
		default: hipLaunchKernelGGL((knn_rescore<kCos>), grid, dim3(256), 0, s, rows, inv_norms, queries, q_stride, stride, dim, cap, cand_cnt, cand_row, cand_dist); break;
	}
}

}  // namespace rxgpu
