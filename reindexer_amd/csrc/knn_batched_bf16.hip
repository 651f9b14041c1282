// Batched queries x corpus, nomination pass on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the f32-input MFMA rate).
//
// Same contract as knn_batched.hip: the GEMM only NOMINATES rows under a rigorous error bound, the exact kernels
// (knn_rescore -> knn_merge, bit-identical to the reference's 64-chain f32 arithmetic) decide.  Because the nomination is allowed
// to be approximate as long as the bound is sound, it can run on a bf16 shadow copy of the corpus:
//     x~ = rne_bf16(x), q~ = rne_bf16(q):  |q~.x~ - q.x| <= ((1+2^-9)^2 - 1) * sum|q_i x_i| <= (2^-8 + 2^-18) |q||x|
//     products of bf16 pairs are exact in f32; the f32 accumulation adds gamma_D * sum|q~_i x~_i|      (knn_query_stats<.., true>)
// so eps_q = (2^-8 * 1.01 + gamma_D) |q| max|x| and every row of the true top-kk passes thr_q = kk-th best sample + 2 eps_q.
// Cost of the looser bound: ~3x more nominated rows to re-score exactly (a few thousand 3 KB gathers per query); gain: the
// corpus pass reads 2 bytes per element instead of 4 and runs on the bf16 MFMA pipe, so a 256-query batch over 10M x 768 is bounded
// by ~2 ms of HBM (15.4 GB shadow) / ~1.6 ms of MFMA (3.93 TFLOP at 2.5 PFLOP/s) instead of 25 ms of f32-input MFMA.
//
// Tile: 256 corpus rows x 256 queries per workgroup (512 threads = 8 waves as 4 row-pairs x 2 query halves; each wave owns
// 2 x 4 blocks of 32x32 -> 128 accumulator VGPRs).  Both MFMA operands want 8 k-contiguous bf16 per lane = one ds_read_b128 from a
// row-major LDS image.  A first version staged the operands through registers (8.1 -> 7.3 ms per 256 queries once its loads were made
// unconditional and pinned ahead of the MFMAs); HBM latency (~2 us) is several times what one K-stage takes to multiply, so the kernel
// below keeps three stages in flight by LDS-DMA instead (4.8 ms).
#include <cstdlib>

#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBfThreads = 512;
constexpr int kBfRows = 256;       // corpus rows per tile

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
	uint32_t u = __float_as_uint(f);
	u += 0x7FFFu + ((u >> 16) & 1u);
	return uint16_t(u >> 16);
}

// rows [n][stride] f32 -> bf16 rows of ld elements (ld = dim rounded up to 64, zero padded): the shadow of corpus rows first_row .. (in
// its layout, knn_kernels.hip.h), or — blocked = 0, first_row = 0 — the padded query block
__global__ __launch_bounds__(256) void knn_to_bf16(const float* src, uint64_t n, uint32_t stride, uint32_t dim, uint16_t* dst, uint32_t ld, uint64_t first_row,
													uint32_t blocked) {
	const uint32_t chunks = ld / 8;   // 8 elements (16 B out) per thread
	const uint64_t total = n * chunks;
	const uint32_t step = shadow_stage_step(blocked != 0);
	for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
		const uint64_t row = i / chunks;
		const uint32_t k = uint32_t(i % chunks) * 8;
		const float* s = src + row * stride + k;
		uint32_t w[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float a = k + 2 * j < dim ? s[2 * j] : 0.f;
			const float b = k + 2 * j + 1 < dim ? s[2 * j + 1] : 0.f;
			w[j] = uint32_t(f32_to_bf16_rne(a)) | (uint32_t(f32_to_bf16_rne(b)) << 16);
		}
		*reinterpret_cast<uint4*>(dst + shadow_elem_base(first_row + row, ld, blocked != 0) + uint64_t(k / 32) * step + k % 32) = make_uint4(w[0], w[1], w[2], w[3]);
	}
}

// one row of the shadow onto another (swap-delete)
__global__ __launch_bounds__(256) void knn_shadow_move(uint16_t* shadow, uint32_t ld, uint64_t from, uint64_t to, uint32_t blocked) {
	const uint32_t step = shadow_stage_step(blocked != 0);
	for (uint32_t k = threadIdx.x * 8; k < ld; k += 256 * 8) {
		const uint64_t o = uint64_t(k / 32) * step + k % 32;
		*reinterpret_cast<uint4*>(shadow + shadow_elem_base(to, ld, blocked != 0) + o) = *reinterpret_cast<const uint4*>(shadow + shadow_elem_base(from, ld, blocked != 0) + o);
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------
// glds variant: both operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4), no staging registers, no ds_write pass.
// One stage = 32 bf16 of the dimension: 256 rows x 64 B + 256 queries x 64 B = 32 KB; 4 LDS buffers, 3 stages in flight, so ~3000
// MFMA cycles (~1.3 us) of HBM latency are covered; the stream of stages runs ACROSS tiles (the next tile's first stages are already
// landing during this tile's epilogue).  (A 4-wave / 512-register variant with 128 x 128 wave tiles measured 6.2 ms against 4.9 ms: with one
// wave per SIMD nothing overlaps its fragment reads with its MFMAs.)  The DMA writes a lane-linear image (wave base + lane x 16 B), so the bank swizzle is applied on
// the SOURCE side: LDS slot p = 4 r + cs holds chunk c = cs ^ ((r >> 2) & 3) of row r; a ds_read_b128 lane group (16 rows, one chunk) then
// covers 16 distinct 16-byte slots.  One raw s_barrier per stage with counted vmcnt (a __syncthreads would drain the DMA queue).
constexpr int kGlXElems = kBfRows * 32;   // the row operand of one stage: 256 rows x 32 bf16 (16 KB)
// QT = queries per tile.  256: 32 KB per stage, 4 buffers, 3 stages in flight.  128 (batches <= 128 queries): 24 KB per stage, 6 buffers,
// 5 stages in flight — the kernel waits on HBM latency, so the bytes in flight per CU are what the narrower query block buys.
constexpr int gl_bufs(int qt) { return qt == 256 ? 4 : 6; }
// Nominations of the filter pass are collected in LDS, one list per wavefront, and reach the global per-query lists in batches: a returning
// global atomic in the stream of LDS-DMAs makes its wave wait for every DMA issued before it (vmcnt counts in order), i.e. drains the
// stages in flight — with ~3 nominations per wave and tile that happened in nearly every epilogue.  LDS atomics count on lgkmcnt instead.
constexpr int kGlHitCap = 192;   // entries (query << 32 | row) per wavefront
typedef __attribute__((address_space(3))) void lds_void;

template <int kMetric, int kMode, int QT>
__global__ __launch_bounds__(kBfThreads) void knn_gemm_bf16_glds(GemmBf16Params p) {
	constexpr int kBufs = gl_bufs(QT), kAhead = kBufs - 1;
	constexpr int kQElems = QT * 32;                 // the query operand of one stage
	constexpr int kStageElems = kGlXElems + kQElems;
	constexpr int QB = QT / 64;                      // 32-query blocks per wave (two query halves)
	constexpr int kQDma = QT * 4 / kBfThreads;       // DMA instructions per thread for the query part (2 or 1)
	constexpr int kIps = 2 + kQDma;                  // DMA instructions per stage per wave
	extern __shared__ __attribute__((aligned(16))) unsigned char bf_lds[];   // the ONLY shared object (a second one de-pipelines the DMA waits)
	uint16_t* stage_s = reinterpret_cast<uint16_t*>(bf_lds);                            // [kBufs][x: 256 x 32 | q: QT x 32]
	float* thr_s = reinterpret_cast<float*>(bf_lds + size_t(kBufs) * kStageElems * 2);   // [QT]
	float* aux_s = thr_s + QT;
	uint32_t* hit_n = reinterpret_cast<uint32_t*>(aux_s + QT);                            // [8]: one counter per wavefront
	unsigned long long* hit_all = reinterpret_cast<unsigned long long*>(hit_n + 8);       // [8][kGlHitCap]
	float* gmax_s = reinterpret_cast<float*>(hit_all + size_t(8) * kGlHitCap);           // [QT / 16] loosest threshold of a lane's 16 queries
	float* gmin_s = gmax_s + QT / 16;                                                     // [QT / 16] smallest |q|^2 of them (L2)

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int rp = wave & 3, qh = wave >> 2;
	const uint32_t stages = p.ld / 32;
	const uint64_t ntiles = (p.n + kBfRows - 1) / kBfRows;
	const uint64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
	const uint64_t total = my_tiles * stages;
	for (int i = tid; i < QT; i += kBfThreads) {
		thr_s[i] = kMode == kGemmFilter ? p.thr[i] : 0.f;
		aux_s[i] = kMetric == kL2 ? p.q_sq[i] : 0.f;
	}
	if (tid < 8) hit_n[tid] = 0;
	__syncthreads();
	// block-level test of the filter epilogue: group gi = (qh * QB + b) * 2 + (lane >> 5) holds the 16 queries one lane compares in block b
	if (kMode == kGemmFilter && tid < QT / 16) {
		const int half = tid & 1, bb = (tid >> 1) % QB, hh = (tid >> 1) / QB;
		float tmax = -__builtin_inff(), amin = __builtin_inff();
		for (int r = 0; r < 16; ++r) {
			const int qi = 32 * bb + (r & 3) + 8 * (r >> 2) + 4 * half + (QT / 2) * hh;
			tmax = fmaxf(tmax, thr_s[qi]);
			amin = fminf(amin, aux_s[qi]);
			if (thr_s[qi] != thr_s[qi]) tmax = __builtin_inff();   // a NaN threshold admits nothing per element; never let it hide the others
		}
		gmax_s[tid] = tmax;
		gmin_s[tid] = amin;
	}
	__syncthreads();
	unsigned long long* hit_s = hit_all + size_t(wave) * kGlHitCap;
	uint32_t* my_n = hit_n + wave;
	auto flush_hits = [&]() {   // this wave's list -> the global per-query lists (only this wave reads or writes its list: no barrier)
		const uint32_t cnt = min(*my_n, uint32_t(kGlHitCap));
		for (uint32_t e = lane; e < cnt; e += 64) {
			const unsigned long long h = hit_s[e];
			const uint32_t qi = uint32_t(h >> 32);
			const uint32_t pos = atomicAdd(&p.cand_cnt[qi], 1u);
			if (pos < p.cap) p.cand_row[size_t(qi) * p.cap + pos] = uint32_t(h);
		}
		if (lane == 0) *my_n = 0;
	};

	// DMA source addressing of this thread: instruction j covers LDS slots [512 j, 512 j + 512) of an operand part
	uint32_t src_r[2], src_c[2];
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		const uint32_t slot = j * kBfThreads + tid;
		src_r[j] = slot >> 2;
		src_c[j] = ((slot & 3) ^ ((src_r[j] >> 2) & 3)) << 3;   // element offset of the chunk inside the row's 32-element stage
	}
	uint64_t iss_tile = blockIdx.x;   // tile / stage of the NEXT stage to issue
	uint32_t iss_stage = 0;
	uint64_t iss_g = 0;
	const uint16_t* xsrc[2] = {nullptr, nullptr};
	auto issue = [&]() {
		if (iss_stage == 0) {
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				const uint64_t row = iss_tile * kBfRows + src_r[j];
				xsrc[j] = p.rows + shadow_elem_base((row < p.n ? row : p.n - 1) * p.row_step, p.ld, (p.blocked & 1u) != 0) + src_c[j];   // clamped: discarded by row_ok in the epilogue
			}
		}
		const uint32_t k0 = iss_stage * 32;
		const uint32_t kx = iss_stage * shadow_stage_step((p.blocked & 1u) != 0);
		uint16_t* buf = stage_s + size_t(iss_g % kBufs) * kStageElems;
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			uint16_t* dx = buf + (j * kBfThreads + wave * 64) * 8;                 // wave-uniform base; the DMA adds lane x 16 B
			__builtin_amdgcn_global_load_lds(xsrc[j] + kx, (lds_void*)(dx), 16, 0, 0);
			if (j < kQDma) __builtin_amdgcn_global_load_lds(p.queries + size_t(src_r[j]) * p.ld + src_c[j] + k0, (lds_void*)(dx + kGlXElems), 16, 0, 0);
		}
		++iss_g;
		if (++iss_stage == stages) {
			iss_stage = 0;
			iss_tile += gridDim.x;
		}
	};
	for (int a = 0; a < kAhead; ++a) {
		if (iss_g < total) issue();
	}

	// fragment addressing: row R of the part, chunk c -> slot 4 R + (c ^ ((R >> 2) & 3)); both rows (lane & 31) + 32 a keep (R >> 2) & 3 = (lane >> 2) & 3
	const uint32_t half = lane >> 5;
	const uint32_t swz = (lane >> 2) & 3;
	const uint32_t xrow = 64 * rp + (lane & 31), qrow = (QT / 2) * qh + (lane & 31);

	uint64_t tile = blockIdx.x;
	uint32_t s = 0;
	f32x16 acc[2][QB];
#pragma unroll
	for (int a = 0; a < 2; ++a) {
#pragma unroll
		for (int b = 0; b < QB; ++b) {
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
		}
	}
	for (uint64_t g = 0; g < total; ++g) {
		// stage g has landed for THIS wave once at most the younger stages' DMAs (kIps per stage) are outstanding
		const uint64_t younger = iss_g - g - 1;
		switch (younger < uint64_t(kAhead - 1) ? int(younger) : kAhead - 1) {
			case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
			case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kIps) : "memory"); break;
			case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kIps) : "memory"); break;
			case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * kIps) : "memory"); break;
			default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * kIps) : "memory"); break;
		}
		__builtin_amdgcn_s_barrier();   // ... and for every wave; also: everyone is done reading the buffer the next DMA overwrites
		asm volatile("" ::: "memory");
		if (iss_g < total) issue();
		const uint16_t* buf = stage_s + size_t(g % kBufs) * kStageElems;
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			const uint32_t cs = ((2 * t + half) ^ swz) << 3;
			bf16x8 bfrag[2], afrag[QB];
#pragma unroll
			for (int a = 0; a < 2; ++a) bfrag[a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(buf + (xrow + 32 * a) * 32 + cs));
#pragma unroll
			for (int b = 0; b < QB; ++b) afrag[b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(buf + kGlXElems + (qrow + 32 * b) * 32 + cs));
#pragma unroll
			for (int a = 0; a < 2; ++a) {
#pragma unroll
				for (int b = 0; b < QB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[b], bfrag[a], acc[a][b], 0, 0, 0);
			}
		}
		if (++s < stages) continue;
		s = 0;
		// ---- tile epilogue (same element mapping as the register-staged kernel)
		const uint64_t row0 = tile * kBfRows;
		tile += gridDim.x;
#pragma unroll
		for (int a = 0; a < 2; ++a) {
			const uint64_t row = row0 + 64 * rp + 32 * a + (lane & 31);
			const bool row_ok = row < p.n;
			const uint64_t rowc = (row_ok ? row : p.n - 1) * p.row_step;
			float row_term = 0.f;
			if constexpr (kMetric == kL2) row_term = p.row_sq[rowc];
			if constexpr (kMetric == kCos) row_term = p.inv_norms[rowc];
			const int qlane = 4 * (lane >> 5) + (QT / 2) * qh;
			if constexpr (kMode == kGemmDense) {
				float* dp = p.dense + size_t(qlane) * p.n + row;
				const size_t n1 = p.n, n5 = 5 * p.n;
#pragma unroll
				for (int b = 0; b < QB; ++b) {
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						const int qo = 32 * b + (r & 3) + 8 * (r >> 2);
						float d;
						if constexpr (kMetric == kL2) {
							d = (aux_s[qo + qlane] + row_term) - 2.0f * acc[a][b][r];
						} else if constexpr (kMetric == kIP) {
							d = -acc[a][b][r];
						} else {
							d = -acc[a][b][r] * row_term;
						}
						if (row_ok) *dp = d;
						dp += ((r & 3) == 3) ? n5 : n1;
						asm volatile("" : "+v"(dp));
						acc[a][b][r] = 0.0f;
					}
				}
			} else {
#pragma unroll
				for (int b = 0; b < QB; ++b) {
					// Block-level test first: the best of the lane's 16 products against the loosest of their 16 thresholds, by the same
					// formula (every step of it is monotone in the product, rounding included, so no nomination can hide behind it); only
					// when some lane of the wave passes do the 16 compares and their 16 (L2: 32) LDS reads run.
					float best = acc[a][b][0];
#pragma unroll
					for (int r = 1; r < 16; ++r) best = fmaxf(best, acc[a][b][r]);
					const int gi = (qh * QB + b) * 2 + (lane >> 5);
					float dbest;
					if constexpr (kMetric == kL2) {
						dbest = (gmin_s[gi] + row_term) - 2.0f * best;
					} else if constexpr (kMetric == kIP) {
						dbest = -best;
					} else {
						dbest = -best * row_term;
					}
					uint32_t mask = 0;
					if (__ballot(row_ok && !(dbest > gmax_s[gi]))) {
#pragma unroll
						for (int r = 0; r < 16; ++r) {
							const int qo = 32 * b + (r & 3) + 8 * (r >> 2);
							float d;
							if constexpr (kMetric == kL2) {
								d = (aux_s[qo + qlane] + row_term) - 2.0f * acc[a][b][r];
							} else if constexpr (kMetric == kIP) {
								d = -acc[a][b][r];
							} else {
								d = -acc[a][b][r] * row_term;
							}
							mask |= (d <= thr_s[qo + qlane]) ? (1u << r) : 0u;
						}
					}
#pragma unroll
					for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
					if (!row_ok) mask = 0;
					// block after block: hoisting the next block's threshold / query-norm reads above this block's compares only adds live
					// registers (the L2 form, two LDS operands per compare, spilled 61 VGPRs into the K loop without this)
					__builtin_amdgcn_sched_barrier(0);
					if (__ballot(mask != 0)) {
						while (mask) {
							const int r = __builtin_ctz(mask);
							mask &= mask - 1;
							const uint32_t qi = 32 * b + (r & 3) + 8 * (r >> 2) + qlane;
							const uint32_t at = atomicAdd(my_n, 1u);
							if (at < uint32_t(kGlHitCap)) {
								hit_s[at] = (static_cast<unsigned long long>(qi) << 32) | uint32_t(row);
							} else {   // list full (rows next to many queries): straight to the global lists
								const uint32_t pos = atomicAdd(&p.cand_cnt[qi], 1u);
								if (pos < p.cap) p.cand_row[size_t(qi) * p.cap + pos] = uint32_t(row);
							}
						}
					}
				}
			}
		}
		if constexpr (kMode == kGemmFilter) {
			if (*my_n >= uint32_t(kGlHitCap / 2)) flush_hits();   // wave-uniform: only this wave writes its counter
		}
	}
	if constexpr (kMode == kGemmFilter) flush_hits();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// split-ring variant.  What bounds the kernel above is the number of HBM bytes a CU has in flight: a stage is 16 KB of rows (HBM, ~2-3 us
// under load) + 16 KB of queries (the 393 KB query block sits in L2), both in one ring of four 32 KB buffers -> three stages = 48 KB of
// rows in flight per CU, 12 MB over the chip, against the ~16-24 MB that 8 TB/s x 2-3 us ask for (the 128-query tile, with five 24 KB
// stages in flight, streams the shadow at 4.5 TB/s; this one at 3.2).  The two operands cannot simply get rings of different depth:
// vmcnt retires in order, so a wave that waits for a query stage issued two iterations ago also waits for every row stage it issued
// before that.  Hence the loaders are split by WAVE: waves 0-3 issue only row stages into a ring of RB buffers (RB - 1 in flight), waves
// 4-7 only query stages into a ring of three (two in flight); each wave waits for its own stream, the per-stage barrier publishes both.
// All eight waves multiply as before.  The per-row terms of the epilogue (|x|^2, 1/|x|) travel by LDS-DMA in front of their tile's
// first row stage: a global load in the epilogue would wait for every DMA in flight.
constexpr int gl_row_bufs(int qt) { return qt == 256 ? 6 : 7; }   // (seven row + two query buffers for the 256-query tile measured no better: 4.88 ms)
constexpr int gl_query_bufs(int) { return 3; }
constexpr int kGlHitCapSplit = 144;
// Row terms of the tiles whose first row stage has been issued: the row loaders run RB - 1 stages ahead, i.e. up to
// 1 + (RB - 2) / stages tiles beyond the one in its epilogue (stages >= 2: ld is a multiple of 64) -> at most 4 tiles with RB <= 7.
// (A ring of two was overrun at ld = 64 / 128, where the loaders are 2-3 tiles ahead: ADVICE round 4.)
constexpr int kGlTermRing = 4;

template <int kMetric, int kMode, int QT, int RB = gl_row_bufs(QT), int QBUFS = gl_query_bufs(QT)>
__global__ __launch_bounds__(kBfThreads) void knn_gemm_bf16_split(GemmBf16Params p) {
	static_assert(RB >= 3 && RB <= 7 && QBUFS >= 2 && QBUFS <= 3, "the vmcnt switches below cover these ring depths");
	static_assert(2 + (RB - 2) / 2 <= kGlTermRing, "term ring shorter than the row loaders' lead in tiles at two stages per tile");
	constexpr int kQElems = QT * 32;                 // the query operand of one stage
	constexpr int QB = QT / 64;                      // 32-query blocks per wave (two query halves)
	constexpr int kQIps = QT * 4 / 256;              // DMA instructions per query-loader wave and stage (4 or 2)
	constexpr bool kTerms = kMetric != kIP;          // the epilogue needs one float per row
	extern __shared__ __attribute__((aligned(16))) unsigned char bf_lds[];   // the ONLY shared object (a second one de-pipelines the DMA waits)
	uint16_t* rows_s = reinterpret_cast<uint16_t*>(bf_lds);                                    // [RB][256 x 32]
	uint16_t* qry_s = rows_s + size_t(RB) * kGlXElems;                                         // [QBUFS][QT x 32]
	float* term_s = reinterpret_cast<float*>(qry_s + size_t(QBUFS) * kQElems);                 // [kGlTermRing][256] row terms of the tiles in flight
	float* thr_s = term_s + kGlTermRing * kBfRows;                                                       // [QT]
	float* aux_s = thr_s + QT;
	uint32_t* hit_n = reinterpret_cast<uint32_t*>(aux_s + QT);                                 // [8]: one counter per wavefront
	unsigned long long* hit_all = reinterpret_cast<unsigned long long*>(hit_n + 8);            // [8][kGlHitCapSplit]
	float* gmax_s = reinterpret_cast<float*>(hit_all + size_t(8) * kGlHitCapSplit);            // [QT / 16] loosest threshold of a lane's 16 queries
	float* gmin_s = gmax_s + QT / 16;                                                          // [QT / 16] smallest |q|^2 of them (L2)

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int rp = wave & 3, qh = wave >> 2;
	const bool row_loader = wave < 4;                // wave-uniform role in the DMA streams
	const uint32_t stages = p.ld / 32;
	const uint64_t ntiles = (p.n + kBfRows - 1) / kBfRows;
	const uint64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
	const uint64_t total = my_tiles * stages;
	for (int i = tid; i < QT; i += kBfThreads) {
		thr_s[i] = kMode == kGemmFilter ? p.thr[i] : 0.f;
		aux_s[i] = kMetric == kL2 ? p.q_sq[i] : 0.f;
	}
	if (tid < 8) hit_n[tid] = 0;
	__syncthreads();
	if (kMode == kGemmFilter && tid < QT / 16) {   // block-level test of the filter epilogue, as in the kernel above
		const int half = tid & 1, bb = (tid >> 1) % QB, hh = (tid >> 1) / QB;
		float tmax = -__builtin_inff(), amin = __builtin_inff();
		for (int r = 0; r < 16; ++r) {
			const int qi = 32 * bb + (r & 3) + 8 * (r >> 2) + 4 * half + (QT / 2) * hh;
			tmax = fmaxf(tmax, thr_s[qi]);
			amin = fminf(amin, aux_s[qi]);
			if (thr_s[qi] != thr_s[qi]) tmax = __builtin_inff();
		}
		gmax_s[tid] = tmax;
		gmin_s[tid] = amin;
	}
	__syncthreads();
	unsigned long long* hit_s = hit_all + size_t(wave) * kGlHitCapSplit;
	uint32_t* my_n = hit_n + wave;
	auto flush_hits = [&]() {
		const uint32_t cnt = min(*my_n, uint32_t(kGlHitCapSplit));
		for (uint32_t e = lane; e < cnt; e += 64) {
			const unsigned long long h = hit_s[e];
			const uint32_t qi = uint32_t(h >> 32);
			const uint32_t pos = atomicAdd(&p.cand_cnt[qi], 1u);
			if (pos < p.cap) p.cand_row[size_t(qi) * p.cap + pos] = uint32_t(h);
		}
		if (lane == 0) *my_n = 0;
	};

	// DMA addressing of this thread inside its 4-wave loader group: instruction j covers LDS slots [256 j, 256 j + 256) of the operand
	// (slot p = 4 r + cs holds chunk c = cs ^ ((r >> 2) & 3) of row r, the swizzle of the kernel above)
	uint32_t src_r[4], src_c[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const uint32_t slot = j * 256 + rp * 64 + lane;
		src_r[j] = slot >> 2;
		src_c[j] = ((slot & 3) ^ ((src_r[j] >> 2) & 3)) << 3;
	}
	const uint16_t* src[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) src[j] = p.queries + size_t(src_r[j] % QT) * p.ld + src_c[j];   // query loaders: fixed; row loaders: set per tile
	uint64_t iss_tile = blockIdx.x;   // tile / stage of the NEXT stage this wave issues
	uint32_t iss_stage = 0;
	uint64_t iss_g = 0;
	uint32_t iss_buf = 0;             // ring position of the next stage (rows: mod RB, queries: mod QBUFS)
	uint32_t iss_par = 0;             // which slot of term_s the next tile's row terms go to (tile counter mod kGlTermRing)
	auto issue = [&]() {
		const uint32_t k0 = iss_stage * (row_loader ? shadow_stage_step((p.blocked & 1u) != 0) : 32u);
		if (row_loader) {
			if (iss_stage == 0) {
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const uint64_t row = iss_tile * kBfRows + src_r[j];
					src[j] = p.rows + shadow_elem_base((row < p.n ? row : p.n - 1) * p.row_step, p.ld, (p.blocked & 1u) != 0) + src_c[j];   // clamped: discarded by row_ok in the epilogue
				}
				if constexpr (kTerms) {   // this wave's 64 row terms, in front of the tile's first rows (in-order retirement: there when they are)
					const uint64_t row = iss_tile * kBfRows + 64 * rp + lane;
					const uint64_t rowc = (row < p.n ? row : p.n - 1) * p.row_step;
					const float* tsrc = (kMetric == kL2 ? p.row_sq : p.inv_norms) + rowc;
					float* tdst = term_s + iss_par * kBfRows + 64 * rp;
					__builtin_amdgcn_global_load_lds(tsrc, (lds_void*)(tdst), 4, 0, 0);
				}
				iss_par = (iss_par + 1u) % uint32_t(kGlTermRing);
			}
			uint16_t* buf = rows_s + size_t(iss_buf) * kGlXElems;
#pragma unroll
			for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(src[j] + k0, (lds_void*)(buf + (j * 256 + rp * 64) * 8), 16, 0, 0);
			iss_buf = iss_buf + 1 == uint32_t(RB) ? 0u : iss_buf + 1;
		} else {
			uint16_t* buf = qry_s + size_t(iss_buf) * kQElems;
#pragma unroll
			for (int j = 0; j < kQIps; ++j) __builtin_amdgcn_global_load_lds(src[j] + k0, (lds_void*)(buf + (j * 256 + rp * 64) * 8), 16, 0, 0);
			iss_buf = iss_buf + 1 == uint32_t(QBUFS) ? 0u : iss_buf + 1;
		}
		++iss_g;
		if (++iss_stage == stages) {
			iss_stage = 0;
			iss_tile += gridDim.x;
		}
	};
	const int ahead = row_loader ? RB - 1 : QBUFS - 1;
	for (int a = 0; a < ahead; ++a) {
		if (iss_g < total) issue();
	}

	const uint32_t half = lane >> 5;
	const uint32_t swz = (lane >> 2) & 3;
	const uint32_t xrow = 64 * rp + (lane & 31), qrow = (QT / 2) * qh + (lane & 31);

	uint64_t tile = blockIdx.x;
	uint32_t s = 0, rbuf = 0, qbuf = 0, tpar = 0;
	f32x16 acc[2][QB];
#pragma unroll
	for (int a = 0; a < 2; ++a) {
#pragma unroll
		for (int b = 0; b < QB; ++b) {
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
		}
	}
	for (uint64_t g = 0; g < total; ++g) {
		// stage g of THIS wave's stream has landed once at most the younger stages' DMAs are outstanding (4 resp. kQIps per stage; a tile's
		// row terms ride in front of its first stage, so counting them out only ever waits for one instruction more than needed)
		const uint64_t younger = iss_g - g - 1;
		if (row_loader) {
			switch (younger < uint64_t(RB - 2) ? int(younger) : RB - 2) {
				case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
				case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
				case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
				case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
				case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
				default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
			}
		} else {
			switch (younger < uint64_t(QBUFS - 2) ? int(younger) : QBUFS - 2) {
				case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
				default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kQIps) : "memory"); break;
			}
		}
		__builtin_amdgcn_s_barrier();   // ... and for every wave; also: everyone is done reading the buffers the next DMAs overwrite
		asm volatile("" ::: "memory");
		if (iss_g < total) issue();
		const uint16_t* xb = rows_s + size_t(rbuf) * kGlXElems;
		const uint16_t* qb = qry_s + size_t(qbuf) * kQElems;
		rbuf = rbuf + 1 == uint32_t(RB) ? 0u : rbuf + 1;
		qbuf = qbuf + 1 == uint32_t(QBUFS) ? 0u : qbuf + 1;
		if (p.blocked & 2u) __builtin_amdgcn_s_setprio(1);   // RXGPU_GEMM_PRIO=1 (A/B): the stage's fragment reads and MFMAs ahead of the other wave's epilogue / DMA issue
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			const uint32_t cs = ((2 * t + half) ^ swz) << 3;
			bf16x8 bfrag[2], afrag[QB];
#pragma unroll
			for (int a = 0; a < 2; ++a) bfrag[a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xb + (xrow + 32 * a) * 32 + cs));
#pragma unroll
			for (int b = 0; b < QB; ++b) afrag[b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qb + (qrow + 32 * b) * 32 + cs));
#pragma unroll
			for (int a = 0; a < 2; ++a) {
#pragma unroll
				for (int b = 0; b < QB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[b], bfrag[a], acc[a][b], 0, 0, 0);
			}
		}
		if (p.blocked & 2u) __builtin_amdgcn_s_setprio(0);
		if (++s < stages) continue;
		s = 0;
		// ---- tile epilogue (same element mapping as the kernel above; row terms from LDS)
		const uint64_t row0 = tile * kBfRows;
		tile += gridDim.x;
		const float* terms = term_s + tpar * kBfRows;
		tpar = (tpar + 1u) % uint32_t(kGlTermRing);
#pragma unroll
		for (int a = 0; a < 2; ++a) {
			const uint64_t row = row0 + 64 * rp + 32 * a + (lane & 31);
			const bool row_ok = row < p.n;
			float row_term = 0.f;
			if constexpr (kTerms) row_term = terms[64 * rp + 32 * a + (lane & 31)];
			const int qlane = 4 * (lane >> 5) + (QT / 2) * qh;
			if constexpr (kMode == kGemmDense) {
				float* dp = p.dense + size_t(qlane) * p.n + row;
				const size_t n1 = p.n, n5 = 5 * p.n;
#pragma unroll
				for (int b = 0; b < QB; ++b) {
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						const int qo = 32 * b + (r & 3) + 8 * (r >> 2);
						float d;
						if constexpr (kMetric == kL2) {
							d = (aux_s[qo + qlane] + row_term) - 2.0f * acc[a][b][r];
						} else if constexpr (kMetric == kIP) {
							d = -acc[a][b][r];
						} else {
							d = -acc[a][b][r] * row_term;
						}
						if (row_ok) *dp = d;
						dp += ((r & 3) == 3) ? n5 : n1;
						asm volatile("" : "+v"(dp));
						acc[a][b][r] = 0.0f;
					}
				}
			} else {
#pragma unroll
				for (int b = 0; b < QB; ++b) {
					float best = acc[a][b][0];
#pragma unroll
					for (int r = 1; r < 16; ++r) best = fmaxf(best, acc[a][b][r]);
					const int gi = (qh * QB + b) * 2 + (lane >> 5);
					float dbest;
					if constexpr (kMetric == kL2) {
						dbest = (gmin_s[gi] + row_term) - 2.0f * best;
					} else if constexpr (kMetric == kIP) {
						dbest = -best;
					} else {
						dbest = -best * row_term;
					}
					uint32_t mask = 0;
					if (__ballot(row_ok && !(dbest > gmax_s[gi]))) {
#pragma unroll
						for (int r = 0; r < 16; ++r) {
							const int qo = 32 * b + (r & 3) + 8 * (r >> 2);
							float d;
							if constexpr (kMetric == kL2) {
								d = (aux_s[qo + qlane] + row_term) - 2.0f * acc[a][b][r];
							} else if constexpr (kMetric == kIP) {
								d = -acc[a][b][r];
							} else {
								d = -acc[a][b][r] * row_term;
							}
							mask |= (d <= thr_s[qo + qlane]) ? (1u << r) : 0u;
						}
					}
#pragma unroll
					for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
					if (!row_ok) mask = 0;
					__builtin_amdgcn_sched_barrier(0);
					if (__ballot(mask != 0)) {
						while (mask) {
							const int r = __builtin_ctz(mask);
							mask &= mask - 1;
							const uint32_t qi = 32 * b + (r & 3) + 8 * (r >> 2) + qlane;
							const uint32_t at = atomicAdd(my_n, 1u);
							if (at < uint32_t(kGlHitCapSplit)) {
								hit_s[at] = (static_cast<unsigned long long>(qi) << 32) | uint32_t(row);
							} else {   // list full (rows next to many queries): straight to the global lists
								const uint32_t pos = atomicAdd(&p.cand_cnt[qi], 1u);
								if (pos < p.cap) p.cand_row[size_t(qi) * p.cap + pos] = uint32_t(row);
							}
						}
					}
				}
			}
		}
		if constexpr (kMode == kGemmFilter) {
			if (*my_n >= uint32_t(kGlHitCapSplit / 2)) flush_hits();   // wave-uniform: only this wave writes its counter
		}
	}
	if constexpr (kMode == kGemmFilter) flush_hits();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// qreg variant (filter pass): the query operand NOT through LDS-DMA.  What bounds the two kernels above is the fill rate of the LDS-DMA
// path: a stage of 32 KB at 256 queries (16 KB rows + 16 KB queries) takes 1.26 us, 24 KB at 128 queries 0.93 us — 25.5 GB/s per CU either
// way (the microarch guide's "ldsdma-fill" figure), with the MFMA pipe busy a third of the time and the LDS a quarter (a software-pipelined
// stage — both halves' fragment reads in front of the first MFMA — changed nothing: 4.57 vs 4.61 ms, profiles/rd5b_gemm_ab.txt).  Half of
// those bytes are the query block, re-streamed from L2 for every tile.  Here the four query-loader waves fetch their stage with plain
// global loads into REGISTERS, two stages ahead, and store it into a two-slot query ring with ds_write_b128 — same LDS image, same
// consumers — so the DMA path carries the rows alone (and the ring gets a seventh row slot for the LDS the third query slot held).
// Measured (tools/bench_gemm_ab.py, 10M x 768, one box, profiles/rd5b_gemm_ab.txt): 256 queries 4.55 -> 4.33 ms (ip), 4.93 -> 4.62 (l2);
// 128 queries 3.4 -> 2.96 ms.  The fill rate was a bound, not the only one: a stage still takes ~1.15 us where its MFMAs need 0.53.
//   * Loads, waits and stores of that path are inline asm: with an LDS-DMA pending the compiler's wait-count pass answers every VGPR
//     dependency on a VMEM load with vmcnt(0) (global_load_lds is FLAT-encoded, "may access LDS"), which would drain the ring each stage.
//   * The registers are in flight ACROSS loop iterations, invisibly to the compiler, so nothing may make it copy them: the two roles run
//     two separate loops (no merge points), the query loop is unrolled by two (stage parity = register set = slot; a tile has an even
//     number of stages since ld is a multiple of 64) and its loads are unconditional (beyond the end they re-read stage 0 — never stored
//     where anyone reads).  tests/test_kernel_resources.py checks the ISA for moves of the staging registers.
template <int kMetric, int QT>
__global__ __launch_bounds__(kBfThreads) void knn_gemm_bf16_qreg(GemmBf16Params p) {
	constexpr int RB = 7;                            // row ring (six stages in flight)
	constexpr int kQElems = QT * 32;                 // the query operand of one stage
	constexpr int QB = QT / 64;                      // 32-query blocks per wave (two query halves)
	constexpr int kQIps = QT * 4 / 256;              // 16-byte pieces per query-loader lane and stage (4 or 2)
	constexpr bool kTerms = kMetric != kIP;          // the epilogue needs one float per row
	extern __shared__ __attribute__((aligned(16))) unsigned char bf_lds[];   // the ONLY shared object
	uint16_t* rows_s = reinterpret_cast<uint16_t*>(bf_lds);                                    // [RB][256 x 32]
	uint16_t* qry_s = rows_s + size_t(RB) * kGlXElems;                                         // [2][QT x 32]
	float* term_s = reinterpret_cast<float*>(qry_s + size_t(2) * kQElems);                     // [kGlTermRing][256]
	float* thr_s = term_s + kGlTermRing * kBfRows;                                             // [QT]
	float* aux_s = thr_s + QT;
	uint32_t* hit_n = reinterpret_cast<uint32_t*>(aux_s + QT);                                 // [8]
	unsigned long long* hit_all = reinterpret_cast<unsigned long long*>(hit_n + 8);            // [8][kGlHitCapSplit]
	float* gmax_s = reinterpret_cast<float*>(hit_all + size_t(8) * kGlHitCapSplit);            // [QT / 16]
	float* gmin_s = gmax_s + QT / 16;                                                          // [QT / 16]

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int rp = wave & 3, qh = wave >> 2;
	const bool row_loader = wave < 4;
	const uint32_t stages = p.ld / 32;
	const uint64_t ntiles = (p.n + kBfRows - 1) / kBfRows;
	const uint64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
	const uint64_t total = my_tiles * stages;   // even
	for (int i = tid; i < QT; i += kBfThreads) {
		thr_s[i] = p.thr[i];
		aux_s[i] = kMetric == kL2 ? p.q_sq[i] : 0.f;
	}
	if (tid < 8) hit_n[tid] = 0;
	__syncthreads();
	if (tid < QT / 16) {   // block-level test of the filter epilogue, as in the kernels above
		const int half = tid & 1, bb = (tid >> 1) % QB, hh = (tid >> 1) / QB;
		float tmax = -__builtin_inff(), amin = __builtin_inff();
		for (int r = 0; r < 16; ++r) {
			const int qi = 32 * bb + (r & 3) + 8 * (r >> 2) + 4 * half + (QT / 2) * hh;
			tmax = fmaxf(tmax, thr_s[qi]);
			amin = fminf(amin, aux_s[qi]);
			if (thr_s[qi] != thr_s[qi]) tmax = __builtin_inff();
		}
		gmax_s[tid] = tmax;
		gmin_s[tid] = amin;
	}
	__syncthreads();
	if (total == 0) return;
	unsigned long long* hit_s = hit_all + size_t(wave) * kGlHitCapSplit;
	uint32_t* my_n = hit_n + wave;
	auto flush_hits = [&]() {
		const uint32_t cnt = min(*my_n, uint32_t(kGlHitCapSplit));
		for (uint32_t e = lane; e < cnt; e += 64) {
			const unsigned long long h = hit_s[e];
			const uint32_t qi = uint32_t(h >> 32);
			const uint32_t pos = atomicAdd(&p.cand_cnt[qi], 1u);
			if (pos < p.cap) p.cand_row[size_t(qi) * p.cap + pos] = uint32_t(h);
		}
		if (lane == 0) *my_n = 0;
	};
	// piece j of a loader lane covers LDS slots [256 j, 256 j + 256) of the operand: slot p = 4 r + cs holds chunk c = cs ^ ((r >> 2) & 3) of row r
	uint32_t src_r[4], src_c[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const uint32_t slot = j * 256 + rp * 64 + lane;
		src_r[j] = slot >> 2;
		src_c[j] = ((slot & 3) ^ ((src_r[j] >> 2) & 3)) << 3;
	}
	const uint32_t half = lane >> 5;
	const uint32_t swz = (lane >> 2) & 3;
	const uint32_t xrow = 64 * rp + (lane & 31), qrow = (QT / 2) * qh + (lane & 31);
	f32x16 acc[2][QB];
#pragma unroll
	for (int a = 0; a < 2; ++a) {
#pragma unroll
		for (int b = 0; b < QB; ++b) {
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
		}
	}
	auto multiply = [&](const uint16_t* xb, const uint16_t* qb) {
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			const uint32_t cs = ((2 * t + half) ^ swz) << 3;
			bf16x8 bfrag[2], afrag[QB];
#pragma unroll
			for (int a = 0; a < 2; ++a) bfrag[a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xb + (xrow + 32 * a) * 32 + cs));
#pragma unroll
			for (int b = 0; b < QB; ++b) afrag[b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qb + (qrow + 32 * b) * 32 + cs));
#pragma unroll
			for (int a = 0; a < 2; ++a) {
#pragma unroll
				for (int b = 0; b < QB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[b], bfrag[a], acc[a][b], 0, 0, 0);
			}
		}
	};
	// the filter epilogue of one tile (element mapping and block test of the kernels above; row terms from LDS)
	auto tile_end = [&](const uint64_t row0, const float* terms) {
#pragma unroll
		for (int a = 0; a < 2; ++a) {
			const uint64_t row = row0 + 64 * rp + 32 * a + (lane & 31);
			const bool row_ok = row < p.n;
			float row_term = 0.f;
			if constexpr (kTerms) row_term = terms[64 * rp + 32 * a + (lane & 31)];
			const int qlane = 4 * (lane >> 5) + (QT / 2) * qh;
#pragma unroll
			for (int b = 0; b < QB; ++b) {
				float best = acc[a][b][0];
#pragma unroll
				for (int r = 1; r < 16; ++r) best = fmaxf(best, acc[a][b][r]);
				const int gi = (qh * QB + b) * 2 + (lane >> 5);
				float dbest;
				if constexpr (kMetric == kL2) {
					dbest = (gmin_s[gi] + row_term) - 2.0f * best;
				} else if constexpr (kMetric == kIP) {
					dbest = -best;
				} else {
					dbest = -best * row_term;
				}
				uint32_t mask = 0;
				if (__ballot(row_ok && !(dbest > gmax_s[gi]))) {
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						const int qo = 32 * b + (r & 3) + 8 * (r >> 2);
						float d;
						if constexpr (kMetric == kL2) {
							d = (aux_s[qo + qlane] + row_term) - 2.0f * acc[a][b][r];
						} else if constexpr (kMetric == kIP) {
							d = -acc[a][b][r];
						} else {
							d = -acc[a][b][r] * row_term;
						}
						mask |= (d <= thr_s[qo + qlane]) ? (1u << r) : 0u;
					}
				}
#pragma unroll
				for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
				if (!row_ok) mask = 0;
				__builtin_amdgcn_sched_barrier(0);
				if (__ballot(mask != 0)) {
					while (mask) {
						const int r = __builtin_ctz(mask);
						mask &= mask - 1;
						const uint32_t qi = 32 * b + (r & 3) + 8 * (r >> 2) + qlane;
						const uint32_t at = atomicAdd(my_n, 1u);
						if (at < uint32_t(kGlHitCapSplit)) {
							hit_s[at] = (static_cast<unsigned long long>(qi) << 32) | uint32_t(row);
						} else {
							const uint32_t pos = atomicAdd(&p.cand_cnt[qi], 1u);
							if (pos < p.cap) p.cand_row[size_t(qi) * p.cap + pos] = uint32_t(row);
						}
					}
				}
			}
		}
		if (*my_n >= uint32_t(kGlHitCapSplit / 2)) flush_hits();
	};

	uint64_t tile = blockIdx.x;
	uint32_t s = 0, rbuf = 0, tpar = 0;
	if (row_loader) {
		// ---- waves 0-3: the row stream by LDS-DMA, RB - 1 stages ahead (as in the split-ring kernel)
		const uint16_t* src[4] = {nullptr, nullptr, nullptr, nullptr};
		uint64_t iss_tile = blockIdx.x, iss_g = 0;
		uint32_t iss_stage = 0, iss_buf = 0, iss_par = 0;
		auto issue = [&]() {
			const uint32_t k0 = iss_stage * shadow_stage_step((p.blocked & 1u) != 0);
			if (iss_stage == 0) {
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const uint64_t row = iss_tile * kBfRows + src_r[j];
					src[j] = p.rows + shadow_elem_base((row < p.n ? row : p.n - 1) * p.row_step, p.ld, (p.blocked & 1u) != 0) + src_c[j];   // clamped: discarded by row_ok
				}
				if constexpr (kTerms) {
					const uint64_t row = iss_tile * kBfRows + 64 * rp + lane;
					const uint64_t rowc = (row < p.n ? row : p.n - 1) * p.row_step;
					const float* tsrc = (kMetric == kL2 ? p.row_sq : p.inv_norms) + rowc;
					float* tdst = term_s + iss_par * kBfRows + 64 * rp;
					__builtin_amdgcn_global_load_lds(tsrc, (lds_void*)(tdst), 4, 0, 0);
				}
				iss_par = (iss_par + 1u) % uint32_t(kGlTermRing);
			}
			uint16_t* buf = rows_s + size_t(iss_buf) * kGlXElems;
#pragma unroll
			for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(src[j] + k0, (lds_void*)(buf + (j * 256 + rp * 64) * 8), 16, 0, 0);
			iss_buf = iss_buf + 1 == uint32_t(RB) ? 0u : iss_buf + 1;
			++iss_g;
			if (++iss_stage == stages) {
				iss_stage = 0;
				iss_tile += gridDim.x;
			}
		};
		for (int a = 0; a < RB - 1; ++a) {
			if (iss_g < total) issue();
		}
		for (uint64_t g = 0; g < total; ++g) {
			const uint64_t younger = iss_g - g - 1;
			switch (younger < uint64_t(RB - 2) ? int(younger) : RB - 2) {
				case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
				case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
				case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
				case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
				case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
				default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
			}
			__builtin_amdgcn_s_barrier();
			asm volatile("" ::: "memory");
			if (iss_g < total) issue();
			multiply(rows_s + size_t(rbuf) * kGlXElems, qry_s + size_t(g & 1) * kQElems);
			rbuf = rbuf + 1 == uint32_t(RB) ? 0u : rbuf + 1;
			if (++s < stages) continue;
			s = 0;
			tile_end(tile * kBfRows, term_s + tpar * kBfRows);
			tile += gridDim.x;
			tpar = (tpar + 1u) % uint32_t(kGlTermRing);
		}
	} else {
		// ---- waves 4-7: the query stream through registers, two stages ahead; stage g -> register set g & 1 -> ring slot g & 1
		const uint16_t* qsrc[kQIps];
#pragma unroll
		for (int j = 0; j < kQIps; ++j) qsrc[j] = p.queries + size_t(src_r[j] % QT) * p.ld + src_c[j];
		const uint32_t st_base = uint32_t(size_t((lds_void*)(qry_s + (rp * 64 + lane) * 8)));
		u32x4 qs0[kQIps], qs1[kQIps];
		uint32_t qls = 0;   // stage (inside its tile) of the next load
#define RX_Q_LOAD(SET)                                                                                                             \
	do {                                                                                                                           \
		const uint32_t k0__ = qls * 32u;                                                                                           \
		_Pragma("unroll") for (int j = 0; j < kQIps; ++j)                                                                         \
			asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(SET[j]) : "v"(qsrc[j] + k0__) : "memory");                       \
		qls = qls + 1 == stages ? 0u : qls + 1;                                                                                    \
	} while (0)
#define RX_Q_STORE(SET, SLOT)                                                                                                      \
	do {                                                                                                                           \
		_Pragma("unroll") for (int j = 0; j < kQIps; ++j)                                                                         \
			asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(st_base), "v"(SET[j]), "n"((SLOT) * kQElems * 2 + j * 256 * 16) : "memory"); \
	} while (0)
		// (Order inside a stage, measured at 10M x 768, 256 queries, ip: this wave loading FIRST and multiplying after — like its partner on the
		// SIMD, the row loader, which issues its DMA pieces first — 4.33 ms; multiplying first and moving its query stage behind the MFMAs, so
		// that one of the pair computes while the other loads, 4.52 ms: the shorter lead of the loads costs more than the overlap brings.)
		RX_Q_LOAD(qs0);   // stage 0
		RX_Q_LOAD(qs1);   // stage 1 (total >= 2)
		asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kQIps) : "memory");
		RX_Q_STORE(qs0, 0);
		for (uint64_t g = 0; g < total; g += 2) {
			// ---- even stage g: set 0 is free (stage g sits in slot 0), stage g + 1 waits in set 1
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's stores of stage g are in the LDS
			__builtin_amdgcn_s_barrier();
			asm volatile("" ::: "memory");
			RX_Q_LOAD(qs0);                                                   // stage g + 2 (beyond the end: a stage nobody reads)
			asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kQIps) : "memory");      // ... stage g + 1 has landed (the older half of the queue)
			RX_Q_STORE(qs1, 1);
			multiply(rows_s + size_t(rbuf) * kGlXElems, qry_s);
			rbuf = rbuf + 1 == uint32_t(RB) ? 0u : rbuf + 1;
			// ---- odd stage g + 1
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			asm volatile("" ::: "memory");
			RX_Q_LOAD(qs1);                                                   // stage g + 3
			asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kQIps) : "memory");      // stage g + 2 has landed
			RX_Q_STORE(qs0, 0);
			multiply(rows_s + size_t(rbuf) * kGlXElems, qry_s + kQElems);
			rbuf = rbuf + 1 == uint32_t(RB) ? 0u : rbuf + 1;
			s += 2;
			if (s < stages) continue;
			s = 0;
			tile_end(tile * kBfRows, term_s + tpar * kBfRows);
			tile += gridDim.x;
			tpar = (tpar + 1u) % uint32_t(kGlTermRing);
		}
#undef RX_Q_LOAD
#undef RX_Q_STORE
	}
	flush_hits();
}

size_t gemm_bf16_split_lds_bytes(int qt, int rb, int qbufs) {
	return (size_t(rb) * kGlXElems + size_t(qbufs) * qt * 32) * sizeof(uint16_t) + size_t(kGlTermRing) * kBfRows * sizeof(float) +
		   2 * size_t(qt) * sizeof(float) + 8 * sizeof(uint32_t) + size_t(8) * kGlHitCapSplit * 8 + 2 * size_t(qt / 16) * sizeof(float);
}

size_t gemm_bf16_glds_lds_bytes(int qt) {
	return size_t(gl_bufs(qt)) * (kGlXElems + qt * 32) * sizeof(uint16_t) + 2 * size_t(qt) * sizeof(float) + 8 * sizeof(uint32_t) + size_t(8) * kGlHitCap * 8 +
		   2 * size_t(qt / 16) * sizeof(float);
}

// RXGPU_GEMM_SPLIT=0 / 1: the single-ring kernel / the split-ring one (A/B in one process: read at every launch)
static bool gemm_split_rings() {
	const char* e = std::getenv("RXGPU_GEMM_SPLIT");
	return e ? std::atoi(e) != 0 : true;
}

// RXGPU_GEMM_QREG=0: the query operand through LDS-DMA like the rows (the split-ring kernel; A/B)
static bool gemm_qreg() {
	const char* e = std::getenv("RXGPU_GEMM_QREG");
	return e ? std::atoi(e) != 0 : true;
}

template <int kMetric, int kMode, int QT>
static hipError_t launch_bf16_glds_one(const GemmBf16Params& p, uint32_t grid, hipStream_t s) {
	if constexpr (kMode == kGemmFilter) {   // (the dense form runs over the 32 K-row sample only)
		if (gemm_split_rings() && gemm_qreg()) {
			const size_t lds = gemm_bf16_split_lds_bytes(QT, 7, 2);
			static std::atomic<uint64_t> raised_qreg{0};
			if (hipError_t e = raise_dynamic_lds_once(raised_qreg, reinterpret_cast<const void*>(&knn_gemm_bf16_qreg<kMetric, QT>), lds); e != hipSuccess) return e;
			hipLaunchKernelGGL((knn_gemm_bf16_qreg<kMetric, QT>), dim3(grid), dim3(kBfThreads), lds, s, p);
			return hipGetLastError();
		}
	}
	if (gemm_split_rings()) {
		const size_t lds = gemm_bf16_split_lds_bytes(QT, gl_row_bufs(QT), gl_query_bufs(QT));
		static std::atomic<uint64_t> raised_split{0};
		if (hipError_t e = raise_dynamic_lds_once(raised_split, reinterpret_cast<const void*>(&knn_gemm_bf16_split<kMetric, kMode, QT>), lds); e != hipSuccess) {
			return e;
		}
		hipLaunchKernelGGL((knn_gemm_bf16_split<kMetric, kMode, QT>), dim3(grid), dim3(kBfThreads), lds, s, p);
		return hipGetLastError();
	}
	const size_t lds = gemm_bf16_glds_lds_bytes(QT);
	static std::atomic<uint64_t> raised{0};
	if (hipError_t e = raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&knn_gemm_bf16_glds<kMetric, kMode, QT>), lds); e != hipSuccess) {
		return e;
	}
	hipLaunchKernelGGL((knn_gemm_bf16_glds<kMetric, kMode, QT>), dim3(grid), dim3(kBfThreads), lds, s, p);
	return hipGetLastError();
}

template <int kMetric, int kMode>
static hipError_t launch_bf16_glds_qt(int qt, const GemmBf16Params& p, uint32_t grid, hipStream_t s) {
	return qt == 128 ? launch_bf16_glds_one<kMetric, kMode, 128>(p, grid, s) : launch_bf16_glds_one<kMetric, kMode, 256>(p, grid, s);
}

// qt: query-tile width, 128 or 256 (the query block and the threshold arrays hold qt rows)
hipError_t launch_gemm_bf16(int metric, int mode, int qt, const GemmBf16Params& p, uint32_t grid, hipStream_t s) {
	if (mode == kGemmDense) {
		switch (metric) {
			case kL2: return launch_bf16_glds_qt<kL2, kGemmDense>(qt, p, grid, s);
			case kIP: return launch_bf16_glds_qt<kIP, kGemmDense>(qt, p, grid, s);
			default: return launch_bf16_glds_qt<kCos, kGemmDense>(qt, p, grid, s);
		}
	}
	switch (metric) {
		case kL2: return launch_bf16_glds_qt<kL2, kGemmFilter>(qt, p, grid, s);
		case kIP: return launch_bf16_glds_qt<kIP, kGemmFilter>(qt, p, grid, s);
		default: return launch_bf16_glds_qt<kCos, kGemmFilter>(qt, p, grid, s);
	}
}

void launch_to_bf16(const float* src, uint64_t n, uint32_t stride, uint32_t dim, uint16_t* dst, uint32_t ld, int cus, hipStream_t s, uint64_t first_row, bool blocked) {
	if (!n) return;
	const uint64_t total = n * (ld / 8);
	const uint32_t blocks = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((total + 255) / 256, uint64_t(cus) * 16)));
	hipLaunchKernelGGL(knn_to_bf16, dim3(blocks), dim3(256), 0, s, src, n, stride, dim, dst, ld, first_row, blocked ? 1u : 0u);
}

void launch_shadow_move(uint16_t* shadow, uint32_t ld, uint64_t from, uint64_t to, bool blocked, hipStream_t s) {
	hipLaunchKernelGGL(knn_shadow_move, dim3(1), dim3(256), 0, s, shadow, ld, from, to, blocked ? 1u : 0u);
}

}  // namespace rxgpu
