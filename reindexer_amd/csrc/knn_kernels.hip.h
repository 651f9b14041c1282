// Brute-force float_vector KNN kernels for gfx950 (MI355X, wave64).
//
// What they replace (reference, CPU): BruteforceSearch::SearchKnn / SearchRange
// (cpp_src/core/index/float_vector/hnswlib/bruteforce.cc:103-143) calling DistCalculator<float>
// (hnswlib.h:147-165) -> L2SqrAVX512 / InnerProductAVX512 (tools/distances/l2_dist.cc:38-72, ip_dist.cc:31-70).
//
// Bit-exactness by construction.  The AVX-512 kernels keep 64 independent fmaf chains, chain L owning the
// elements i == L (mod 64), then fold them with a fixed tree.  Here a row is owned by a 16-lane group of a
// wavefront; lane m of the group loads the float4 at floats [64t+4m, 64t+4m+4) for every 64-float block t, so
// it owns chains 4m..4m+3 completely and in order.  The fold (zmm s0+s1, s2+s3, then _mm512_reduce_add_ps)
// becomes lane-xor 4, 8 (chain index xor 16, 32), lane-xor 2, 1 (chain xor 8, 4) and an in-lane (a0+a2)+(a1+a3).
// IEEE add is commutative, so both partners of a butterfly step hold identical bits.  A wavefront therefore
// scans 4 rows per step with 16-byte loads: 4 fully used 256-byte segments per load instruction.
//
// Roofline: HBM-bound, D*4 algorithmic bytes per row, ~0.5 flop/byte — no MFMA here on purpose.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace rxgpu {

// hipFuncSetAttribute acts on the CURRENT device, and launch helpers are entered from several host threads (sharded indexes run one worker
// per device): the "already raised" flag is one bit per device ordinal, set only after the call succeeded.
inline hipError_t raise_dynamic_lds_once(std::atomic<uint64_t>& done_mask, const void* kernel, size_t bytes) {
	int dev = 0;
	if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
	const uint64_t bit = 1ull << (unsigned(dev) & 63u);
	if (done_mask.load(std::memory_order_acquire) & bit) return hipSuccess;
	if (hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)); e != hipSuccess) return e;
	done_mask.fetch_or(bit, std::memory_order_release);
	return hipSuccess;
}

constexpr int kWave = 64;
constexpr int kGroup = 16;            // lanes per row
constexpr int kRowsPerWave = 4;       // rows per wavefront step
constexpr int kMaxFusedK = 64;        // per-wave register top-k capacity (one (dist,row) pair per lane)
constexpr uint32_t kInvalidRow = 0xFFFFFFFFu;

enum : int { kL2 = 0, kIP = 1, kCos = 2 };

struct ScanParams {
	const float* rows;        // [n][stride]
	const float* inv_norms;   // [n] or nullptr (cosine only)
	const float* queries;     // [nq][dim]
	uint64_t n;
	uint32_t stride;          // floats, multiple of 4
	uint32_t dim;
	uint32_t kk;              // entries kept per list (<= kMaxFusedK)
	float* part_dist;         // [nq][gridDim.x][kk]
	uint32_t* part_row;       // [nq][gridDim.x][kk]
	// optional device-side gate: query blockIdx.y is processed only if gate_cnt[y] > gate_cap (batched-path fallback)
	const uint32_t* gate_cnt;
	uint32_t gate_cap;
};

// bf16-pruned scan (knn_scan.hip: knn_scan_bf16 + knn_filter_approx): approximate distances from the bf16 shadow, exact tail
struct ScanBf16Params {
	ScanParams sp;            // kk, part_dist / part_row ([nq][gridDim.x][kk]); rows / stride / dim / queries unused here
	const uint16_t* rows16;   // bf16 shadow: [n][ld], or tile-blocked (shadow_elem_base)
	uint32_t blocked;         // layout of the shadow
	const float* queries32;   // [nq][ld] f32, zero padded
	const float* row_sq;      // L2
	const float* q_sq;        // L2: [nq]
	uint32_t ld;
	float* approx;            // [nq][n] approximate distance of every row
};

enum : int { kGemmDense = 0, kGemmFilter = 1 };

struct GemmParams {
	const float* rows;        // [n][stride]
	const float* inv_norms;   // cosine
	const float* row_sq;      // L2: |x|^2 per row
	const float* queries;     // [MT][q_stride] zero-padded copy (q_stride % 32 == 0, rows >= nq are zero)
	const float* q_sq;        // L2: |q|^2 per query  [MT]
	uint64_t n;               // rows covered by this launch (sample or whole corpus)
	uint32_t stride, dim, nq, q_stride;
	uint32_t row_step;        // DENSE sample pass: tile row r reads corpus row r * row_step (a strided sample is representative whatever the insertion order); FILTER: 1
	// DENSE
	float* dense;             // [MT][n]
	// FILTER
	const float* thr;         // [MT]
	uint32_t* cand_row;       // [MT][cap]
	uint32_t* cand_cnt;       // [MT]
	uint32_t cap;
};

// bf16 nomination GEMM (knn_batched_bf16.hip): shadow rows / queries as bf16, ld = dim rounded up to 64 (zero padded)
// Layout of the bf16 shadow.  Row-major [n][ld], or TILE-BLOCKED: [tile of 256 rows][k-block of 32 elements][row in tile][32] — one K-stage
// of a nomination tile (256 rows x 32 elements) is then one contiguous 16 KB read instead of 256 pieces of 64 B at a row stride, and a row's
// next k-block lies kShadowStageElems further on.  Element (row, k) = shadow_elem_base(row) + (k / 32) * stage step + k % 32.
constexpr uint32_t kShadowTileRows = 256, kShadowStageElems = kShadowTileRows * 32;
__host__ __device__ inline uint64_t shadow_elem_base(uint64_t row, uint32_t ld, bool blocked) {
	return blocked ? (row / kShadowTileRows) * (uint64_t(kShadowTileRows) * ld) + (row % kShadowTileRows) * 32u : row * ld;
}
__host__ __device__ inline uint32_t shadow_stage_step(bool blocked) { return blocked ? kShadowStageElems : 32u; }

struct GemmBf16Params {
	const uint16_t* rows;     // the shadow (layout: `blocked`)
	uint32_t blocked;
	const uint16_t* queries;  // [256][ld], rows >= nq are zero
	const float* inv_norms;
	const float* row_sq;
	const float* q_sq;        // [256]
	uint64_t n;
	uint32_t ld, nq;
	uint32_t row_step;        // DENSE: tile row r reads corpus row r * row_step; FILTER: 1
	float* dense;             // DENSE: [256][n]
	const float* thr;         // FILTER: [256]
	uint32_t* cand_row;       // [256][cap]
	uint32_t* cand_cnt;
	uint32_t cap;
};

constexpr int kHnswMaxEf = 4096;        // result-heap capacity in LDS (above kHnswLdsCandEf the candidate heap lives in global scratch)
constexpr int kHnswLdsCandEf = 1024;    // largest ef whose candidate heap is tried in LDS first
constexpr int kHnswCandLds = 2048;      // candidate-heap capacity in LDS
constexpr int kHnswMaxNeighbors = 128;  // 2*M <= 128
constexpr uint32_t kHnswOverflow = 0xFFFFFFFFu;
constexpr uint32_t kHnswSpecLog2 = 11;   // speculative team search: 2048-entry direct-mapped table of distances computed ahead of time
constexpr uint32_t kHnswSpecBytes = (2u << kHnswSpecLog2) * 4 + (128 + 3 * 64) * 4;   // table + two link blocks + the trip's id / distance / destination arrays
constexpr int kHnswNblRows = 32;             // team searches: link blocks fetched along with a hop's rows (one 64-word slot each)
constexpr uint32_t kHnswNblBytes = kHnswNblRows * 64 * 4;
constexpr uint32_t kHnswTie = 0xFFFFFFFEu;        // sorted-list search met equal keys: re-run on the heap kernel
constexpr int kHnswSortedMaxEf = 256;            // largest ef the sorted-list search holds in registers (4 entries a lane)
constexpr int kHnswSortedMaxEfDel = 224;         // ... for a graph with deleted nodes: 32 entries of room for the deleted candidates in reach

struct HnswParams {
	const float* rows;
	const float* inv_norms;
	const uint32_t* links0;
	const uint64_t* upper_off;
	const uint32_t* upper;
	const uint8_t* deleted;
	const float* queries;
	uint64_t n;
	uint32_t stride, dim, M, maxM0;
	int maxlevel;
	uint32_t entry;
	int bare;                 // num_deleted == 0 (hnswalg.h:1982)
	uint32_t nq, k, ef;
	uint32_t* visited;        // [slots][visited_words]: a bitset over the nodes zeroed by the launcher, or (vis_hash_log2 > 0) a hash set
	uint64_t visited_words;
	// > 0: the visited set of a search is an open-addressing hash set of 2^vis_hash_log2 words (= visited_words) holding node + 1, zeroed by
	// the search itself — its size follows ef, not the number of nodes.  A search that fills half of it leaves as kHnswOverflow and is re-run
	// on a bitset.  0: one bit per node (N / 8 bytes per search in flight, zeroed by a memset in front of the launch).
	uint32_t vis_hash_log2;
	// Few searches in flight (the latency form of the kernel, at most two workgroups per CU): the hash set lives in the workgroup's LDS
	// behind the heaps — the test-and-set of a hop is an LDS atomic instead of a round trip to L2.  vis_lds_log2 is what the caller allows
	// (0: never); the launcher turns vis_lds on when the launch qualifies and then overrides vis_hash_log2 with it.
	uint32_t vis_lds_log2;
	uint32_t vis_lds;
	uint32_t prefetch_links;     // sorted-list search: fetch the link block of the candidate next in line one hop ahead (LDS-DMA)
	uint32_t team, team_max;     // launches of up to team_max searches run `team` wavefronts per search (hnsw_team_kernel; team <= 1: off)
	uint32_t nbl_off;            // team searches: byte offset of the link-block area (kHnswNblBytes) in the dynamic LDS, 0 = none (set by the launcher)
	uint32_t nbl;                // what the caller allows (RXGPU_HNSW_NBL=1: link blocks come along with a hop's rows; off by default)
	uint32_t spec_off;           // team searches: byte offset of the speculation area (kHnswSpecBytes) in the dynamic LDS, 0 = no speculation (set by the launcher)
	uint32_t spec;               // what the caller allows (RXGPU_HNSW_SPEC=0: off)
	float* out_dist;          // [nq][k]
	uint32_t* out_row;
	uint32_t* out_count;      // [nq]; kHnswOverflow = candidate heap did not fit LDS (re-run in global mode), kHnswTie = re-run on the heaps
	const uint32_t* only;     // optional: list of query indices to process (blockIdx.x indexes this list)
	// overflow queue of the launch (null: none): a search that leaves as kHnswOverflow appends q_base + its query index + 1
	uint32_t* helper_n;
	uint32_t* helper_ids;
	uint32_t helper_cap;
	uint32_t q_base;          // index of this launch's query 0 in the whole batch (chunked launches)
	uint2* gcand;             // global-mode candidate heap storage [slots][gcand_cap] of (dist bits, id)
	uint64_t gcand_cap;
	unsigned long long* stats;   // optional [2]: distance evaluations, hops
	uint32_t lds_cand_cap;       // <= kHnswCandLds (tests shrink it to force the global-heap re-run)
	uint32_t ef_cap;             // result-heap capacity in LDS: ef rounded up to 64
	uint32_t sorted;             // 0 = heap kernel; else the sorted-list search (bare graphs, ef <= kHnswSortedMaxEf; ef_cap = lds_cand_cap = 0)
	// SQ8 graph (HierarchicalNSWImpl<uint8_t>): codes instead of vectors, stored corrective offsets, alpha^2; the queries travel as codes +
	// corrective offset (prepareData, hnswalg.h:510-529) and every distance is scaled by the query's normCoef (queryNormCoef :1855-1863)
	const uint8_t* codes;        // [n][dim]
	const float* corr;           // [n]
	float alpha2;
	const uint8_t* qcodes;       // [nq][dim]
	const float* qcorr;          // [nq]
	const float* qnorm;          // [nq]
};

// what the helper workgroups of a batch poll (hnsw_helper_kernel)
struct HnswHelper {
	const uint32_t* n;        // entries appended so far
	const uint32_t* ids;      // [cap] query index + 1, 0 = not written yet
	const uint32_t* stop;     // set behind the batch's last launch
	const uint32_t* finished; // searches of the batch that have ended (every search workgroup adds one on its way out) ...
	uint32_t expected;        // ... of so many: finished == expected ends the batch for the helpers without anything having to get through a queue
	uint32_t cap;
	unsigned long long ticks_limit;   // wall_clock64 ticks (100 MHz) after which a helper gives up
};

// The resident search kernel of an index (hnsw_server_kernel): workgroup w serves slot w of a mailbox in pinned host memory.  A planner
// thread's single-query SearchKnn is then a store into the mailbox and a poll of it — no launch, no copy, no completion signal on its path.
// Every word has ONE writer: `post` / `req` / the query block / `stop` the host, `done` / `leaving` / the result block the device.
struct HnswServer {
	const uint32_t* post;      // [slots] sequence number of the slot's newest request (host)
	uint32_t* done;            // [slots] sequence number of the slot's newest finished request (device, behind the results)
	const uint32_t* req;       // [slots][2] k, ef of the request
	uint32_t* took;            // [slots] wall-clock ticks (100 MHz) the slot's newest search took on the device, written with its answer (device)
	const uint32_t* stop;      // host word: != 0 -> leave now (the index is about to change)
	uint32_t* leaving;         // host word: this generation's number, written when it decides to leave (stop / idle / lifetime) — the host
	                           // may enqueue the next generation at once: same stream, so it starts when this one is gone
	unsigned long long* dev;   // device words: [0] leave flag, [1] wall clock of the last request taken
	uint32_t generation;
	uint32_t kcap;             // result entries per slot
	unsigned long long idle_ticks, life_ticks;   // wall_clock64 ticks (100 MHz): leave after so long without a request / so long after the start
};

// In-place graph update (rxgpu_hnsw_patch_graph): one workgroup per touched node scatters its staged lists into the resident arrays
struct HnswPatch {
	const uint32_t* ids;          // [n_dirty] node ids
	const uint64_t* upper_at;     // [n_dirty] first upper block of the node in `upper` (new nodes: assigned by the host; old nodes: ~0 = keep)
	const uint32_t* upper_src;    // [n_dirty] first staged upper block of the node in src_upper
	const int32_t* levels;        // [n_dirty] upper levels of the node (= its block count)
	const uint32_t* src_links0;   // [n_dirty][1 + maxM0]
	const uint32_t* src_upper;    // staged upper blocks, (1 + M) words each
	const uint8_t* src_deleted;   // [n_dirty]
	uint32_t* links0;
	uint64_t* upper_off;
	uint32_t* upper;
	uint8_t* deleted;
	uint32_t M, maxM0;
};

// Streaming session (hnsw_stream.hip): Layer0SearchState (hnswalg.h:741-777) resident in device memory between calls
constexpr int kStreamBegin = 0, kStreamContinue = 1, kStreamResume = 2;
constexpr uint32_t kStreamOk = 0, kStreamNeedGlobal = 1, kStreamError = 2;
constexpr int kStreamLdsTop = 1024;    // top_candidates (two arrays: emitStreamingBatch rebuilds the heap)
constexpr int kStreamLdsExt = 1024;    // top_candidates_extras (two arrays: mergeExtrasIntoTopCandidates rebuilds it)
constexpr int kStreamLdsCand = 3072;   // candidate_set
struct HnswStreamState {
	int cand_n, top_n, ext_n;
	float lower;             // lowerBound
	uint32_t top_sel, ext_sel;
	uint32_t status, out_count, exhausted;
};
struct HnswStream {
	const float* query;      // [dim], normalised by the host for cosine
	const uint8_t* qcodes;   // SQ8 graph: the query as codes (prepareData, hnswalg.h:510-529) + its corrective offset and normCoef; else null
	float qcorr, qnorm;
	uint32_t* visited;       // [ceil(n/32)]
	float* cand_d;           // [cap]
	uint32_t* cand_i;
	float* top_d;            // [2][cap]
	uint32_t* top_i;
	float* ext_d;            // [2][cap]
	uint32_t* ext_i;
	uint32_t cap;            // one entry per node is always enough: a node sits in at most one heap
	uint32_t ef;             // StreamingSearchOptions::ef (0 -> 100, hnswalg.h:1867)
	HnswStreamState* state;
	float* out_dist;         // [batch capacity]
	uint32_t* out_row;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-byte row load; kStream = non-temporal (rows are read exactly once per scan — keep them out of L2/MALL's way)
template <bool kStream>
__device__ __forceinline__ float4 load_row4(const float4* p) {
	f32x4 v;
	if constexpr (kStream) {
		v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
	} else {
		v = *reinterpret_cast<const f32x4*>(p);
	}
	return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ bool pair_lt(float d1, uint32_t i1, float d2, uint32_t i2) {
	// std::less<std::pair<float,label>> (priority_queue.h comparator of SearchResultQueue)
	return d1 < d2 || (!(d2 < d1) && i1 < i2);
}

// Sorted (ascending by (dist,row)) list spread over the lanes of one wavefront: lane i holds the i-th best.
struct WaveTopK {
	float bd;
	uint32_t bi;
	float thr_d;       // entry kk-1 once the list is full, +inf before
	uint32_t thr_i;
	uint32_t filled;
	uint32_t kk;

	__device__ __forceinline__ void init(uint32_t kk_) {
		bd = __builtin_inff();
		bi = kInvalidRow;
		thr_d = __builtin_inff();
		thr_i = kInvalidRow;
		filled = 0;
		kk = kk_;
	}
	// (d, idx) must be wave-uniform.
	__device__ __forceinline__ void insert(float d, uint32_t idx, int lane) {
		const bool before = pair_lt(bd, bi, d, idx);
		const uint32_t pos = __popcll(__ballot(before));
		const float up_d = __shfl_up(bd, 1);
		const uint32_t up_i = __shfl_up(bi, 1);
		if (lane == int(pos)) {
			bd = d;
			bi = idx;
		} else if (lane > int(pos)) {
			bd = up_d;
			bi = up_i;
		}
		if (filled < kk) ++filled;
		if (filled == kk) {
			thr_d = __shfl(bd, int(kk) - 1);
			thr_i = __shfl(bi, int(kk) - 1);
		}
	}
	__device__ __forceinline__ bool admits(float d, uint32_t idx) const { return filled < kk || pair_lt(d, idx, thr_d, thr_i); }
};

// Two entries per lane: the fused scan for 64 < kk <= 128 (e.g. hybrid queries with k = 100).  Lane i holds entries i (slot 0) and 64 + i (slot 1)
// of the sorted list; the interface is WaveTopK's, so the scan kernels are templated on the list type and the kk <= 64 code is untouched.
struct WaveTopK2 {
	float d0, d1;
	uint32_t i0, i1;
	float thr_d;
	uint32_t thr_i;
	uint32_t filled;
	uint32_t kk;

	__device__ __forceinline__ void init(uint32_t kk_) {
		d0 = d1 = __builtin_inff();
		i0 = i1 = kInvalidRow;
		thr_d = __builtin_inff();
		thr_i = kInvalidRow;
		filled = 0;
		kk = kk_;
	}
	// (d, idx) must be wave-uniform.
	__device__ __forceinline__ void insert(float d, uint32_t idx, int lane) {
		const uint32_t p0 = __popcll(__ballot(pair_lt(d0, i0, d, idx)));
		const float up1_d = __shfl_up(d1, 1);
		const uint32_t up1_i = __shfl_up(i1, 1);
		if (p0 < 64) {   // lands in slot 0: its last entry carries over to the head of slot 1
			const float carry_d = __shfl(d0, 63);
			const uint32_t carry_i = __shfl(i0, 63);
			const float up_d = __shfl_up(d0, 1);
			const uint32_t up_i = __shfl_up(i0, 1);
			if (lane == int(p0)) {
				d0 = d;
				i0 = idx;
			} else if (lane > int(p0)) {
				d0 = up_d;
				i0 = up_i;
			}
			if (lane == 0) {
				d1 = carry_d;
				i1 = carry_i;
			} else {
				d1 = up1_d;
				i1 = up1_i;
			}
		} else {
			const uint32_t p1 = __popcll(__ballot(pair_lt(d1, i1, d, idx)));
			if (lane == int(p1)) {
				d1 = d;
				i1 = idx;
			} else if (lane > int(p1)) {
				d1 = up1_d;
				i1 = up1_i;
			}
		}
		if (filled < kk) ++filled;
		if (filled == kk) {
			const int e = int(kk) - 1;
			thr_d = e < 64 ? __shfl(d0, e) : __shfl(d1, e - 64);
			thr_i = e < 64 ? __shfl(i0, e) : __shfl(i1, e - 64);
		}
	}
	__device__ __forceinline__ bool admits(float d, uint32_t idx) const { return filled < kk || pair_lt(d, idx, thr_d, thr_i); }
};
constexpr int kMaxFusedK2 = 128;

// Fold of the 64 chains; every lane of the group ends with the row's sum.  `acc` holds chains 4m..4m+3.
template <bool kIpTail>
__device__ __forceinline__ float fold_chains(float4 acc, const float* __restrict__ row, const float* __restrict__ q, uint32_t dim,
											  int m) {
	// zmm (s0+s1)+(s2+s3): chain L with L^16, then L^32  ==  lane m with m^4, then m^8
	acc.x += __shfl_xor(acc.x, 4);
	acc.y += __shfl_xor(acc.y, 4);
	acc.z += __shfl_xor(acc.z, 4);
	acc.w += __shfl_xor(acc.w, 4);
	acc.x += __shfl_xor(acc.x, 8);
	acc.y += __shfl_xor(acc.y, 8);
	acc.z += __shfl_xor(acc.z, 8);
	acc.w += __shfl_xor(acc.w, 8);
	if constexpr (kIpTail) {
		// ip_dist.cc:61-65: 16-wide fmadd loop into the folded vector; lane m owns elements 4(m&3)..+3 of it
		const uint32_t i0 = dim & ~63u, i1 = dim & ~15u;
		for (uint32_t i = i0; i < i1; i += 16) {
			const float4 x = *reinterpret_cast<const float4*>(row + i + 4 * (m & 3));
			const float4 qq = *reinterpret_cast<const float4*>(q + i + 4 * (m & 3));
			acc.x = __builtin_fmaf(qq.x, x.x, acc.x);
			acc.y = __builtin_fmaf(qq.y, x.y, acc.y);
			acc.z = __builtin_fmaf(qq.z, x.z, acc.z);
			acc.w = __builtin_fmaf(qq.w, x.w, acc.w);
		}
	}
	// _mm512_reduce_add_ps: element c with c^8, then c^4  ==  lane m^2, then m^1; then (t0+t2)+(t1+t3)
	acc.x += __shfl_xor(acc.x, 2);
	acc.y += __shfl_xor(acc.y, 2);
	acc.z += __shfl_xor(acc.z, 2);
	acc.w += __shfl_xor(acc.w, 2);
	acc.x += __shfl_xor(acc.x, 1);
	acc.y += __shfl_xor(acc.y, 1);
	acc.z += __shfl_xor(acc.z, 1);
	acc.w += __shfl_xor(acc.w, 1);
	return (acc.x + acc.z) + (acc.y + acc.w);
}

template <int kMetric>
__device__ __forceinline__ void chain_step(float4& acc, const float4 q, const float4 x) {
	if constexpr (kMetric == kL2) {
		const float dx = q.x - x.x, dy = q.y - x.y, dz = q.z - x.z, dw = q.w - x.w;
		acc.x = __builtin_fmaf(dx, dx, acc.x);
		acc.y = __builtin_fmaf(dy, dy, acc.y);
		acc.z = __builtin_fmaf(dz, dz, acc.z);
		acc.w = __builtin_fmaf(dw, dw, acc.w);
	} else {
		acc.x = __builtin_fmaf(q.x, x.x, acc.x);
		acc.y = __builtin_fmaf(q.y, x.y, acc.y);
		acc.z = __builtin_fmaf(q.z, x.z, acc.z);
		acc.w = __builtin_fmaf(q.w, x.w, acc.w);
	}
}

// DistCalculator epilogue (hnswlib.h:147-165,192-197): alpha2 = 1, corrective offsets = 0 for fp32.
template <int kMetric>
__device__ __forceinline__ float metric_epilogue(float sum, const float* __restrict__ inv_norms, uint64_t row) {
	if constexpr (kMetric == kL2) {
		return 1.0f * sum + 0.0f + 0.0f;
	} else {
		float d = -(1.0f * sum + 0.0f + 0.0f);
		if constexpr (kMetric == kCos) d *= inv_norms[row];
		return d;
	}
}

// Any-dimension distance of the row owned by this 16-lane group (all 16 lanes return the same bits).
template <int kMetric>
__device__ __forceinline__ float group_distance_generic(const float* __restrict__ row, const float* __restrict__ q, uint32_t dim,
														 int m) {
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
	const uint32_t nb = dim >> 6;
	const float4* rp = reinterpret_cast<const float4*>(row) + m;
	const float4* qp = reinterpret_cast<const float4*>(q) + m;
#pragma unroll 4
	for (uint32_t t = 0; t < nb; ++t) chain_step<kMetric>(acc, qp[16 * t], rp[16 * t]);
	const float folded = fold_chains<kMetric != kL2>(acc, row, q, dim, m);
	// scalar tail: sequential fmaf chain from 0 (l2_dist.cc:12-26 / ip_dist.cc:10-20 as built in the pinned oracle)
	float tail = 0.0f;
	const uint32_t t0 = kMetric == kL2 ? (dim & ~63u) : (dim & ~15u);
	for (uint32_t i = t0; i < dim; ++i) {
		if constexpr (kMetric == kL2) {
			const float df = q[i] - row[i];
			tail = __builtin_fmaf(df, df, tail);
		} else {
			tail = __builtin_fmaf(q[i], row[i], tail);
		}
	}
	return folded + tail;
}

// the closure step of SearchRange on the device (hnsw_range_kernel)
struct HnswRange {
	const float* seed_dist;     // the ef-search's hits
	const uint32_t* seed_row;
	uint32_t seed_n;
	float radius;
	uint32_t* visited;          // [ceil(n / 32)], zeroed
	uint32_t* frontier;         // [2][cap]
	float* out_dist;            // [cap]
	uint32_t* out_row;
	unsigned long long* total;  // hits found (may exceed cap: then the expansion is incomplete and the caller retries with more room)
	uint64_t cap;
};

// SQ8: DistCalculator<uint8_t>::operator()(query, row, id) (hnswlib.h:147-165) over vector_dists::L2SqrDistance<uint8_t> /
// InnerProductDistance<uint8_t> (tools/distances/l2_dist.cc:168-199, ip_dist.cc:163-192, AVX-512 form).  The integer part is exact; the
// contract is in the REDUCTION: per 64-byte block the reference's zmm lane j collects elements {2j, 2j+1} and {32+2j, 33+2j}, the 16 lane
// sums are converted to float and added one after another (rounds past 2^24), then the scalar tail (an int) is added as a float.
// A 16-lane group owns a row: lane m reads the 4 bytes [4m, 4m+4) of every block with one 32-bit load and feeds reference lanes
// 2(m & 7) and 2(m & 7) + 1 (low half for m < 8, high half above) through v_dot4_u32_u8 on the masked halves of the word; an xor-8
// exchange completes the 16 reference sums, which every lane then adds in the reference's order.  A row is D bytes instead of 4 D:
// the search is bound by exactly these gathers.
template <int kMetric>
__device__ __forceinline__ void batch_distances_sq8(const HnswParams& p, const uint8_t* q, float qcorr, float qnorm, const uint32_t* ids, int cnt,
													float* dists, int lane) {
	const int m = lane & 15, g = lane >> 4;
	const uint32_t nblk = p.dim / 64, tail0 = nblk * 64;
	const bool words_ok = (p.dim & 3u) == 0;   // every row (and the query) then starts on a 4-byte boundary
	for (int base = 0; base < cnt; base += kRowsPerWave) {
		const int idx = base + g;
		const bool ok = idx < cnt;
		const uint64_t row = ids[ok ? idx : base];
		const uint8_t* r8 = p.codes + row * p.dim;
		uint32_t s0 = 0, s1 = 0;
		for (uint32_t t = 0; t < nblk; ++t) {
			uint32_t a, b;
			if (words_ok) {
				a = reinterpret_cast<const uint32_t*>(r8)[16 * t + m];
				b = reinterpret_cast<const uint32_t*>(q)[16 * t + m];
			} else {
				const uint8_t* pa = r8 + 64 * t + 4 * m;
				const uint8_t* pb = q + 64 * t + 4 * m;
				a = uint32_t(pa[0]) | (uint32_t(pa[1]) << 8) | (uint32_t(pa[2]) << 16) | (uint32_t(pa[3]) << 24);
				b = uint32_t(pb[0]) | (uint32_t(pb[1]) << 8) | (uint32_t(pb[2]) << 16) | (uint32_t(pb[3]) << 24);
			}
			const uint32_t alo = a & 0xFFFFu, ahi = a >> 16, blo = b & 0xFFFFu, bhi = b >> 16;
			if constexpr (kMetric == kL2) {   // (a - b)^2 = a^2 + b^2 - 2ab, exact in uint32 (the reference's madd_epi16 of the differences)
				s0 += __builtin_amdgcn_udot4(alo, alo, 0u, false) + __builtin_amdgcn_udot4(blo, blo, 0u, false) - 2u * __builtin_amdgcn_udot4(alo, blo, 0u, false);
				s1 += __builtin_amdgcn_udot4(ahi, ahi, 0u, false) + __builtin_amdgcn_udot4(bhi, bhi, 0u, false) - 2u * __builtin_amdgcn_udot4(ahi, bhi, 0u, false);
			} else {
				s0 = __builtin_amdgcn_udot4(alo, blo, s0, false);
				s1 = __builtin_amdgcn_udot4(ahi, bhi, s1, false);
			}
		}
		s0 += __shfl_xor(s0, 8, 64);   // low half (lanes 0-7) + high half (lanes 8-15) of the same reference lanes
		s1 += __shfl_xor(s1, 8, 64);
		float result = 0.f;
		const int group_base = lane & ~15;
#pragma unroll
		for (int t = 0; t < 8; ++t) {   // result += (float)lane[j], j = 0 .. 15
			result += float(__shfl(s0, group_base + t, 64));
			result += float(__shfl(s1, group_base + t, 64));
		}
		int tail = 0;   // the scalar tail: an int accumulator
		for (uint32_t i = tail0 + m; i < p.dim; i += 16) {
			if constexpr (kMetric == kL2) {
				const int df = int(r8[i]) - int(q[i]);
				tail += df * df;
			} else {
				tail += int(r8[i]) * int(q[i]);
			}
		}
#pragma unroll
		for (int off = 1; off < 16; off <<= 1) tail += __shfl_xor(tail, off, 64);
		result = result + float(tail);
		float dist;
		if constexpr (kMetric == kL2) {
			dist = p.alpha2 * result + qcorr + p.corr[row];
		} else {
			dist = -(p.alpha2 * result + qcorr + p.corr[row]);
			if constexpr (kMetric == kCos) dist *= p.inv_norms[row];
		}
		dist = qnorm * dist;
		if (ok && m == 0) dists[idx] = dist;
	}
}


}  // namespace rxgpu
