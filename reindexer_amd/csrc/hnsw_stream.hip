// HNSW streaming (batched) KNN on gfx950: BeginStreamingSearch / ContinueStreamingSearch
// (cpp_src/core/index/float_vector/hnswlib/hnswalg.h:1865-1975) with the streaming branches of initLayer0SearchState (:846-850),
// layer0ShouldStopBeforePop (:865-868), runLayer0Step (:882-893, 939-940), mergeExtrasIntoTopCandidates (:1893-1926) and
// emitStreamingBatch (:1928-1945).
//
// A session is a resumable best-first enumeration: every evaluated node goes into candidate_set, nodes enter top_candidates when
// they are POPPED, nodes pushed out of a full top_candidates are parked in top_candidates_extras, and each Continue(batch) runs
// with ef = max(ef, batch), then hands out the `batch` best of top_candidates.  One wavefront drives one session; the three
// heaps (PriorityQueue + CompareByFirst, replayed with the reference's sift mechanics so ties break identically) persist in the
// session's device memory between calls.  While they fit they are staged in LDS for the duration of a call (heap updates are
// dependent single-lane accesses: ~100 ns in LDS against ~1 us in HBM); a call that would outgrow LDS stops at a step boundary
// with status kStreamNeedGlobal and is resumed by the all-global variant — never on the CPU.
#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {

namespace {

struct Heap {
	float* d;
	uint32_t* i;
	int n;
};
// PriorityQueue<pair<float,tableint>, vector, CompareByFirst> (priority_queue.h:7-152, hnswalg.h:581-585), executed by ONE lane
__device__ __forceinline__ void sh_sift_up(Heap& h, int child) {
	const float vd = h.d[child];
	const uint32_t vi = h.i[child];
	while (child > 0) {
		const int parent = (child - 1) / 2;
		if (!(h.d[parent] < vd)) break;
		h.d[child] = h.d[parent];
		h.i[child] = h.i[parent];
		child = parent;
	}
	h.d[child] = vd;
	h.i[child] = vi;
}
__device__ __forceinline__ void sh_sift_down(Heap& h, int parent, int size) {
	const float vd = h.d[parent];
	const uint32_t vi = h.i[parent];
	for (;;) {
		const int left = parent * 2 + 1;
		if (left >= size) break;
		int best = left;
		const int right = left + 1;
		if (right < size && h.d[left] < h.d[right]) best = right;
		if (!(vd < h.d[best])) break;
		h.d[parent] = h.d[best];
		h.i[parent] = h.i[best];
		parent = best;
	}
	h.d[parent] = vd;
	h.i[parent] = vi;
}
__device__ __forceinline__ void sh_emplace(Heap& h, float vd, uint32_t vi) {
	h.d[h.n] = vd;
	h.i[h.n] = vi;
	++h.n;
	if (h.n >= 2) sh_sift_up(h, h.n - 1);
}
__device__ __forceinline__ void sh_pop(Heap& h) {
	if (h.n >= 2) {
		const float td = h.d[0];
		const uint32_t ti = h.i[0];
		h.d[0] = h.d[h.n - 1];
		h.i[0] = h.i[h.n - 1];
		h.d[h.n - 1] = td;
		h.i[h.n - 1] = ti;
		if (h.n > 2) sh_sift_down(h, 0, h.n - 1);
	}
	--h.n;
}
__device__ __forceinline__ void sh_replace_top(Heap& h, float vd, uint32_t vi) {
	h.d[0] = vd;
	h.i[0] = vi;
	sh_sift_down(h, 0, h.n);
}

template <int kMetric>
__device__ __forceinline__ void stream_distances(const HnswParams& p, const HnswStream& s, const float* q, const uint32_t* ids, int cnt, float* dists,
												 int lane) {
	if (p.codes) {   // quantised graph (HierarchicalNSWImpl<uint8_t>): the same session over codes — wave-uniform branch
		batch_distances_sq8<kMetric>(p, s.qcodes, s.qcorr, s.qnorm, ids, cnt, dists, lane);
		return;
	}
	const int m = lane & 15, g = lane >> 4;
	for (int base = 0; base < cnt; base += kRowsPerWave) {
		const int idx = base + g;
		const bool ok = idx < cnt;
		const uint64_t row = ids[ok ? idx : base];
		const float sum = group_distance_generic<kMetric>(p.rows + row * p.stride, q, p.dim, m);
		const float dist = 1.0f * metric_epilogue<kMetric>(sum, p.inv_norms, row);
		if (ok && m == 0) dists[idx] = dist;
	}
}

}  // namespace

// mode: kStreamBegin = descent + initLayer0SearchState; kStreamContinue = one ContinueStreamingSearch; kStreamResume = the same call
// continued after kStreamNeedGlobal (no second mergeExtras).  kLds: heaps staged in LDS for this call.
template <int kMetric, bool kLds>
__global__ __launch_bounds__(64) void hnsw_stream_kernel(HnswParams p, HnswStream s, uint32_t batch, int mode) {
	extern __shared__ __attribute__((aligned(16))) unsigned char stream_lds[];
	__shared__ uint32_t nb_id[kHnswMaxNeighbors];
	__shared__ float nb_d[kHnswMaxNeighbors];
	__shared__ uint32_t s_cur;
	__shared__ int s_flag;
	__shared__ int s_n[3];
	const int lane = threadIdx.x;
	const float* q = s.query;
	HnswStreamState* st = s.state;

	if (mode == kStreamBegin) {
		// ---- getLayer0EntryPoint (hnswalg.h:799-827)
		uint32_t cur = p.entry;
		if (lane == 0) nb_id[0] = cur;
		__syncthreads();
		stream_distances<kMetric>(p, s, q, nb_id, 1, nb_d, lane);
		__syncthreads();
		float curdist = nb_d[0];
		for (int level = p.maxlevel; level > 0; --level) {
			bool changed = true;
			while (changed) {
				__syncthreads();
				const uint32_t* ll = p.upper + (p.upper_off[cur] + uint64_t(level - 1)) * (1 + p.M);
				const int cnt = int(ll[0]);
				for (int j = lane; j < cnt; j += 64) nb_id[j] = ll[1 + j];
				__syncthreads();
				stream_distances<kMetric>(p, s, q, nb_id, cnt, nb_d, lane);
				__syncthreads();
				changed = false;
				for (int i = 0; i < cnt; ++i) {
					const float d = nb_d[i];
					if (d < curdist) {
						curdist = d;
						cur = nb_id[i];
						changed = true;
					}
				}
			}
		}
		// ---- initLayer0SearchState, streaming: the entry point only enters candidate_set
		if (lane == 0) {
			const bool ep_ok = p.bare || !p.deleted[cur];
			const float lower = ep_ok ? curdist : 3.402823466e+38f;
			s.cand_d[0] = -lower;
			s.cand_i[0] = cur;
			s.visited[cur >> 5] |= 1u << (cur & 31);
			st->cand_n = 1;
			st->top_n = 0;
			st->ext_n = 0;
			st->lower = lower;
			st->top_sel = 0;
			st->ext_sel = 0;
			st->status = kStreamOk;
			st->out_count = 0;
			st->exhausted = 0;
		}
		return;
	}

	// ---- heaps: LDS staging or the session's global arrays
	const uint32_t cap_top = kLds ? uint32_t(kStreamLdsTop) : s.cap;
	const uint32_t cap_ext = kLds ? uint32_t(kStreamLdsExt) : s.cap;
	const uint32_t cap_cand = kLds ? uint32_t(kStreamLdsCand) : s.cap;
	Heap top[2], ext[2], cand;
	uint32_t tsel = st->top_sel, esel = st->ext_sel;
	if constexpr (kLds) {
		float* f = reinterpret_cast<float*>(stream_lds);
		top[0] = Heap{f, reinterpret_cast<uint32_t*>(f + kStreamLdsTop), 0};
		f += 2 * kStreamLdsTop;
		top[1] = Heap{f, reinterpret_cast<uint32_t*>(f + kStreamLdsTop), 0};
		f += 2 * kStreamLdsTop;
		ext[0] = Heap{f, reinterpret_cast<uint32_t*>(f + kStreamLdsExt), 0};
		f += 2 * kStreamLdsExt;
		ext[1] = Heap{f, reinterpret_cast<uint32_t*>(f + kStreamLdsExt), 0};
		f += 2 * kStreamLdsExt;
		cand = Heap{f, reinterpret_cast<uint32_t*>(f + kStreamLdsCand), 0};
		// stage in (the launcher guarantees the sizes fit); current heaps land in slot 0
		const int tn = st->top_n, en = st->ext_n, cn = st->cand_n;
		const float* gtd = s.top_d + size_t(tsel) * s.cap;
		const uint32_t* gti = s.top_i + size_t(tsel) * s.cap;
		const float* ged = s.ext_d + size_t(esel) * s.cap;
		const uint32_t* gei = s.ext_i + size_t(esel) * s.cap;
		for (int j = lane; j < tn; j += 64) {
			top[0].d[j] = gtd[j];
			top[0].i[j] = gti[j];
		}
		for (int j = lane; j < en; j += 64) {
			ext[0].d[j] = ged[j];
			ext[0].i[j] = gei[j];
		}
		for (int j = lane; j < cn; j += 64) {
			cand.d[j] = s.cand_d[j];
			cand.i[j] = s.cand_i[j];
		}
		tsel = 0;
		esel = 0;
		top[0].n = tn;
		ext[0].n = en;
		cand.n = cn;
		__syncthreads();
	} else {
		for (int k = 0; k < 2; ++k) {
			top[k] = Heap{s.top_d + size_t(k) * s.cap, s.top_i + size_t(k) * s.cap, 0};
			ext[k] = Heap{s.ext_d + size_t(k) * s.cap, s.ext_i + size_t(k) * s.cap, 0};
		}
		cand = Heap{s.cand_d, s.cand_i, st->cand_n};
		top[tsel].n = st->top_n;
		ext[esel].n = st->ext_n;
	}
	float lower = st->lower;
	const int ef = int(s.ef > batch ? s.ef : batch);   // state.ef = max(state.ef, batchSize), hnswalg.h:1960-1961
	uint32_t status = kStreamOk;

	// ---- mergeExtrasIntoTopCandidates (hnswalg.h:1893-1926), lane 0
	if (mode == kStreamContinue && lane == 0) {
		Heap& T = top[tsel];
		if (!(T.n >= ef || ext[esel].n == 0)) {
			if (T.n) {
				Heap& te = ext[esel];
				Heap& ne = ext[esel ^ 1];
				ne.n = 0;
				const int need = ef - T.n;
				int delta = te.n > need ? te.n - need : 0;
				while (delta-- > 0) {
					sh_emplace(ne, te.d[0], te.i[0]);
					sh_pop(te);
				}
				while (te.n && T.n < ef) {
					sh_emplace(T, te.d[0], te.i[0]);
					sh_pop(te);
				}
				esel ^= 1;
			} else {
				// top_candidates = move(extras): the array (heap layout included) becomes the top heap
				Heap& E = ext[esel];
				if (uint32_t(E.n) > cap_top) {
					status = kStreamError;   // unreachable: the launcher stages in LDS only when kStreamLdsExt <= kStreamLdsTop holds the heap
				} else {
					for (int j = 0; j < E.n; ++j) {
						T.d[j] = E.d[j];
						T.i[j] = E.i[j];
					}
					T.n = E.n;
					E.n = 0;
					while (T.n > ef) {
						sh_emplace(E, T.d[0], T.i[0]);
						sh_pop(T);
					}
				}
			}
			if (T.n) lower = T.d[0];
		}
	}
	if (lane == 0) s_flag = int(status);
	__syncthreads();
	status = uint32_t(s_flag);
	__syncthreads();

	bool finished = false;
	if (status == kStreamOk) {
		for (;;) {
			if (lane == 0) {
				Heap& T = top[tsel];
				int flag = 0;
				if (cand.n == 0) {
					flag = 1;
				} else {
					const float cdist = -cand.d[0];
					if (cdist > lower && T.n >= ef) {   // streaming: never the bare-bone shortcut (hnswalg.h:865-868)
						flag = 1;
					} else if (kLds && (uint32_t(cand.n) + p.maxM0 > cap_cand || uint32_t(ext[esel].n) + 1 > cap_ext)) {
						flag = 2;   // stop at a step boundary; the global variant resumes
					} else {
						const uint32_t id = cand.i[0];
						sh_pop(cand);
						if (p.bare || !p.deleted[id]) {   // hnswalg.h:882-893
							if (T.n < ef) {
								sh_emplace(T, cdist, id);
							} else if (lower > cdist) {
								const float od = T.d[0];
								const uint32_t oi = T.i[0];
								sh_replace_top(T, cdist, id);
								sh_emplace(ext[esel], od, oi);
							}
							lower = T.d[0];
						}
						s_cur = id;
					}
				}
				s_flag = flag;
			}
			__syncthreads();
			const int flag = s_flag;
			if (flag) {
				finished = flag == 1;
				if (flag == 2) status = kStreamNeedGlobal;
				break;
			}
			const uint32_t node = s_cur;
			const uint32_t* ll = p.links0 + size_t(node) * (1 + p.maxM0);
			const int cnt = int(ll[0]);
			int nfresh = 0;
			for (int base = 0; base < cnt; base += 64) {
				const int j = base + lane;
				bool fresh = false;
				uint32_t id = 0;
				if (j < cnt) {
					id = ll[1 + j];
					const uint32_t bit = 1u << (id & 31);
					fresh = !(atomicOr(&s.visited[id >> 5], bit) & bit);
				}
				const uint64_t fm = __ballot(fresh);
				if (fresh) nb_id[nfresh + __popcll(fm & ((1ull << lane) - 1))] = id;
				nfresh += __popcll(fm);
			}
			__syncthreads();
			stream_distances<kMetric>(p, s, q, nb_id, nfresh, nb_d, lane);
			__syncthreads();
			if (lane == 0) {
				for (int i = 0; i < nfresh; ++i) sh_emplace(cand, -nb_d[i], nb_id[i]);   // streaming: every evaluated node (hnswalg.h:939-940)
			}
			__syncthreads();
		}
	}

	// ---- emitStreamingBatch (hnswalg.h:1928-1945), lane 0
	if (lane == 0) {
		uint32_t emitted = 0;
		if (finished) {
			Heap& A = top[tsel];
			Heap& B = top[tsel ^ 1];
			B.n = 0;
			while (A.n > int(batch)) {
				sh_emplace(B, A.d[0], A.i[0]);
				sh_pop(A);
			}
			while (A.n > 0) {
				s.out_dist[emitted] = A.d[0];
				s.out_row[emitted] = A.i[0];
				++emitted;
				sh_pop(A);
			}
			tsel ^= 1;
			st->exhausted = (cand.n == 0 && top[tsel].n == 0 && ext[esel].n == 0) ? 1u : 0u;
		}
		st->out_count = emitted;
		st->status = status;
		st->lower = lower;
		st->cand_n = cand.n;
		st->top_n = top[tsel].n;
		st->ext_n = ext[esel].n;
		if constexpr (!kLds) {
			st->top_sel = tsel;
			st->ext_sel = esel;
		} else {
			st->top_sel = 0;
			st->ext_sel = 0;
		}
		s_cur = tsel | (esel << 1);
		s_n[0] = top[tsel].n;
		s_n[1] = ext[esel].n;
		s_n[2] = cand.n;
	}
	__syncthreads();
	if constexpr (kLds) {   // stage out into slot 0 of the session arrays
		const uint32_t sel = s_cur;
		const Heap& T = top[sel & 1];
		const Heap& E = ext[(sel >> 1) & 1];
		const int tn = s_n[0], en = s_n[1], cn = s_n[2];
		for (int j = lane; j < tn; j += 64) {
			s.top_d[j] = T.d[j];
			s.top_i[j] = T.i[j];
		}
		for (int j = lane; j < en; j += 64) {
			s.ext_d[j] = E.d[j];
			s.ext_i[j] = E.i[j];
		}
		for (int j = lane; j < cn; j += 64) {
			s.cand_d[j] = cand.d[j];
			s.cand_i[j] = cand.i[j];
		}
	}
}

void launch_hnsw_stream(int metric, const HnswParams& p, const HnswStream& s, uint32_t batch, int mode, bool lds, hipStream_t st) {
	const size_t lds_bytes = lds ? size_t(2 * kStreamLdsTop + 2 * kStreamLdsExt + kStreamLdsCand) * 8 : 0;
#define RX_STREAM(M, L)                                                                                                      \
	do {                                                                                                                     \
		static std::atomic<uint64_t> raised{0};                                                                              \
		if (L) (void)raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&hnsw_stream_kernel<M, L>), lds_bytes);     \
		hipLaunchKernelGGL((hnsw_stream_kernel<M, L>), dim3(1), dim3(64), lds_bytes, st, p, s, batch, mode);                  \
	} while (0)
	if (lds) {
		switch (metric) {
			case kL2: RX_STREAM(kL2, true); break;
			case kIP: RX_STREAM(kIP, true); break;
			default: RX_STREAM(kCos, true); break;
		}
	} else {
		switch (metric) {
			case kL2: RX_STREAM(kL2, false); break;
			case kIP: RX_STREAM(kIP, false); break;
			default: RX_STREAM(kCos, false); break;
		}
	}
#undef RX_STREAM
}

}  // namespace rxgpu
