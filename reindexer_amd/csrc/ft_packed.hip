// PackedIdRelVec streams decoded on the device: the dictionary's posting lists arrive as the reference keeps them
// (cpp_src/core/ft/idrelset.h:155-280, one varint stream per word) in ONE upload, and come out as the flat arrays of the merge kernels —
// instead of the host flattening every word (PositionPostings::AppendPacked) and uploading eight arrays per word.
//
// One thread per word, two launches: count (postings / positions / entries, validation), then — after the host has laid the words out
// in one pool — write.  A stream has no sync points (element lengths depend on flags inside the elements, ids and fields are
// delta-coded against the element before), so a single list is decoded serially; the parallelism is the dictionary's (10^5..10^6 words,
// most of them short).  Lists beyond kFtPackedDeviceMaxBytes stay with the host decoder — see rxgpu_ft_set_words_packed; decoding one long
// list with all lanes (speculative element starts + pointer jumping) is DESIGN.md 6.5.
//
// HBM-bound byte work in principle (bytes in, ~4x the bytes out); in this thread-per-word form the rate is set by the longest list of
// a launch and by uncoalesced per-thread streams — measured numbers in DESIGN.md 5.4b.
#include "ft_packed_decode.h"
#include "rxgpu_internal.h"

namespace rxgpu {

__global__ __launch_bounds__(256) void ft_packed_count(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords,
														 uint32_t num_fields, FtPackedCounts* counts) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	const uint64_t b0 = byte_off[w], b1 = byte_off[w + 1];
	counts[w] = ft_decode_packed(bytes + b0, b1 - b0, array_found_pos[w], num_fields, kFtRangeDocs, FtPackedOut{});
}

__global__ __launch_bounds__(256) void ft_packed_write(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords,
														 uint32_t num_fields, const FtPackedOut* outs, FtPackedCounts* counts) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	const uint64_t b0 = byte_off[w], b1 = byte_off[w + 1];
	if (!outs[w].doc) return;   // an empty word
	counts[w] = ft_decode_packed(bytes + b0, b1 - b0, array_found_pos[w], num_fields, kFtRangeDocs, outs[w]);
}

hipError_t launch_ft_packed_count(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords, uint32_t num_fields,
								   FtPackedCounts* counts, hipStream_t st) {
	if (!nwords) return hipSuccess;
	hipLaunchKernelGGL(ft_packed_count, dim3((nwords + 255) / 256), dim3(256), 0, st, bytes, byte_off, array_found_pos, nwords, num_fields, counts);
	return hipGetLastError();
}

hipError_t launch_ft_packed_write(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords, uint32_t num_fields,
								   const FtPackedOut* outs, FtPackedCounts* counts, hipStream_t st) {
	if (!nwords) return hipSuccess;
	hipLaunchKernelGGL(ft_packed_write, dim3((nwords + 255) / 256), dim3(256), 0, st, bytes, byte_off, array_found_pos, nwords, num_fields, outs, counts);
	return hipGetLastError();
}

}  // namespace rxgpu
