// Host-side fp32 distances used ONLY while BUILDING an HNSW graph (graph construction stays on the CPU — SURVEY §8 a16;
// every SEARCH distance is computed by the HIP kernels).  The summation order is the reference's AVX-512 order
// (cpp_src/tools/distances/l2_dist.cc:38-72, ip_dist.cc:31-70): 64 independent fmaf chains, chain L owning i == L (mod 64),
// folded (s0+s1)+(s2+s3) -> 16 -> 8 -> 4 -> (t0+t2)+(t1+t3), IP's 16-wide tail loop, then a sequential scalar tail — so the
// graph built here is link-for-link the graph the reference builds.  The inner loops are over independent chains, which
// lets the compiler vectorise them without changing a single rounding.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstring>

namespace rxgpu::host {

float L2SqrAvx512Order(const float* a, const float* b, size_t d) noexcept;
float InnerProductAvx512Order(const float* a, const float* b, size_t d) noexcept;

}  // namespace rxgpu::host
