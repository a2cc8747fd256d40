// How much do N workgroups pay for ONE same-address agent-scope atomic each (the done / ready counters of a chained kernel)?
//   hipcc --offload-arch=gfx950 -O3 atomic_contention.hip -o atomic_contention && ./atomic_contention
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(unsigned* c) { if (c == nullptr) __builtin_trap(); }
__global__ void k_noret(unsigned* c) { if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_ret(unsigned* c, unsigned* out) {
    if (threadIdx.x == 0) { unsigned d = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (d == 0xFFFFFFFFu) out[0] = d; }
}
__global__ void k_tree(unsigned* c, unsigned* out) {  // 32 per leaf, the leaf's last arrival goes up
    if (threadIdx.x == 0) {
        unsigned d = __hip_atomic_fetch_add(c + 64 + (blockIdx.x >> 5) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((d & 31u) == 31u) { unsigned e = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (e == 0xFFFFFFFFu) out[0] = e; }
    }
}
__global__ void k_percu(unsigned* c) {  // distinct addresses (128 B apart): the floor for "one atomic per workgroup"
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c + blockIdx.x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// polling: every workgroup reads `n` 16-byte records with sc1 loads (L1 bypass) vs plain loads
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k_read_sc1(const unsigned char* rec, unsigned* out, int n) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(rec), 0, 0x7FFFFFFF, 0x00020000);
    unsigned acc = 0;
    for (int t = threadIdx.x; t < n; t += 256) { u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, t * 32, 0, 16); acc += v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void k_read_plain(const unsigned char* rec, unsigned* out, int n) {
    unsigned acc = 0;
    for (int t = threadIdx.x; t < n; t += 256) { u4 v = *reinterpret_cast<const u4*>(rec + t * 32); acc += v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
template <typename F> float timeit(F f, int reps = 200) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
    unsigned *c, *out; unsigned char* rec;
    hipMalloc(&c, 1 << 20); hipMemset(c, 0, 1 << 20); hipMalloc(&out, 64); hipMalloc(&rec, 1 << 20); hipMemset(rec, 0, 1 << 20);
    for (int wgs : {256, 1024, 2048}) {
        std::printf("%5d workgroups: empty %.2f us | one atomic each, same address: no return %.2f, returned %.2f | tree 32 x 32 %.2f | distinct addresses %.2f\n",
                    wgs, timeit([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0, c); }),
                    timeit([&] { hipLaunchKernelGGL(k_noret, dim3(wgs), dim3(256), 0, 0, c); }),
                    timeit([&] { hipLaunchKernelGGL(k_ret, dim3(wgs), dim3(256), 0, 0, c, out); }),
                    timeit([&] { hipLaunchKernelGGL(k_tree, dim3(wgs), dim3(256), 0, 0, c, out); }),
                    timeit([&] { hipLaunchKernelGGL(k_percu, dim3(wgs), dim3(256), 0, 0, c); }));
        for (int n : {64, 1024})
            std::printf("      every workgroup reads %4d records: sc1 loads %.2f us, plain loads %.2f us\n", n,
                        timeit([&] { hipLaunchKernelGGL(k_read_sc1, dim3(wgs), dim3(256), 0, 0, rec, out, n); }),
                        timeit([&] { hipLaunchKernelGGL(k_read_plain, dim3(wgs), dim3(256), 0, 0, rec, out, n); }));
    }
    return 0;
}
