// Dependent-instruction latencies of ONE wave on gfx950 (the numbers every chain kernel of this library is designed against:
// inverse RANK, rANS order 1, FPAQ, LZ parse, inverse BWT). Each kernel runs `iters` iterations of 16 dependent copies of a
// pattern on a single wave; the host times it with HIP events and prints ns per pattern instance.
//   hipcc --offload-arch=gfx950 -O2 -o lat_bench tools/gpu/lat_bench.hip && ./lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ void k_valu_dep(uint32_t* out, int iters) {
    uint32_t v = threadIdx.x;
    for (int i = 0; i < iters; i++) asm volatile(REP16("v_add_u32 %0, %0, %0\n") : "+v"(v));
    out[threadIdx.x] = v;
}
__global__ void k_valu_indep(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; i++)
        asm volatile("v_add_u32 %0, %0, %0\nv_add_u32 %1, %1, %1\nv_add_u32 %2, %2, %2\nv_add_u32 %3, %3, %3\n"
                     "v_add_u32 %0, %0, %0\nv_add_u32 %1, %1, %1\nv_add_u32 %2, %2, %2\nv_add_u32 %3, %3, %3\n"
                     "v_add_u32 %0, %0, %0\nv_add_u32 %1, %1, %1\nv_add_u32 %2, %2, %2\nv_add_u32 %3, %3, %3\n"
                     "v_add_u32 %0, %0, %0\nv_add_u32 %1, %1, %1\nv_add_u32 %2, %2, %2\nv_add_u32 %3, %3, %3\n"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    out[threadIdx.x] = a + b + c + d;
}
__global__ void k_salu_dep(uint32_t* out, int iters) {
    uint32_t s = iters;
    for (int i = 0; i < iters; i++) asm volatile(REP16("s_add_u32 %0, %0, %0\n") : "+s"(s)::"scc");
    out[threadIdx.x] = s;
}
__global__ void k_salu_indep(uint32_t* out, int iters) {
    uint32_t a = iters, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; i++)
        asm volatile("s_add_u32 %0, %0, %0\ns_add_u32 %1, %1, %1\ns_add_u32 %2, %2, %2\ns_add_u32 %3, %3, %3\n"
                     "s_add_u32 %0, %0, %0\ns_add_u32 %1, %1, %1\ns_add_u32 %2, %2, %2\ns_add_u32 %3, %3, %3\n"
                     "s_add_u32 %0, %0, %0\ns_add_u32 %1, %1, %1\ns_add_u32 %2, %2, %2\ns_add_u32 %3, %3, %3\n"
                     "s_add_u32 %0, %0, %0\ns_add_u32 %1, %1, %1\ns_add_u32 %2, %2, %2\ns_add_u32 %3, %3, %3\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d)::"scc");
    out[threadIdx.x] = a + b + c + d;
}
// v_readlane -> s_and (lane select of the next v_readlane): VALU -> SGPR -> SALU -> VALU(lane select)
__global__ void k_readlane_salu(uint32_t* out, int iters) {
    uint32_t v = (threadIdx.x * 7 + 3) & 63, l = 1, t = 0;
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("v_readlane_b32 %1, %2, %0\ns_and_b32 %0, %1, 63\n") : "+s"(l), "+s"(t) : "v"(v) : "scc");
    out[threadIdx.x] = l + t;
}
// v_readlane -> VALU consuming the SGPR -> v_readfirstlane ... (VALU -> SGPR -> VALU operand)
__global__ void k_readlane_valu(uint32_t* out, int iters) {
    uint32_t v = threadIdx.x, t = 0;
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("v_readlane_b32 %1, %0, 5\ns_nop 1\nv_add_u32 %0, %1, %0\n") : "+v"(v), "+s"(t));
    out[threadIdx.x] = v + t;
}
// SALU result consumed by a VALU whose result goes back to the SALU through v_readfirstlane
__global__ void k_salu_valu_roundtrip(uint32_t* out, int iters) {
    uint32_t v = threadIdx.x, s = 3;
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("s_add_u32 %1, %1, 1\nv_add_u32 %0, %1, %0\nv_readfirstlane_b32 %1, %0\n") : "+v"(v), "+s"(s)::"scc");
    out[threadIdx.x] = v + s;
}
// v_cmp (VCC) -> v_cndmask -> v_cmp ...
__global__ void k_cmp_cndmask(uint32_t* out, int iters) {
    uint32_t v = threadIdx.x, w = 77;
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\ns_nop 1\nv_cndmask_b32 %0, %1, %0, vcc\n") : "+v"(v) : "v"(w) : "vcc");
    out[threadIdx.x] = v;
}
// v_cmp into an SGPR pair -> s_bcnt1 -> VALU using the count
__global__ void k_cmp_bcnt(uint32_t* out, int iters) {
    uint32_t v = threadIdx.x, c = 0;
    uint64_t m = 0;
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("v_cmp_lt_u32 %2, %0, 33\ns_bcnt1_i32_b64 %1, %2\nv_add_u32 %0, %1, %0\n") : "+v"(v), "+s"(c), "+s"(m)::"scc");
    out[threadIdx.x] = v + c;
}
__global__ void k_dpp_dep(uint32_t* out, int iters) {
    uint32_t v = threadIdx.x;
    for (int i = 0; i < iters; i++) asm volatile(REP16("s_nop 1\nv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(v));
    out[threadIdx.x] = v;
}
// the skeleton of the inverse RANK step: readlane -> 2 SALU -> v_cmp(SGPR) -> v_cndmask -> readlane
__global__ void k_rank_skeleton(uint32_t* out, int iters) {
    uint32_t e = threadIdx.x * 256 + threadIdx.x, q = 1000 - threadIdx.x, t = 0, l = 3;
    for (int i = 0; i < iters; i++)
        asm volatile(REP16("v_readlane_b32 %2, %0, %3\ns_add_u32 %2, %2, 0x1234\ns_lshr_b32 %2, %2, 9\nv_cmp_gt_i32 vcc, %1, %2\ns_nop 1\n"
                           "v_cndmask_b32 %0, %0, %1, vcc\n")
                     : "+v"(e), "+v"(q), "+s"(t), "+s"(l)::"vcc", "scc");
    out[threadIdx.x] = e + t;
}
// LDS pointer chase (one lane's address depends on the previous read)
__global__ void k_lds_chase(uint32_t* out, int iters) {
    __shared__ uint32_t tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) tab[i] = ((i * 1103515245u + 12345u) >> 4) & 4095;
    __syncthreads();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = tab[x];
    }
    out[threadIdx.x] = x;
}
// global pointer chase over `n` words (L2-resident when small, HBM when large)
__global__ void k_global_chase(const uint32_t* tab, uint32_t* out, int iters) {
    uint32_t x = threadIdx.x & 7;                        // 8 distinct chains per wave, like the inverse BWT
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = tab[x];
    }
    out[threadIdx.x] = x;
}
// scalar-load pointer chase (s_load_dword through the scalar cache)
__global__ void k_sload_chase(const uint32_t* tab, uint32_t* out, int iters) {
    uint32_t x = 0;
    const __attribute__((address_space(4))) uint32_t* t = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)tab;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = t[x];
    }
    out[threadIdx.x] = x;
}
// taken branch per pattern
__global__ void k_branch(uint32_t* out, int iters) {
    uint32_t s = 0;
    for (int i = 0; i < iters * 16; i++) asm volatile("s_add_u32 %0, %0, 1\n" : "+s"(s)::"scc");
    out[threadIdx.x] = s;
}

// shader clock during a lone-wave kernel: s_memtime (core clock) against s_memrealtime (100 MHz)
__global__ void k_clock(uint64_t* out, int iters) {
    uint32_t v = threadIdx.x;
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) asm volatile(REP16("v_add_u32 %0, %0, %0\n") : "+v"(v));
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = v; }
}

template <typename F>
static void run(const char* name, int iters, int lanes, F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch(iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("{\"pattern\": \"%s\", \"ns_per_instance\": %.2f}\n", name, ms * 1e6 / ((double)iters * 16));
    fflush(stdout);
}

// tab[i] = (a * i + c) mod words: one cycle over all words (Hull-Dobell: c odd, a = 1 mod 4), filled on the device
__global__ void k_fill_cycle(uint32_t* tab, uint32_t mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= mask) tab[i] = (i * 1664525u + 1013904223u) & mask;
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 4096);
    const int N = 40000;
    {
        uint64_t* d;
        uint64_t h[3];
        hipMalloc(&d, 64);
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_clock, 1, 64, 0, 0, d, 400000 << rep);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("{\"pattern\": \"clock probe: %d x16 dependent v_add\", \"s_memtime_ticks\": %llu, \"s_memrealtime_ticks_100MHz\": %llu, \"memtime_MHz\": %.1f, \"memtime_ticks_per_v_add\": %.3f}\n",
                   400000 << rep, (unsigned long long)h[0], (unsigned long long)h[1], h[0] * 100.0 / h[1], (double)h[0] / ((double)(400000 << rep) * 16));
            fflush(stdout);
        }
        hipFree(d);
    }
    run("valu_dep: v_add -> v_add", N, 64, [&](int it) { hipLaunchKernelGGL(k_valu_dep, 1, 64, 0, 0, out, it); });
    run("valu_indep: v_add (4 independent chains)", N, 64, [&](int it) { hipLaunchKernelGGL(k_valu_indep, 1, 64, 0, 0, out, it); });
    run("salu_dep: s_add -> s_add", N, 64, [&](int it) { hipLaunchKernelGGL(k_salu_dep, 1, 64, 0, 0, out, it); });
    run("salu_indep: s_add (4 independent chains)", N, 64, [&](int it) { hipLaunchKernelGGL(k_salu_indep, 1, 64, 0, 0, out, it); });
    run("v_readlane -> s_and -> (lane select) v_readlane", N, 64, [&](int it) { hipLaunchKernelGGL(k_readlane_salu, 1, 64, 0, 0, out, it); });
    run("v_readlane -> s_nop 1 -> v_add(sgpr) -> v_readlane", N, 64, [&](int it) { hipLaunchKernelGGL(k_readlane_valu, 1, 64, 0, 0, out, it); });
    run("s_add -> v_add(sgpr) -> v_readfirstlane -> s_add", N, 64, [&](int it) { hipLaunchKernelGGL(k_salu_valu_roundtrip, 1, 64, 0, 0, out, it); });
    run("v_cmp(vcc) -> s_nop 1 -> v_cndmask -> v_cmp", N, 64, [&](int it) { hipLaunchKernelGGL(k_cmp_cndmask, 1, 64, 0, 0, out, it); });
    run("v_cmp(sgpr pair) -> s_bcnt1 -> v_add(sgpr) -> v_cmp", N, 64, [&](int it) { hipLaunchKernelGGL(k_cmp_bcnt, 1, 64, 0, 0, out, it); });
    run("s_nop 1 -> v_mov_dpp wave_shr:1 -> (same)", N, 64, [&](int it) { hipLaunchKernelGGL(k_dpp_dep, 1, 64, 0, 0, out, it); });
    run("rank skeleton: v_readlane -> s_add -> s_lshr -> v_cmp -> s_nop 1 -> v_cndmask", N, 64, [&](int it) { hipLaunchKernelGGL(k_rank_skeleton, 1, 64, 0, 0, out, it); });
    run("lds chase: ds_read -> ds_read", N / 4, 64, [&](int it) { hipLaunchKernelGGL(k_lds_chase, 1, 64, 0, 0, out, it); });
    run("loop: s_add + s_cmp + taken s_cbranch", N, 64, [&](int it) { hipLaunchKernelGGL(k_branch, 1, 64, 0, 0, out, it); });
    for (size_t words : {(size_t)1 << 12, (size_t)1 << 18, (size_t)1 << 22, (size_t)1 << 28}) {
        uint32_t* d;
        hipMalloc(&d, words * 4);
        hipLaunchKernelGGL(k_fill_cycle, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, 0, d, (uint32_t)(words - 1));
        hipDeviceSynchronize();
        std::string nm = "global chase over " + std::to_string(words * 4 >> 10) + " KiB: global_load -> global_load";
        run(nm.c_str(), 4000, 64, [&](int it) { hipLaunchKernelGGL(k_global_chase, 1, 64, 0, 0, d, out, it); });
        if (words <= ((size_t)1 << 22)) {
            std::string ns = "scalar chase over " + std::to_string(words * 4 >> 10) + " KiB: s_load -> s_load";
            run(ns.c_str(), 4000, 64, [&](int it) { hipLaunchKernelGGL(k_sload_chase, 1, 64, 0, 0, d, out, it); });
        }
        hipFree(d);
    }
    return 0;
}
